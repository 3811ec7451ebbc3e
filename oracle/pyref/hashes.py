"""Hash primitives for the CPU oracle (TEST INFRASTRUCTURE ONLY).

keccak256_legacy : Keccak-256 with the ORIGINAL 0x01 padding (not SHA3's 0x06), i.e. what
                   go-ethereum/crypto/sha3 NewKeccak256 computes -- EthereumSum256,
                   curves/altbn128.go:517-522.  Written out here because hashlib lacks it.
blake2b512       : unkeyed BLAKE2b-512 (golang.org/x/crypto/blake2b New512(nil),
                   curves/bls12_381.go:362-367,397-400); hashlib's is the same function.
"""
import hashlib

_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
       0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
       0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
       0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
       0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M if n else x


def keccak_f1600(A):
    """A[x][y] lanes, in place."""
    for rc in _RC:
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        for x in range(5):
            for y in range(5):
                A[x][y] ^= D[x]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        for x in range(5):
            for y in range(5):
                A[x][y] = B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y] & _M)
        A[0][0] ^= rc
    return A


def keccak256_legacy(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    padlen = rate - (len(msg) % rate)
    pad = bytearray(padlen)
    pad[0] ^= 0x01
    pad[-1] ^= 0x80
    msg += pad
    A = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            A[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i: off + 8 * i + 8], "little")
        keccak_f1600(A)
    out = b"".join(A[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    return out


def blake2b512(data: bytes) -> bytes:
    return hashlib.blake2b(data, digest_size=64).digest()


# ---- BLAKE2b with an explicit parameter block, and BLAKE2Xb on top of it --------------------------------------
# hashPubKeysToExponents (bgls/blsHAE.go:80-93) draws the aggregation exponents from
# golang.org/x/crypto/blake2b.NewXOF(16 n, nil) -- BLAKE2Xb, unkeyed.  hashlib refuses depth = 0, which the
# expansion nodes need, so the compression function is restated here (RFC 7693) and checked against hashlib on
# every parameter combination hashlib does accept (tests/test_oracle.py).
_B2_IV = (0x6a09e667f3bcc908, 0xbb67ae8584caa73b, 0x3c6ef372fe94f82b, 0xa54ff53a5f1d36f1,
          0x510e527fade682d1, 0x9b05688c2b3e6c1f, 0x1f83d9abfb41bd6b, 0x5be0cd19137e2179)
_B2_SIGMA = ((0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15), (14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3),
             (11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4), (7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8),
             (9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13), (2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9),
             (12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11), (13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10),
             (6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5), (10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0))
_M64 = (1 << 64) - 1


def _b2_compress(h, block, t, last):
    m = [int.from_bytes(block[8 * i:8 * i + 8], "little") for i in range(16)]
    v = list(h) + list(_B2_IV)
    v[12] ^= t & _M64
    v[13] ^= t >> 64
    if last:
        v[14] ^= _M64

    def g(a, b, c, d, x, y):
        v[a] = (v[a] + v[b] + x) & _M64
        v[d] = _ror64(v[d] ^ v[a], 32)
        v[c] = (v[c] + v[d]) & _M64
        v[b] = _ror64(v[b] ^ v[c], 24)
        v[a] = (v[a] + v[b] + y) & _M64
        v[d] = _ror64(v[d] ^ v[a], 16)
        v[c] = (v[c] + v[d]) & _M64
        v[b] = _ror64(v[b] ^ v[c], 63)

    for r in range(12):
        s = _B2_SIGMA[r % 10]
        g(0, 4, 8, 12, m[s[0]], m[s[1]]); g(1, 5, 9, 13, m[s[2]], m[s[3]])
        g(2, 6, 10, 14, m[s[4]], m[s[5]]); g(3, 7, 11, 15, m[s[6]], m[s[7]])
        g(0, 5, 10, 15, m[s[8]], m[s[9]]); g(1, 6, 11, 12, m[s[10]], m[s[11]])
        g(2, 7, 8, 13, m[s[12]], m[s[13]]); g(3, 4, 9, 14, m[s[14]], m[s[15]])
    return [h[i] ^ v[i] ^ v[i + 8] for i in range(8)]


def _ror64(x, n):
    return ((x >> n) | (x << (64 - n))) & _M64


def blake2b_param(data: bytes, digest_size=64, fanout=1, depth=1, leaf_size=0, node_offset=0, xof_length=0,
                  node_depth=0, inner_size=0) -> bytes:
    """Unkeyed BLAKE2b with the full parameter block (RFC 7693 2.5 / BLAKE2X 2.1: the upper half of the 8-byte
    node offset carries the XOF length)."""
    p = bytes([digest_size, 0, fanout, depth]) + leaf_size.to_bytes(4, "little") + node_offset.to_bytes(4, "little") + \
        xof_length.to_bytes(4, "little") + bytes([node_depth, inner_size]) + bytes(14) + bytes(32)
    h = [_B2_IV[i] ^ int.from_bytes(p[8 * i:8 * i + 8], "little") for i in range(8)]
    n = len(data)
    nblk = max(1, (n + 127) // 128)
    for b in range(nblk):
        blk = data[128 * b:128 * b + 128]
        last = b + 1 == nblk
        h = _b2_compress(h, blk + bytes(128 - len(blk)), n if last else 128 * (b + 1), last)
    return b"".join(x.to_bytes(8, "little") for x in h)[:digest_size]


def blake2xb(data: bytes, out_len: int) -> bytes:
    """BLAKE2Xb, unkeyed, out_len < 2^32 - 1 known in advance: root H0 = BLAKE2b-512(data) with the XOF length in the
    parameter block; output block i = BLAKE2b(H0) with digest = min(64, remaining), fanout = depth = 0,
    leaf length 64, node offset i, inner length 64 (x/crypto/blake2b/blake2x.go as called by blsHAE.go:81)."""
    root = hashlib.blake2b(data, digest_size=64, node_offset=out_len << 32).digest()
    out = []
    got = 0
    i = 0
    while got < out_len:
        take = min(64, out_len - got)
        out.append(blake2b_param(root, digest_size=take, fanout=0, depth=0, leaf_size=64, node_offset=i, xof_length=out_len,
                                 node_depth=0, inner_size=64))
        got += take
        i += 1
    return b"".join(out)
