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
