// Device hash primitives for hash-to-G1.
//  keccak256_legacy : Keccak-256 with the original 0x01 domain padding = go-ethereum's
//                     NewKeccak256, i.e. EthereumSum256 (curves/altbn128.go:517-522).
//  blake2b512       : unkeyed BLAKE2b-512 = golang.org/x/crypto/blake2b New512(nil)
//                     (curves/bls12_381.go:362-367,397-400).
// Both take the input as (prefix bytes || message || suffix bytes) through a byte getter so the
// counter byte / "G1_x" tag never has to be materialised next to the message.
#pragma once
#include "fp.hpp"

namespace bgls {

struct ByteSrc {       // logical byte string: pre[0..npre) || msg[0..len) || suf[0..nsuf)
  const uint8_t* msg;
  size_t len;
  uint8_t pre[1];
  int npre;
  uint8_t suf[4];
  int nsuf;
  BGLS_HD size_t total() const { return (size_t)npre + len + (size_t)nsuf; }
  BGLS_HD u32 at(size_t pos) const {
    if (pos < (size_t)npre) return pre[0];
    pos -= npre;
    if (pos < len) return msg[pos];
    pos -= len;
    return suf[pos & 3];
  }
  // eight logical bytes from pos on as a little-endian word; bytes at and beyond `total` read as zero.  Round 6: a word that lies wholly inside
  // the message is ONE 8-byte load (unaligned addresses are served by the hardware) -- the absorb loops used to fetch every byte on its own,
  // 65 dependent single-byte loads per 64-byte message and hash, and the try-and-increment rounds ran at the pace of those loads, not of
  // their arithmetic.  Words that touch the prefix, the suffix or the end of the input take the byte path.
  BGLS_HD u64 le64(size_t pos, size_t total) const {
    if (pos >= (size_t)npre && pos + 8 <= (size_t)npre + len) {
      u64 w;
      __builtin_memcpy(&w, msg + (pos - (size_t)npre), 8);
      return w;
    }
    u64 w = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const size_t q = pos + b;
      const u32 byte = q < total ? at(q) : 0u;
      w |= (u64)byte << (8 * b);
    }
    return w;
  }
};

BGLS_HD u64 rotl64(u64 x, int n) { return (x << n) | (x >> (64 - n)); }
BGLS_HD u64 rotr64(u64 x, int n) { return (x >> n) | (x << (64 - n)); }

#if defined(__HIP_DEVICE_COMPILE__)
#define BGLS_TABLE __device__ __constant__ const
#else
#define BGLS_TABLE static const
#endif

BGLS_TABLE u64 KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
    0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
    0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

// (round 6: inlined into keccak256_legacy -- as a call it took its 25 lanes by reference, i.e. through the stack, and every round read and
// wrote them in scratch memory)
BGLS_HD void keccak_f1600(u64 (&st)[25]) {
  constexpr int rotc[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  constexpr int piln[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  for (int round = 0; round < 24; ++round) {
    u64 bc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      u64 t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
#pragma unroll
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    u64 t = st[1];
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      int j = piln[i];
      u64 b0 = st[j];
      st[j] = rotl64(t, rotc[i]);
      t = b0;
    }
#pragma unroll
    for (int j = 0; j < 25; j += 5) {
#pragma unroll
      for (int i = 0; i < 5; ++i) bc[i] = st[j + i];
#pragma unroll
      for (int i = 0; i < 5; ++i) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= KECCAK_RC[round];
  }
}

// out = 8 x u32, out[0] = digest bytes 0..3 as a BIG-endian word ... (so out[] read as a
// big-endian 256-bit integer has out[0] most significant)
inline BGLS_FN void keccak256_legacy(const ByteSrc& src, u32 (&out_be)[8]) {
  constexpr size_t RATE = 136;
  u64 st[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) st[i] = 0;
  const size_t total = src.total();
  const size_t nblk = total / RATE + 1;
  for (size_t blk = 0; blk < nblk; ++blk) {
    const bool last = (blk + 1 == nblk);
#pragma unroll
    for (int i = 0; i < 17; ++i) {
      const size_t pos = blk * RATE + 8 * i;
      u64 w = src.le64(pos, total);
      if (total >= pos && total < pos + 8) w |= (u64)0x01u << (8 * (total - pos));      // the legacy domain byte right behind the input
      if (last && i == 16) w ^= (u64)0x80u << 56;
      st[i] ^= w;
    }
    keccak_f1600(st);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    u32 lo = (u32)st[i], hi = (u32)(st[i] >> 32);
    out_be[2 * i] = __builtin_bswap32(lo);
    out_be[2 * i + 1] = __builtin_bswap32(hi);
  }
}

BGLS_TABLE u64 BLAKE2B_IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull,
                                0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full,
                                0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
BGLS_TABLE uint8_t BLAKE2B_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

BGLS_HD void blake2b_g(u64 (&v)[16], int a, int b, int c, int d, u64 x, u64 y) {
  v[a] = v[a] + v[b] + x;
  v[d] = rotr64(v[d] ^ v[a], 32);
  v[c] = v[c] + v[d];
  v[b] = rotr64(v[b] ^ v[c], 24);
  v[a] = v[a] + v[b] + y;
  v[d] = rotr64(v[d] ^ v[a], 16);
  v[c] = v[c] + v[d];
  v[b] = rotr64(v[b] ^ v[c], 63);
}

// one compression (RFC 7693 3.2): h <- F(h, m, t, last)
BGLS_HD void blake2b_compress(u64 (&h)[8], const u64 (&m)[16], u64 t, bool last) {
  u64 v[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = h[i];
    v[i + 8] = BLAKE2B_IV[i];
  }
  v[12] ^= t;
  if (last) v[14] = ~v[14];
  for (int r = 0; r < 12; ++r) {
    const uint8_t* s = BLAKE2B_SIGMA[r];
    blake2b_g(v, 0, 4, 8, 12, m[s[0]], m[s[1]]);
    blake2b_g(v, 1, 5, 9, 13, m[s[2]], m[s[3]]);
    blake2b_g(v, 2, 6, 10, 14, m[s[4]], m[s[5]]);
    blake2b_g(v, 3, 7, 11, 15, m[s[6]], m[s[7]]);
    blake2b_g(v, 0, 5, 10, 15, m[s[8]], m[s[9]]);
    blake2b_g(v, 1, 6, 11, 12, m[s[10]], m[s[11]]);
    blake2b_g(v, 2, 7, 8, 13, m[s[12]], m[s[13]]);
    blake2b_g(v, 3, 4, 9, 14, m[s[14]], m[s[15]]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}

// out_be[0..16): the 64 digest bytes as big-endian u32 words, out_be[0] most significant
inline BGLS_FN void blake2b512(const ByteSrc& src, u32 (&out_be)[16]) {
  u64 h[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = BLAKE2B_IV[i];
  h[0] ^= 0x01010040ull;
  const size_t total = src.total();
  const size_t nblk = total == 0 ? 1 : (total + 127) / 128;
  for (size_t blk = 0; blk < nblk; ++blk) {
    u64 m[16];
    for (int i = 0; i < 16; ++i) m[i] = src.le64(blk * 128 + 8 * (size_t)i, total);
    const bool last = (blk + 1 == nblk);
    blake2b_compress(h, m, last ? (u64)total : (u64)(blk + 1) * 128, last);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u32 lo = (u32)h[i], hi = (u32)(h[i] >> 32);
    out_be[2 * i] = __builtin_bswap32(lo);
    out_be[2 * i + 1] = __builtin_bswap32(hi);
  }
}

// ---- BLAKE2Xb, unkeyed (golang.org/x/crypto/blake2b NewXOF(size, nil), the hash of bgls/blsHAE.go:80-93) ----
// Root: BLAKE2b-512 of the input whose parameter block carries the XOF length in bytes 12..15 (h[1] ^= len << 32).
// The root is a sequential chain over the whole input, so it is computed where the bytes are produced (host side of
// the boundary, blake2xb_root); the expansion nodes are independent and run one per lane (k_blake2x_expand).
BGLS_HD void blake2xb_root_init(u64 (&h)[8], u32 xof_len) {
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = BLAKE2B_IV[i];
  h[0] ^= 0x01010040ull;
  h[1] ^= (u64)xof_len << 32;
}
// node i: BLAKE2b(root; digest = take, fanout 0, depth 0, leaf length 64, node offset i, XOF length, inner length 64)
inline BGLS_FN void blake2xb_node(const u64 (&root)[8], u32 i, u32 xof_len, u32 take, u64 (&out)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) out[k] = BLAKE2B_IV[k];
  out[0] ^= (u64)take | (64ull << 32);
  out[1] ^= (u64)i | ((u64)xof_len << 32);
  out[2] ^= 64ull << 8;
  u64 m[16];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    m[k] = root[k];
    m[k + 8] = 0;
  }
  blake2b_compress(out, m, 64, true);
}

}  // namespace bgls
