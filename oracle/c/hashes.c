/* CPU oracle hash primitives -- TEST INFRASTRUCTURE ONLY.
 * keccak256_legacy: Keccak-256, original 0x01 padding (EthereumSum256, curves/altbn128.go:517-522).
 * blake2b512: unkeyed BLAKE2b-512 (curves/bls12_381.go:362-367,397-400), RFC 7693. */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

static uint64_t rol(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
static const uint64_t RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull,
    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull,
    0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull,
    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
static const int ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};

static void keccak_f(uint64_t A[5][5]) {          /* A[x][y] */
  for (int r = 0; r < 24; r++) {
    uint64_t C[5], D[5], B[5][5];
    for (int x = 0; x < 5; x++) C[x] = A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4];
    for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rol(C[(x + 1) % 5], 1);
    for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) A[x][y] ^= D[x];
    for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) B[y][(2 * x + 3 * y) % 5] = rol(A[x][y], ROT[x][y]);
    for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) A[x][y] = B[x][y] ^ (~B[(x + 1) % 5][y] & B[(x + 2) % 5][y]);
    A[0][0] ^= RC[r];
  }
}

void keccak256_legacy(const uint8_t* in, size_t len, uint8_t out[32]) {
  const size_t rate = 136;
  uint64_t A[5][5]; memset(A, 0, sizeof A);
  uint8_t blk[136];
  size_t off = 0;
  for (;;) {
    size_t take = len - off < rate ? len - off : rate;
    memset(blk, 0, rate); memcpy(blk, in + off, take);
    int last = take < rate;
    if (last) { blk[take] ^= 0x01; blk[rate - 1] ^= 0x80; }
    for (size_t i = 0; i < rate / 8; i++) { uint64_t w = 0; for (int k = 7; k >= 0; k--) w = (w << 8) | blk[8 * i + k]; A[i % 5][i / 5] ^= w; }
    keccak_f(A);
    off += take;
    if (last) break;
  }
  for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(A[i % 5][i / 5] >> (8 * k));
}

static const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                               0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
static const uint8_t SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
static uint64_t ror(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
#define G(a, b, c, d, x, y) do { v[a] += v[b] + (x); v[d] = ror(v[d] ^ v[a], 32); v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 24); \
  v[a] += v[b] + (y); v[d] = ror(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 63); } while (0)

/* BLAKE2b, unkeyed, explicit parameter words p[0..2] XORed into h[0..2] (RFC 7693 2.5); digest bytes = p0 & 0xff */
static void blake2b_core(const uint8_t* in, size_t len, uint64_t p0, uint64_t p1, uint64_t p2, uint8_t* out, size_t outlen) {
  uint64_t h[8]; memcpy(h, IV, sizeof h); h[0] ^= p0; h[1] ^= p1; h[2] ^= p2;
  size_t nblk = len == 0 ? 1 : (len + 127) / 128;
  for (size_t b = 0; b < nblk; b++) {
    uint8_t blk[128]; memset(blk, 0, 128);
    size_t take = len - b * 128 < 128 ? len - b * 128 : 128; memcpy(blk, in + b * 128, take);
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; i++) { uint64_t w = 0; for (int k = 7; k >= 0; k--) w = (w << 8) | blk[8 * i + k]; m[i] = w; }
    int last = b + 1 == nblk;
    uint64_t t = last ? (uint64_t)len : (uint64_t)(b + 1) * 128;
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
    v[12] ^= t; if (last) v[14] = ~v[14];
    for (int r = 0; r < 12; r++) { const uint8_t* s = SIGMA[r];
      G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]]); G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
      G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]); G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]); }
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
  }
  uint8_t full[64];
  for (int i = 0; i < 8; i++) for (int k = 0; k < 8; k++) full[8 * i + k] = (uint8_t)(h[i] >> (8 * k));
  memcpy(out, full, outlen);
}

void blake2b512(const uint8_t* in, size_t len, uint8_t out[64]) { blake2b_core(in, len, 0x01010040ull, 0, 0, out, 64); }

/* BLAKE2Xb, unkeyed (golang.org/x/crypto/blake2b NewXOF(out_len, nil), called by bgls/blsHAE.go:81): root digest with the
 * XOF length in parameter bytes 12..15, then block i = BLAKE2b(root; digest min(64, rest), fanout 0, depth 0, leaf 64,
 * node offset i, XOF length, inner length 64). */
int oracle_blake2xb(const uint8_t* in, size_t len, uint32_t out_len, uint8_t* out) {
  uint8_t root[64];
  blake2b_core(in, len, 0x01010040ull, (uint64_t)out_len << 32, 0, root, 64);
  uint32_t got = 0, i = 0;
  while (got < out_len) {
    uint32_t take = out_len - got < 64 ? out_len - got : 64;
    blake2b_core(root, 64, (uint64_t)take | (64ull << 32), (uint64_t)i | ((uint64_t)out_len << 32), 64ull << 8, out + got, take);
    got += take; i++;
  }
  return 0;
}
