/*
 * CPU oracle, C restatement of the aggregate-verify hot path -- TEST INFRASTRUCTURE ONLY.
 * Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
 * (bgls_amd/libbgls_hip.so) never links, loads or calls it.
 *
 * Compiled twice (consts_bn.h: alt-bn128, 4 x u64; consts_bls.h: BLS12-381, 6 x u64) with
 * -DPFX=bn / -DPFX=bls.  64-bit limbs + unsigned __int128 CIOS: deliberately a different
 * arithmetic formulation from the HIP kernels (32-bit limbs, split product / lazy reduction).
 *
 * What it restates (reference file:line):
 *   verifyAggSig                     bgls/bgls.go:94-119
 *   containsDuplicateMessage         bgls/bgls.go:139-150
 *   verifyMultiSignature             bgls/bgls.go:89-92, VerifySingleSignature :59-70
 *   AggregatePoints                  curves/curve.go:73-121
 *   concurrentPairingProduct         curves/curve.go:125-170 (one task per pairing; `faithful`
 *                                    mode runs a full final exponentiation per pair like the
 *                                    reference, default mode shares one -- same GT value)
 *   tryAndIncrementEvm               curves/hash.go:53-77 (+ altbn128.go:409-414,494-522)
 *   bls12 HashToG1 / FouqueTibouchi  curves/bls12_381.go:349-400, sw(): curves/hash.go:86-190,254-265
 * The pairing itself (upstream bn256/cloudflare, dis2/bls12: absent) follows
 * oracle/pyref/pairing.py, which is pinned against the textbook definition.
 * PARITY: hash-to-G1 pinned by the reference's 22 KATs; Verify* booleans pinned by bilinearity;
 * GT bytes UNPINNED against the upstream Go libraries (no vector exists in the reference).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>

#ifdef CURVE_BN
#include "consts_bn.h"
#else
#include "consts_bls.h"
#endif

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(PFX, name)

typedef unsigned __int128 u128;
typedef struct { uint64_t v[NL]; } fp;
typedef struct { fp c0, c1; } fp2;
typedef struct { fp2 a0, a1, a2; } fp6;
typedef struct { fp6 g, h; } fp12;

void keccak256_legacy(const uint8_t* in, size_t len, uint8_t out[32]);
void blake2b512(const uint8_t* in, size_t len, uint8_t out[64]);

/* ------------------------------------------------------------------ Fp */
static int fp_is_zero(const fp* a) { uint64_t o = 0; for (int i = 0; i < NL; i++) o |= a->v[i]; return o == 0; }
static int fp_eq(const fp* a, const fp* b) { uint64_t o = 0; for (int i = 0; i < NL; i++) o |= a->v[i] ^ b->v[i]; return o == 0; }
static int raw_geq(const uint64_t* a, const uint64_t* b) {
  for (int i = NL - 1; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
  return 1;
}
static uint64_t raw_add(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  u128 c = 0; for (int i = 0; i < NL; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; } return (uint64_t)c;
}
static uint64_t raw_sub(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t bw = 0; for (int i = 0; i < NL; i++) { u128 d = (u128)a[i] - b[i] - bw; r[i] = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1; } return bw;
}
static void fp_add(fp* r, const fp* a, const fp* b) {
  uint64_t t[NL]; uint64_t c = raw_add(t, a->v, b->v);
  if (c || raw_geq(t, PMOD)) raw_sub(t, t, PMOD);
  memcpy(r->v, t, sizeof t);
}
static void fp_sub(fp* r, const fp* a, const fp* b) {
  uint64_t t[NL]; if (raw_sub(t, a->v, b->v)) raw_add(t, t, PMOD);
  memcpy(r->v, t, sizeof t);
}
static void fp_neg(fp* r, const fp* a) { if (fp_is_zero(a)) *r = *a; else raw_sub(r->v, PMOD, a->v); }
static void fp_mul(fp* r, const fp* a, const fp* b) {           /* CIOS */
  uint64_t t[NL + 2]; memset(t, 0, sizeof t);
  for (int i = 0; i < NL; i++) {
    u128 c = 0;
    for (int j = 0; j < NL; j++) { c += (u128)a->v[j] * b->v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[NL]; t[NL] = (uint64_t)c; t[NL + 1] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * N0INV;
    c = (u128)m * PMOD[0] + t[0]; c >>= 64;
    for (int j = 1; j < NL; j++) { c += (u128)m * PMOD[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[NL]; t[NL - 1] = (uint64_t)c; t[NL] = t[NL + 1] + (uint64_t)(c >> 64);
  }
  if (t[NL] || raw_geq(t, PMOD)) raw_sub(t, t, PMOD);
  memcpy(r->v, t, sizeof(uint64_t) * NL);
}
static void fp_sqr(fp* r, const fp* a) { fp_mul(r, a, a); }
static void fp_set(fp* r, const uint64_t* c) { memcpy(r->v, c, sizeof(uint64_t) * NL); }
static void fp_to_mont(fp* r, const fp* a) { fp t; fp_set(&t, R2); fp_mul(r, a, &t); }
static void fp_from_mont(fp* r, const fp* a) { fp one; memset(&one, 0, sizeof one); one.v[0] = 1; fp_mul(r, a, &one); }
static void fp_pow(fp* r, const fp* a, const uint64_t* e, int nlimbs) {
  fp acc; fp_set(&acc, ONE);
  for (int i = nlimbs * 64 - 1; i >= 0; i--) { fp_sqr(&acc, &acc); if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(&acc, &acc, a); }
  *r = acc;
}
static void exp_p_minus(uint64_t* e, uint64_t k) { memcpy(e, PMOD, sizeof(uint64_t) * NL); e[0] -= k; }  /* p = ...b (no borrow for k<=3) */
static void fp_inv(fp* r, const fp* a) { uint64_t e[NL]; exp_p_minus(e, 2); fp_pow(r, a, e, NL); }
static void exp_shr(uint64_t* e, int s) { for (int i = 0; i < NL; i++) e[i] = (e[i] >> s) | (i + 1 < NL ? e[i + 1] << (64 - s) : 0); }
/* calcQuadRes: a^((q+1)/4), curves/hash.go:178-190 */
static void fp_sqrt_cand(fp* r, const fp* a) { uint64_t e[NL]; memcpy(e, PMOD, sizeof e); exp_shr(e, 2); e[0] += 1; fp_pow(r, a, e, NL); }
/* isQuadRes: Euler criterion, 0 counts as a square, curves/hash.go:254-265 */
static int fp_is_qr(const fp* a) {
  if (fp_is_zero(a)) return 1;
  uint64_t e[NL]; memcpy(e, PMOD, sizeof e); exp_shr(e, 1);
  fp t, one; fp_pow(&t, a, e, NL); fp_set(&one, ONE); return fp_eq(&t, &one);
}
static void fp_from_be(fp* r, const uint8_t* b) {
  for (int i = 0; i < NL; i++) { uint64_t w = 0; for (int k = 0; k < 8; k++) w = (w << 8) | b[8 * (NL - 1 - i) + k]; r->v[i] = w; }
}
static void fp_to_be(uint8_t* b, const fp* a) {
  for (int i = 0; i < NL; i++) for (int k = 0; k < 8; k++) b[8 * (NL - 1 - i) + k] = (uint8_t)(a->v[i] >> (56 - 8 * k));
}
static int fp_read(fp* r, const uint8_t* b) { fp t; fp_from_be(&t, b); if (raw_geq(t.v, PMOD)) return 0; fp_to_mont(r, &t); return 1; }
static void fp_write(uint8_t* b, const fp* a) { fp t; fp_from_mont(&t, a); fp_to_be(b, &t); }

/* ------------------------------------------------------------------ Fp2 (i^2 = -1, curves/complexNum.go) */
static void f2_add(fp2* r, const fp2* a, const fp2* b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
static void f2_sub(fp2* r, const fp2* a, const fp2* b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
static void f2_neg(fp2* r, const fp2* a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
static void f2_conj(fp2* r, const fp2* a) { r->c0 = a->c0; fp_neg(&r->c1, &a->c1); }
static void f2_mul(fp2* r, const fp2* a, const fp2* b) {      /* Karatsuba, 3 Fp products */
  fp t0, t1, sa, sb, m; fp_mul(&t0, &a->c0, &b->c0); fp_mul(&t1, &a->c1, &b->c1);
  fp_add(&sa, &a->c0, &a->c1); fp_add(&sb, &b->c0, &b->c1); fp_mul(&m, &sa, &sb);
  fp_sub(&m, &m, &t0); fp_sub(&m, &m, &t1); fp_sub(&r->c0, &t0, &t1); r->c1 = m;
}
static void f2_sqr(fp2* r, const fp2* a) {                       /* (a0+a1)(a0-a1), 2 a0 a1 */
  fp s, d, m; fp_add(&s, &a->c0, &a->c1); fp_sub(&d, &a->c0, &a->c1); fp_mul(&m, &a->c0, &a->c1);
  fp_mul(&r->c0, &s, &d); fp_add(&r->c1, &m, &m);
}
static void f2_muls(fp2* r, const fp2* a, const fp* s) { fp_mul(&r->c0, &a->c0, s); fp_mul(&r->c1, &a->c1, s); }
static void f2_small(fp2* r, const fp2* a, int k) { fp2 acc = *a; for (int i = 1; i < k; i++) f2_add(&acc, &acc, a); *r = acc; }
static void f2_mulxi(fp2* r, const fp2* a) {
  fp2 t; f2_small(&t, a, XI_RE); fp2 o; fp_sub(&o.c0, &t.c0, &a->c1); fp_add(&o.c1, &t.c1, &a->c0); *r = o;
}
static void f2_inv(fp2* r, const fp2* a) {
  fp n, t; fp_sqr(&n, &a->c0); fp_sqr(&t, &a->c1); fp_add(&n, &n, &t); fp_inv(&n, &n);
  fp_mul(&r->c0, &a->c0, &n); fp_mul(&t, &a->c1, &n); fp_neg(&r->c1, &t);
}
static int f2_is_zero(const fp2* a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static int f2_eq(const fp2* a, const fp2* b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static void f2_one(fp2* r) { fp_set(&r->c0, ONE); memset(&r->c1, 0, sizeof(fp)); }
static void f2_load(fp2* r, const uint64_t* c) { fp_set(&r->c0, c); fp_set(&r->c1, c + NL); }

/* ------------------------------------------------------------------ Fp6 / Fp12 (schoolbook; follows oracle/pyref/tower.py) */
static void f6_add(fp6* r, const fp6* a, const fp6* b) { f2_add(&r->a0, &a->a0, &b->a0); f2_add(&r->a1, &a->a1, &b->a1); f2_add(&r->a2, &a->a2, &b->a2); }
static void f6_sub(fp6* r, const fp6* a, const fp6* b) { f2_sub(&r->a0, &a->a0, &b->a0); f2_sub(&r->a1, &a->a1, &b->a1); f2_sub(&r->a2, &a->a2, &b->a2); }
static void f6_neg(fp6* r, const fp6* a) { f2_neg(&r->a0, &a->a0); f2_neg(&r->a1, &a->a1); f2_neg(&r->a2, &a->a2); }
static void f6_mul(fp6* r, const fp6* a, const fp6* b) {      /* Karatsuba, 6 Fp2 products */
  fp2 t0, t1, t2, s, u, c0, c1, c2;
  f2_mul(&t0, &a->a0, &b->a0); f2_mul(&t1, &a->a1, &b->a1); f2_mul(&t2, &a->a2, &b->a2);
  f2_add(&s, &a->a1, &a->a2); f2_add(&u, &b->a1, &b->a2); f2_mul(&c0, &s, &u); f2_sub(&c0, &c0, &t1); f2_sub(&c0, &c0, &t2); f2_mulxi(&c0, &c0); f2_add(&c0, &c0, &t0);
  f2_add(&s, &a->a0, &a->a1); f2_add(&u, &b->a0, &b->a1); f2_mul(&c1, &s, &u); f2_sub(&c1, &c1, &t0); f2_sub(&c1, &c1, &t1); f2_mulxi(&s, &t2); f2_add(&c1, &c1, &s);
  f2_add(&s, &a->a0, &a->a2); f2_add(&u, &b->a0, &b->a2); f2_mul(&c2, &s, &u); f2_sub(&c2, &c2, &t0); f2_sub(&c2, &c2, &t2); f2_add(&c2, &c2, &t1);
  r->a0 = c0; r->a1 = c1; r->a2 = c2;
}
static void f6_mulv(fp6* r, const fp6* a) { fp6 o; f2_mulxi(&o.a0, &a->a2); o.a1 = a->a0; o.a2 = a->a1; *r = o; }
static void f6_inv(fp6* r, const fp6* a) {
  fp2 t0, t1, t2, u, d;
  f2_sqr(&t0, &a->a0); f2_mul(&u, &a->a1, &a->a2); f2_mulxi(&u, &u); f2_sub(&t0, &t0, &u);
  f2_sqr(&t1, &a->a2); f2_mulxi(&t1, &t1); f2_mul(&u, &a->a0, &a->a1); f2_sub(&t1, &t1, &u);
  f2_sqr(&t2, &a->a1); f2_mul(&u, &a->a0, &a->a2); f2_sub(&t2, &t2, &u);
  f2_mul(&d, &a->a2, &t1); f2_mul(&u, &a->a1, &t2); f2_add(&d, &d, &u); f2_mulxi(&d, &d); f2_mul(&u, &a->a0, &t0); f2_add(&d, &d, &u);
  f2_inv(&d, &d);
  f2_mul(&r->a0, &t0, &d); f2_mul(&r->a1, &t1, &d); f2_mul(&r->a2, &t2, &d);
}
static void f12_one(fp12* r) { memset(r, 0, sizeof *r); f2_one(&r->g.a0); }
static void f12_mul(fp12* r, const fp12* a, const fp12* b) {  /* Karatsuba over Fp6 */
  fp6 gg, hh, s, u, c1, t;
  f6_mul(&gg, &a->g, &b->g); f6_mul(&hh, &a->h, &b->h);
  f6_add(&s, &a->g, &a->h); f6_add(&u, &b->g, &b->h); f6_mul(&c1, &s, &u); f6_sub(&c1, &c1, &gg); f6_sub(&c1, &c1, &hh);
  f6_mulv(&t, &hh); f6_add(&r->g, &gg, &t); r->h = c1;
}
static void f12_sqr(fp12* r, const fp12* a) {                  /* complex squaring, 2 Fp6 products */
  fp6 gh, s, u, t;
  f6_mul(&gh, &a->g, &a->h); f6_add(&s, &a->g, &a->h); f6_mulv(&u, &a->h); f6_add(&u, &u, &a->g);
  f6_mul(&t, &s, &u); f6_sub(&t, &t, &gh); f6_mulv(&u, &gh); f6_sub(&t, &t, &u);
  r->g = t; f6_add(&r->h, &gh, &gh);
}
static void f12_conj(fp12* r, const fp12* a) { r->g = a->g; f6_neg(&r->h, &a->h); }
static void f12_inv(fp12* r, const fp12* a) {
  fp6 d, t; f6_mul(&d, &a->g, &a->g); f6_mul(&t, &a->h, &a->h); f6_mulv(&t, &t); f6_sub(&d, &d, &t); f6_inv(&d, &d);
  fp12 o; f6_mul(&o.g, &a->g, &d); f6_mul(&t, &a->h, &d); f6_neg(&o.h, &t); *r = o;
}
static void f12_pow64(fp12* r, const fp12* a, uint64_t e) {
  fp12 acc; f12_one(&acc);
  for (int i = 63; i >= 0; i--) { f12_sqr(&acc, &acc); if ((e >> i) & 1) f12_mul(&acc, &acc, a); }
  *r = acc;
}
/* Granger-Scott squaring: same value as f12_sqr for elements of the cyclotomic subgroup (everything after the easy part of
 * the final exponentiation), 9 Fp2 squarings instead of 12 Fp2 products.  Used by the final exponentiation only. */
static void fp4_sqr(fp2* c0, fp2* c1, const fp2* a, const fp2* b) {
  fp2 t0, t1, s; f2_sqr(&t0, a); f2_sqr(&t1, b);
  f2_add(&s, a, b); f2_sqr(&s, &s); f2_sub(&s, &s, &t0); f2_sub(c1, &s, &t1);
  f2_mulxi(&s, &t1); f2_add(c0, &s, &t0);
}
static void f12_cyclo_sqr(fp12* r, const fp12* f) {
  fp2 z0 = f->g.a0, z4 = f->g.a1, z3 = f->g.a2, z2 = f->h.a0, z1 = f->h.a1, z5 = f->h.a2, t0, t1, t2, t3, u;
  fp4_sqr(&t0, &t1, &z0, &z1);
  f2_sub(&u, &t0, &z0); f2_add(&u, &u, &u); f2_add(&z0, &u, &t0);
  f2_add(&u, &t1, &z1); f2_add(&u, &u, &u); f2_add(&z1, &u, &t1);
  fp4_sqr(&t0, &t1, &z2, &z3);
  fp4_sqr(&t2, &t3, &z4, &z5);
  f2_sub(&u, &t0, &z4); f2_add(&u, &u, &u); f2_add(&z4, &u, &t0);
  f2_add(&u, &t1, &z5); f2_add(&u, &u, &u); f2_add(&z5, &u, &t1);
  f2_mulxi(&t0, &t3);
  f2_add(&u, &t0, &z2); f2_add(&u, &u, &u); f2_add(&z2, &u, &t0);
  f2_sub(&u, &t2, &z3); f2_add(&u, &u, &u); f2_add(&z3, &u, &t2);
  r->g.a0 = z0; r->g.a1 = z4; r->g.a2 = z3; r->h.a0 = z2; r->h.a1 = z1; r->h.a2 = z5;
}
static void f12_pow64_cyc(fp12* r, const fp12* a, uint64_t e) {     /* a in the cyclotomic subgroup */
  fp12 acc; f12_one(&acc);
  for (int i = 63; i >= 0; i--) { f12_cyclo_sqr(&acc, &acc); if ((e >> i) & 1) f12_mul(&acc, &acc, a); }
  *r = acc;
}
static fp2* w_coef(fp12* a, int k) { fp6* s = (k & 1) ? &a->h : &a->g; return k / 2 == 0 ? &s->a0 : k / 2 == 1 ? &s->a1 : &s->a2; }
static void f12_frob(fp12* r, const fp12* a, int j) {
  fp12 in = *a, o;
  for (int k = 0; k < 6; k++) {
    fp2 x = *w_coef(&in, k), g;
    if (j & 1) f2_conj(&x, &x);
    f2_load(&g, GAMMA + ((size_t)(j - 1) * 6 + k) * 2 * NL);
    f2_mul(w_coef(&o, k), &x, &g);
  }
  *r = o;
}
static int f12_is_one(const fp12* a) { fp12 o; f12_one(&o); return memcmp(a, &o, sizeof o) == 0; }
/* a * (b0 + b1 v) in Fp6: 5 Fp2 products */
static void f6_mul_01(fp6* r, const fp6* a, const fp2* b0, const fp2* b1) {
  fp2 v0, v1, s, u, c0, c1, c2;
  f2_mul(&v0, &a->a0, b0); f2_mul(&v1, &a->a1, b1);
  f2_mul(&c0, &a->a2, b1); f2_mulxi(&c0, &c0); f2_add(&c0, &c0, &v0);
  f2_add(&s, &a->a0, &a->a1); f2_add(&u, b0, b1); f2_mul(&c1, &s, &u); f2_sub(&c1, &c1, &v0); f2_sub(&c1, &c1, &v1);
  f2_mul(&c2, &a->a2, b0); f2_add(&c2, &c2, &v1);
  r->a0 = c0; r->a1 = c1; r->a2 = c2;
}
static void f6_mul_0(fp6* r, const fp6* a, const fp2* b0) { f2_mul(&r->a0, &a->a0, b0); f2_mul(&r->a1, &a->a1, b0); f2_mul(&r->a2, &a->a2, b0); }
/* f * (e[0] w^pos[0] + e[1] w^pos[1] + e[2] w^pos[2]) for the two line shapes (pos = {0,1,3}: D-type, {0,2,3}: M-type): the
 * sparse form of f12_mul (same value), 13 Fp2 products instead of 18.  With w^2 = v:
 *   D: l = e0 + (e1 + e3 v) w          M: l = (e0 + e2 v) + (e3 v) w */
static void f12_mul_sparse(fp12* f, const fp2* e, const int* pos) {
  fp6 t0, t1, t2, s; fp2 z, u;
  memset(&z, 0, sizeof z);
  if (pos[1] == 1) {
    f6_mul_0(&t0, &f->g, &e[0]);                          /* g l0 */
    f6_mul_01(&t1, &f->h, &e[1], &e[2]);                  /* h l1 */
    f6_add(&s, &f->g, &f->h); f2_add(&u, &e[0], &e[1]);
    f6_mul_01(&t2, &s, &u, &e[2]);                        /* (g + h)(l0 + l1) */
  } else {
    f6_mul_01(&t0, &f->g, &e[0], &e[1]);                  /* g l0 */
    f6_mul_0(&t1, &f->h, &e[2]); f6_mulv(&t1, &t1);       /* h l1, l1 = e3 v */
    f6_add(&s, &f->g, &f->h); f2_add(&u, &e[1], &e[2]);
    f6_mul_01(&t2, &s, &e[0], &u);                        /* (g + h)(l0 + l1) */
  }
  f6_sub(&t2, &t2, &t0); f6_sub(&t2, &t2, &t1);
  f6_mulv(&s, &t1); f6_add(&f->g, &t0, &s); f->h = t2;
}
/* GT wire format: see oracle/pyref/pairing.py gt_bytes */
static void gt_write(uint8_t* b, const fp12* a) {
  const fp6* six[2] = {&a->h, &a->g}; int o = 0;
  for (int s = 0; s < 2; s++) { const fp2* e[3] = {&six[s]->a2, &six[s]->a1, &six[s]->a0};
    for (int k = 0; k < 3; k++) { fp_write(b + o, &e[k]->c1); o += FPB; fp_write(b + o, &e[k]->c0); o += FPB; } }
}
static int gt_read(fp12* a, const uint8_t* b) {
  fp6* six[2] = {&a->h, &a->g}; int o = 0, ok = 1;
  for (int s = 0; s < 2; s++) { fp2* e[3] = {&six[s]->a2, &six[s]->a1, &six[s]->a0};
    for (int k = 0; k < 3; k++) { ok &= fp_read(&e[k]->c1, b + o); o += FPB; ok &= fp_read(&e[k]->c0, b + o); o += FPB; } }
  return ok;
}

/* ------------------------------------------------------------------ groups: affine law with inversions */
typedef struct { fp x, y; int inf; } g1a;
typedef struct { fp2 x, y; int inf; } g2a;

static void g1_add(g1a* r, const g1a* P, const g1a* Q) {
  if (P->inf) { *r = *Q; return; } if (Q->inf) { *r = *P; return; }
  fp m, t, u;
  if (fp_eq(&P->x, &Q->x)) {
    fp_add(&t, &P->y, &Q->y);
    if (fp_is_zero(&t)) { memset(r, 0, sizeof *r); r->inf = 1; return; }
    fp_sqr(&m, &P->x); fp_add(&u, &m, &m); fp_add(&m, &u, &m); fp_add(&t, &P->y, &P->y); fp_inv(&t, &t); fp_mul(&m, &m, &t);
  } else { fp_sub(&m, &Q->y, &P->y); fp_sub(&t, &Q->x, &P->x); fp_inv(&t, &t); fp_mul(&m, &m, &t); }
  g1a o; o.inf = 0; fp_sqr(&o.x, &m); fp_sub(&o.x, &o.x, &P->x); fp_sub(&o.x, &o.x, &Q->x);
  fp_sub(&t, &P->x, &o.x); fp_mul(&o.y, &m, &t); fp_sub(&o.y, &o.y, &P->y); *r = o;
}
static void g2_add(g2a* r, const g2a* P, const g2a* Q) {
  if (P->inf) { *r = *Q; return; } if (Q->inf) { *r = *P; return; }
  fp2 m, t, u;
  if (f2_eq(&P->x, &Q->x)) {
    f2_add(&t, &P->y, &Q->y);
    if (f2_is_zero(&t)) { memset(r, 0, sizeof *r); r->inf = 1; return; }
    f2_sqr(&m, &P->x); f2_add(&u, &m, &m); f2_add(&m, &u, &m); f2_add(&t, &P->y, &P->y); f2_inv(&t, &t); f2_mul(&m, &m, &t);
  } else { f2_sub(&m, &Q->y, &P->y); f2_sub(&t, &Q->x, &P->x); f2_inv(&t, &t); f2_mul(&m, &m, &t); }
  g2a o; o.inf = 0; f2_sqr(&o.x, &m); f2_sub(&o.x, &o.x, &P->x); f2_sub(&o.x, &o.x, &Q->x);
  f2_sub(&t, &P->x, &o.x); f2_mul(&o.y, &m, &t); f2_sub(&o.y, &o.y, &P->y); *r = o;
}
/* k P by double-and-add in Jacobian coordinates (a = 0), ONE inversion at the end: the point is the one the affine chord-and-
 * tangent chain gives (curves/curve.go:190-214 scales through the curve library), 30x cheaper than an inversion per step.
 * F: field type, PA: affine point type, the f_* are that field's operations. */
#define JAC_MUL(NAME, F, PA, f_add, f_sub, f_mul, f_sqr, f_inv, f_is_zero)                                                  \
  static void NAME(PA* r, const PA* P, const uint64_t* k, int nbits) {                                                      \
    F X, Y, Z, A, B, C, D, E, T, U; int inf = 1;                                                                            \
    if (P->inf) { memset(r, 0, sizeof *r); r->inf = 1; return; }                                                            \
    for (int i = nbits - 1; i >= 0; i--) {                                                                                  \
      if (!inf) {                                                                                                           \
        if (f_is_zero(&Y)) inf = 1;                                                                                         \
        else {  /* dbl-2009-l */                                                                                            \
          f_sqr(&A, &X); f_sqr(&B, &Y); f_sqr(&C, &B);                                                                      \
          f_add(&D, &X, &B); f_sqr(&D, &D); f_sub(&D, &D, &A); f_sub(&D, &D, &C); f_add(&D, &D, &D);                        \
          f_add(&E, &A, &A); f_add(&E, &E, &A);                                                                             \
          f_mul(&Z, &Y, &Z); f_add(&Z, &Z, &Z);                                                                             \
          f_sqr(&T, &E); f_sub(&T, &T, &D); f_sub(&X, &T, &D);                                                              \
          f_add(&C, &C, &C); f_add(&C, &C, &C); f_add(&C, &C, &C);                                                          \
          f_sub(&T, &D, &X); f_mul(&T, &E, &T); f_sub(&Y, &T, &C);                                                          \
        }                                                                                                                   \
      }                                                                                                                     \
      if ((k[i >> 6] >> (i & 63)) & 1) {                                                                                    \
        if (inf) { X = P->x; Y = P->y; memset(&Z, 0, sizeof Z); f_one_of(&Z); inf = 0; }                                    \
        else {  /* madd-2007-bl */                                                                                          \
          f_sqr(&A, &Z); f_mul(&B, &P->x, &A); f_mul(&C, &P->y, &Z); f_mul(&C, &C, &A);                                     \
          f_sub(&D, &B, &X); f_sub(&E, &C, &Y);                                                                             \
          if (f_is_zero(&D)) {                                                                                              \
            if (f_is_zero(&E)) {   /* same point: double (affine input, Z = 1 after this) */                                \
              PA d2; NAME##_aff_dbl(&d2, P); X = d2.x; Y = d2.y; memset(&Z, 0, sizeof Z); f_one_of(&Z); inf = d2.inf;       \
            } else inf = 1;                                                                                                 \
          } else {                                                                                                          \
            f_add(&E, &E, &E); f_sqr(&T, &D); f_add(&U, &T, &T); f_add(&U, &U, &U);       /* r = 2(S2-Y1), HH, I = 4 HH */ \
            F J, V; f_mul(&J, &D, &U); f_mul(&V, &X, &U);                                                                   \
            f_add(&Z, &Z, &D); f_sqr(&Z, &Z); f_sub(&Z, &Z, &A); f_sub(&Z, &Z, &T);                                         \
            f_sqr(&X, &E); f_sub(&X, &X, &J); f_sub(&X, &X, &V); f_sub(&X, &X, &V);                                         \
            f_sub(&V, &V, &X); f_mul(&V, &E, &V); f_mul(&J, &Y, &J); f_add(&J, &J, &J); f_sub(&Y, &V, &J);                  \
          }                                                                                                                 \
        }                                                                                                                   \
      }                                                                                                                     \
    }                                                                                                                       \
    if (inf || f_is_zero(&Z)) { memset(r, 0, sizeof *r); r->inf = 1; return; }                                              \
    f_inv(&A, &Z); f_sqr(&B, &A); f_mul(&r->x, &X, &B); f_mul(&B, &B, &A); f_mul(&r->y, &Y, &B); r->inf = 0;                \
  }
static void fp_one_of(fp* r) { fp_set(r, ONE); }
static void f2_one_of(fp2* r) { memset(r, 0, sizeof *r); fp_set(&r->c0, ONE); }
static void g1_mul_aff_dbl(g1a* r, const g1a* P) { g1_add(r, P, P); }
static void g2_mul_aff_dbl(g2a* r, const g2a* P) { g2_add(r, P, P); }
#define f_one_of fp_one_of
JAC_MUL(g1_mul, fp, g1a, fp_add, fp_sub, fp_mul, fp_sqr, fp_inv, fp_is_zero)
#undef f_one_of
#define f_one_of f2_one_of
JAC_MUL(g2_mul, fp2, g2a, f2_add, f2_sub, f2_mul, f2_sqr, f2_inv, f2_is_zero)
#undef f_one_of
static int g1_read(g1a* p, const uint8_t* b) {
  int z = 1; for (int i = 0; i < 2 * FPB; i++) z &= (b[i] == 0);
  p->inf = z; return fp_read(&p->x, b) & fp_read(&p->y, b + FPB);
}
static void g1_write(uint8_t* b, const g1a* p) { if (p->inf) { memset(b, 0, 2 * FPB); return; } fp_write(b, &p->x); fp_write(b + FPB, &p->y); }
static int g2_read(g2a* p, const uint8_t* b) {      /* x_im || x_re || y_im || y_re */
  int z = 1; for (int i = 0; i < 4 * FPB; i++) z &= (b[i] == 0);
  p->inf = z; return fp_read(&p->x.c1, b) & fp_read(&p->x.c0, b + FPB) & fp_read(&p->y.c1, b + 2 * FPB) & fp_read(&p->y.c0, b + 3 * FPB);
}
static void g2_write(uint8_t* b, const g2a* p) {
  if (p->inf) { memset(b, 0, 4 * FPB); return; }
  fp_write(b, &p->x.c1); fp_write(b + FPB, &p->x.c0); fp_write(b + 2 * FPB, &p->y.c1); fp_write(b + 3 * FPB, &p->y.c0);
}

/* ------------------------------------------------------------------ pairing (oracle/pyref/pairing.py) */
typedef struct { fp2 X, Y, Z; } g2p;
static void dbl_step(g2p* R, fp2 co[3]) {
  fp half; fp_set(&half, HALF);
  fp2 A, B, C, E, F, G, H, I, J, t, b2;
  f2_load(&b2, CB2);
  f2_mul(&A, &R->X, &R->Y); f2_muls(&A, &A, &half);
  f2_sqr(&B, &R->Y); f2_sqr(&C, &R->Z);
  f2_small(&t, &C, 3); f2_mul(&E, &b2, &t); f2_small(&F, &E, 3);
  f2_add(&G, &B, &F); f2_muls(&G, &G, &half);
  f2_add(&t, &R->Y, &R->Z); f2_sqr(&H, &t); f2_add(&t, &B, &C); f2_sub(&H, &H, &t);
  f2_sub(&I, &E, &B); f2_sqr(&J, &R->X);
  f2_sub(&t, &B, &F); f2_mul(&R->X, &A, &t);
  f2_sqr(&t, &E); f2_small(&t, &t, 3); f2_sqr(&R->Y, &G); f2_sub(&R->Y, &R->Y, &t);
  f2_mul(&R->Z, &B, &H);
  f2_neg(&co[0], &H); f2_small(&co[1], &J, 3); co[2] = I;
}
static void add_step(g2p* R, const fp2* xq, const fp2* yq, fp2 co[3]) {
  fp2 th, la, C, D, E, F, G, Hh, t, u, j;
  f2_mul(&t, yq, &R->Z); f2_sub(&th, &R->Y, &t);
  f2_mul(&t, xq, &R->Z); f2_sub(&la, &R->X, &t);
  f2_sqr(&C, &th); f2_sqr(&D, &la); f2_mul(&E, &la, &D); f2_mul(&F, &R->Z, &C); f2_mul(&G, &R->X, &D);
  f2_add(&Hh, &E, &F); f2_add(&t, &G, &G); f2_sub(&Hh, &Hh, &t);
  f2_mul(&t, &th, xq); f2_mul(&u, &la, yq); f2_sub(&j, &t, &u);
  f2_mul(&R->X, &la, &Hh);
  f2_sub(&t, &G, &Hh); f2_mul(&t, &th, &t); f2_mul(&u, &E, &R->Y); f2_sub(&R->Y, &t, &u);
  f2_mul(&R->Z, &R->Z, &E);
  co[0] = la; f2_neg(&co[1], &th); co[2] = j;
}
static void apply_line(fp12* f, const fp2 co[3], const g1a* P) {
  fp2 e[3];
#if CURVE_IS_BN
  static const int pos[3] = {0, 1, 3};
  f2_muls(&e[0], &co[0], &P->y); f2_muls(&e[1], &co[1], &P->x); e[2] = co[2];
#else
  static const int pos[3] = {0, 2, 3};
  e[0] = co[2]; f2_muls(&e[1], &co[1], &P->x); f2_muls(&e[2], &co[0], &P->y);
#endif
  f12_mul_sparse(f, e, pos);
}
static void miller(fp12* f, const g1a* P, const g2a* Q) {
  f12_one(f);
  if (P->inf || Q->inf) return;
  g2p R; R.X = Q->x; R.Y = Q->y; f2_one(&R.Z);
  fp2 nyq, co[3]; f2_neg(&nyq, &Q->y);
  for (int i = 1; i < LOOP_LEN; i++) {
    dbl_step(&R, co); f12_sqr(f, f); apply_line(f, co, P);
    if (LOOP_NAF[i]) { add_step(&R, &Q->x, LOOP_NAF[i] > 0 ? &Q->y : &nyq, co); apply_line(f, co, P); }
  }
#if CURVE_IS_BN
  fp2 g, x1, y1, x2, y2, t;
  f2_load(&g, GAMMA + (0 * 6 + 2) * 2 * NL); f2_conj(&t, &Q->x); f2_mul(&x1, &t, &g);
  f2_load(&g, GAMMA + (0 * 6 + 3) * 2 * NL); f2_conj(&t, &Q->y); f2_mul(&y1, &t, &g);
  f2_load(&g, GAMMA + (1 * 6 + 2) * 2 * NL); f2_mul(&x2, &Q->x, &g);
  f2_load(&g, GAMMA + (1 * 6 + 3) * 2 * NL); f2_mul(&y2, &Q->y, &g); f2_neg(&y2, &y2);
  add_step(&R, &x1, &y1, co); apply_line(f, co, P);
  add_step(&R, &x2, &y2, co); apply_line(f, co, P);
#else
  f12_conj(f, f);
#endif
}
static void final_exp(fp12* r, const fp12* in) {
  fp12 f, t, u;
  f12_conj(&t, in); f12_inv(&u, in); f12_mul(&f, &t, &u);
  f12_frob(&t, &f, 2); f12_mul(&f, &t, &f);
#if CURVE_IS_BN
  fp12 ft1, ft2, ft3, y0, y1, y2, y3, y4, y5, y6, t0, t1;
  f12_pow64_cyc(&ft1, &f, U_ABS); f12_pow64_cyc(&ft2, &ft1, U_ABS); f12_pow64_cyc(&ft3, &ft2, U_ABS);
  f12_frob(&y0, &f, 1); f12_frob(&t, &f, 2); f12_mul(&y0, &y0, &t); f12_frob(&t, &f, 3); f12_mul(&y0, &y0, &t);
  f12_conj(&y1, &f); f12_frob(&y2, &ft2, 2);
  f12_frob(&y3, &ft1, 1); f12_conj(&y3, &y3);
  f12_frob(&t, &ft2, 1); f12_mul(&y4, &ft1, &t); f12_conj(&y4, &y4);
  f12_conj(&y5, &ft2);
  f12_frob(&t, &ft3, 1); f12_mul(&y6, &ft3, &t); f12_conj(&y6, &y6);
  f12_cyclo_sqr(&t0, &y6); f12_mul(&t0, &t0, &y4); f12_mul(&t0, &t0, &y5);
  f12_mul(&t1, &y3, &y5); f12_mul(&t1, &t1, &t0);
  f12_mul(&t0, &t0, &y2);
  f12_cyclo_sqr(&t1, &t1); f12_mul(&t1, &t1, &t0); f12_cyclo_sqr(&t1, &t1);
  f12_mul(&t0, &t1, &y1); f12_mul(&t1, &t1, &y0);
  f12_cyclo_sqr(&t0, &t0); f12_mul(r, &t1, &t0);
#else
  /* (p^4-p^2+1)/r = c (x+p)(x^2+p^2-1) + 1, c = cofactor, x < 0 */
  fp12 a, b, d, ax, bx;
  f12_one(&a);
  { /* f^cofactor on the signed-digit (NAF) form of the exponent: in the cyclotomic subgroup the inverse is the conjugate */
    int8_t naf[132]; int nd = 0; unsigned __int128 k = ((unsigned __int128)COFACTOR[1] << 64) | COFACTOR[0];
    while (k) { int d = 0; if (k & 1) { d = 2 - (int)(k & 3); if (d < 0) k += 1; else k -= 1; } naf[nd++] = (int8_t)d; k >>= 1; }
    fp12 fi; f12_conj(&fi, &f);
    for (int i = nd - 1; i >= 0; i--) { f12_cyclo_sqr(&a, &a); if (naf[i] > 0) f12_mul(&a, &a, &f); else if (naf[i] < 0) f12_mul(&a, &a, &fi); }
  }
  f12_pow64_cyc(&ax, &a, U_ABS); f12_conj(&ax, &ax); f12_frob(&t, &a, 1); f12_mul(&b, &ax, &t);
  f12_pow64_cyc(&bx, &b, U_ABS); f12_conj(&bx, &bx); f12_pow64_cyc(&bx, &bx, U_ABS); f12_conj(&bx, &bx);
  f12_frob(&t, &b, 2); f12_mul(&d, &bx, &t); f12_conj(&t, &b); f12_mul(&d, &d, &t);
  f12_mul(r, &d, &f);
#endif
}

/* ------------------------------------------------------------------ hash to G1 */
#if CURVE_IS_BN
static int hash_to_g1(g1a* out, const uint8_t* msg, size_t len) {   /* tryAndIncrementEvm */
  uint8_t* buf = (uint8_t*)malloc(len + 1); uint8_t h[32];
  memcpy(buf + 1, msg, len);
  for (int c = 0; c < 256; c++) {
    buf[0] = (uint8_t)c; keccak256_legacy(buf, len + 1, h);
    fp x, y2, r, t; fp_from_be(&x, h);
    fp_to_mont(&x, &x);                               /* h mod q (Montgomery form) */
    fp_sqr(&y2, &x); fp_mul(&y2, &y2, &x); fp_set(&t, CB); fp_add(&y2, &y2, &t);
    fp_sqrt_cand(&r, &y2); fp_sqr(&t, &r);
    if (fp_eq(&t, &y2)) {
      buf[0] = 0xFF; keccak256_legacy(buf, len + 1, h);
      if (h[31] & 1) fp_neg(&r, &r);
      out->x = x; out->y = r; out->inf = 0; free(buf); return 1;
    }
  }
  free(buf); return 0;
}
#else
static int plain_parity(const fp* a_mont) {          /* parity(): a > q - a, curves/hash.go:169-172 */
  fp a, d; fp_from_mont(&a, a_mont); raw_sub(d.v, PMOD, a.v);
  return !raw_geq(d.v, a.v);
}
static void sw_encode(g1a* out, const fp* t) {        /* sw(), curves/hash.go:97-167, blind=false */
  fp one, b, w, x[3], g, y, tmp; fp_set(&one, ONE); fp_set(&b, CB);
  fp_sqr(&w, t); fp_add(&w, &w, &one); fp_add(&w, &w, &b); fp_inv(&w, &w); fp_mul(&w, &w, t);
  fp_set(&tmp, SQRT_M3); fp_mul(&w, &w, &tmp);
  fp_mul(&tmp, t, &w); fp_set(&x[0], Z_SW); fp_sub(&x[0], &x[0], &tmp);
  int i = 0;
  fp_sqr(&g, &x[0]); fp_mul(&g, &g, &x[0]); fp_add(&g, &g, &b);
  if (!fp_is_qr(&g)) {
    i = 1; fp_neg(&x[1], &x[0]); fp_sub(&x[1], &x[1], &one);
    fp_sqr(&g, &x[1]); fp_mul(&g, &g, &x[1]); fp_add(&g, &g, &b);
    if (!fp_is_qr(&g)) { i = 2; fp_sqr(&x[2], &w); fp_inv(&x[2], &x[2]); fp_add(&x[2], &x[2], &one);
      fp_sqr(&g, &x[2]); fp_mul(&g, &g, &x[2]); fp_add(&g, &g, &b); }
  }
  fp_sqrt_cand(&y, &g);
  if (plain_parity(&y) != plain_parity(t)) fp_neg(&y, &y);
  out->x = x[i]; out->y = y; out->inf = 0;
}
static void fouque_tibouchi(g1a* out, const uint8_t h[64]) {   /* bls12FouqueTibouchi, bls12_381.go:378-393 */
  /* t = int_be(h) mod q : (hi * 2^384 + lo) via Montgomery products */
  fp lo, hi, t, r2, r3, tp; memset(&hi, 0, sizeof hi);
  fp_from_be(&lo, h + 16);
  for (int i = 0; i < 2; i++) { uint64_t w = 0; for (int k = 0; k < 8; k++) w = (w << 8) | h[8 * (1 - i) + k]; hi.v[i] = w; }
  fp_set(&r2, R2); fp_set(&r3, R3); fp_mul(&lo, &lo, &r2); fp_mul(&hi, &hi, &r3); fp_add(&t, &lo, &hi);
  fp_from_mont(&tp, &t);
  fp root1, root2; fp_set(&root1, FT_ROOT1); fp_set(&root2, FT_ROOT2);
  if (fp_is_zero(&tp)) { memset(out, 0, sizeof *out); out->inf = 1; return; }
  g1a gen; fp_set(&gen.x, G1GEN); fp_set(&gen.y, G1GEN + NL); gen.inf = 0;
  if (fp_eq(&tp, &root1)) { *out = gen; return; }
  if (fp_eq(&tp, &root2)) { *out = gen; fp_neg(&out->y, &gen.y); return; }
  g1a p; sw_encode(&p, &t); g1_mul(out, &p, COFACTOR, 128);
}
static int hash_to_g1(g1a* out, const uint8_t* msg, size_t len) {
  uint8_t* buf = (uint8_t*)malloc(len + 4); uint8_t h[64]; g1a p1, p2;
  memcpy(buf, msg, len); memcpy(buf + len, "G1_0", 4);
  blake2b512(buf, len + 4, h); fouque_tibouchi(&p1, h);
  buf[len + 3] = '1'; blake2b512(buf, len + 4, h); fouque_tibouchi(&p2, h);
  g1_add(out, &p1, &p2); free(buf); return 1;
}
#endif

/* ------------------------------------------------------------------ threaded drivers */
typedef struct {
  const uint8_t *g1s, *g2s, *blob; const uint64_t* off; size_t n; int hash_first, faithful, tid, nthreads;
  fp12 acc; int bad;
} job_t;
static void* pair_worker(void* arg) {
  job_t* j = (job_t*)arg; f12_one(&j->acc); j->bad = 0;
  for (size_t i = j->tid; i < j->n; i += j->nthreads) {
    g1a P; g2a Q; fp12 f;
    if (j->hash_first) { if (!hash_to_g1(&P, j->blob + j->off[i], (size_t)(j->off[i + 1] - j->off[i]))) j->bad = 1; }
    else if (!g1_read(&P, j->g1s + i * 2 * FPB)) j->bad = 1;
    if (!g2_read(&Q, j->g2s + i * 4 * FPB)) j->bad = 1;
    miller(&f, &P, &Q);
    if (j->faithful) final_exp(&f, &f);            /* one full pairing per task, as curves/curve.go:132-134 */
    f12_mul(&j->acc, &j->acc, &f);
  }
  return NULL;
}
static int run_pairs(fp12* out, const uint8_t* g1s, const uint8_t* g2s, const uint8_t* blob, const uint64_t* off, size_t n,
                     int hash_first, int faithful, int threads) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  job_t* jobs = (job_t*)calloc((size_t)threads, sizeof(job_t)); pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; t++) {
    job_t j = {g1s, g2s, blob, off, n, hash_first, faithful, t, threads}; jobs[t] = j;
    if (threads == 1) pair_worker(&jobs[t]); else pthread_create(&th[t], NULL, pair_worker, &jobs[t]);
  }
  int bad = 0; f12_one(out);
  for (int t = 0; t < threads; t++) { if (threads > 1) pthread_join(th[t], NULL); bad |= jobs[t].bad; f12_mul(out, out, &jobs[t].acc); }
  free(jobs); free(th); return bad;
}

/* ------------------------------------------------------------------ exports */
int FN(hash_to_g1)(const uint8_t* msg, size_t len, uint8_t* out) { g1a p; if (!hash_to_g1(&p, msg, len)) return -3; g1_write(out, &p); return 0; }
int FN(miller)(const uint8_t* g1, const uint8_t* g2, uint8_t* out) {
  g1a P; g2a Q; fp12 f; if (!g1_read(&P, g1) || !g2_read(&Q, g2)) return -2; miller(&f, &P, &Q); gt_write(out, &f); return 0;
}
int FN(final_exp)(const uint8_t* in, uint8_t* out) { fp12 f; if (!gt_read(&f, in)) return -2; final_exp(&f, &f); gt_write(out, &f); return 0; }
int FN(pairing_product)(const uint8_t* g1s, const uint8_t* g2s, size_t n, uint8_t* out, int threads, int faithful) {
  fp12 f; if (run_pairs(&f, g1s, g2s, NULL, NULL, n, 0, faithful, threads)) return -2;
  if (!faithful) final_exp(&f, &f);
  gt_write(out, &f); return 0;
}
int FN(miller_product)(const uint8_t* g1s, const uint8_t* g2s, size_t n, uint8_t* out, int threads) {   /* no final exponentiation */
  fp12 f; if (run_pairs(&f, g1s, g2s, NULL, NULL, n, 0, 0, threads)) return -2; gt_write(out, &f); return 0;
}
int FN(gt_mul)(const uint8_t* a, const uint8_t* b, uint8_t* out) { fp12 x, y; if (!gt_read(&x, a) || !gt_read(&y, b)) return -2; f12_mul(&x, &x, &y); gt_write(out, &x); return 0; }
static int has_dup(const uint8_t* blob, const uint64_t* off, size_t n) {      /* containsDuplicateMessage */
  if (n < 2) return 0;
  size_t cap = 1; while (cap < 2 * n) cap <<= 1;
  uint32_t* tab = (uint32_t*)calloc(cap, 4); int dup = 0;
  for (size_t i = 0; i < n && !dup; i++) {
    const uint8_t* m = blob + off[i]; size_t len = (size_t)(off[i + 1] - off[i]);
    uint64_t h = 1469598103934665603ull; for (size_t k = 0; k < len; k++) { h ^= m[k]; h *= 1099511628211ull; }
    size_t s = (size_t)(h ^ (h >> 31)) & (cap - 1);
    while (tab[s]) { size_t j = tab[s] - 1; size_t l2 = (size_t)(off[j + 1] - off[j]);
      if (l2 == len && memcmp(blob + off[j], m, len) == 0) { dup = 1; break; } s = (s + 1) & (cap - 1); }
    tab[s] = (uint32_t)i + 1;
  }
  free(tab); return dup;
}
int FN(verify_aggregate)(const uint8_t* sig, const uint8_t* keys, const uint8_t* blob, const uint64_t* off, size_t n,
                         int allow_dups, int threads, int faithful) {
  if (!allow_dups && has_dup(blob, off, n)) return 0;
  fp12 f, fs; g1a S; g2a G2;
  if (run_pairs(&f, NULL, keys, blob, off, n, 1, faithful, threads)) return -2;
  if (!g1_read(&S, sig)) return -2;
  if (!S.inf) fp_neg(&S.y, &S.y);
  f2_load(&G2.x, G2GEN); f2_load(&G2.y, G2GEN + 2 * NL); G2.inf = 0;
  miller(&fs, &S, &G2); if (faithful) final_exp(&fs, &fs);
  f12_mul(&f, &f, &fs); if (!faithful) final_exp(&f, &f);
  return f12_is_one(&f);
}
int FN(aggregate_points)(int group, const uint8_t* pts, size_t n, uint8_t* out) {
  if (group == 1) { g1a acc, p; memset(&acc, 0, sizeof acc); acc.inf = 1;
    for (size_t i = 0; i < n; i++) { if (!g1_read(&p, pts + i * 2 * FPB)) return -2; g1_add(&acc, &acc, &p); } g1_write(out, &acc); }
  else { g2a acc, p; memset(&acc, 0, sizeof acc); acc.inf = 1;
    for (size_t i = 0; i < n; i++) { if (!g2_read(&p, pts + i * 4 * FPB)) return -2; g2_add(&acc, &acc, &p); } g2_write(out, &acc); }
  return 0;
}
int FN(verify_multi)(const uint8_t* sig, const uint8_t* keys, size_t n, const uint8_t* msg, size_t len, int faithful) {
  g2a apk, p, G2; memset(&apk, 0, sizeof apk); apk.inf = 1;
  for (size_t i = 0; i < n; i++) { if (!g2_read(&p, keys + i * 4 * FPB)) return -2; g2_add(&apk, &apk, &p); }
  g1a H, S; if (!hash_to_g1(&H, msg, len)) return -3; if (!H.inf) fp_neg(&H.y, &H.y);
  if (!g1_read(&S, sig)) return -2;
  f2_load(&G2.x, G2GEN); f2_load(&G2.y, G2GEN + 2 * NL); G2.inf = 0;
  fp12 f1, f2; miller(&f1, &H, &apk); miller(&f2, &S, &G2);
  if (faithful) { final_exp(&f1, &f1); final_exp(&f2, &f2); }
  f12_mul(&f1, &f1, &f2); if (!faithful) final_exp(&f1, &f1);
  return f12_is_one(&f1);
}
int FN(scale_point)(int group, const uint8_t* pt, const uint8_t* k_be32, int negative, uint8_t* out) {
  uint64_t k[4]; for (int i = 0; i < 4; i++) { uint64_t w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | k_be32[8 * (3 - i) + j]; k[i] = w; }
  if (group == 1) { g1a p, r; if (!g1_read(&p, pt)) return -2; if (negative && !p.inf) fp_neg(&p.y, &p.y); g1_mul(&r, &p, k, 256); g1_write(out, &r); }
  else { g2a p, r; if (!g2_read(&p, pt)) return -2; if (negative && !p.inf) f2_neg(&p.y, &p.y); g2_mul(&r, &p, k, 256); g2_write(out, &r); }
  return 0;
}
/* G2 membership by the DEFINITION: canonical coordinates, on the twist y^2 = x^3 + b', and [r]Q = infinity.  What the
 * reference gets from upstream when a G2 Point is constructed (curves/altbn128.go:157-179,329-376 -> bn256 G2.Unmarshal;
 * curves/bls12_381.go:242-264 Check()).  1 = member, 0 = not, -2 = non-canonical encoding. */
int FN(g2_in_subgroup)(const uint8_t* pt) {
  g2a p, r; if (!g2_read(&p, pt)) return -2;
  if (p.inf) return 1;
  fp2 l, rr, b2; f2_load(&b2, CB2);
  f2_sqr(&l, &p.y); f2_sqr(&rr, &p.x); f2_mul(&rr, &rr, &p.x); f2_add(&rr, &rr, &b2);
  if (!f2_eq(&l, &rr)) return 0;
  g2_mul(&r, &p, ORDER, 256);
  return r.inf ? 1 : 0;
}
/* G1 membership by the DEFINITION: canonical coordinates, on the curve y^2 = x^3 + b, and [r]P = infinity (what the reference
 * gets when a G1 Point is constructed with a check: curves/bls12_381.go:196-264 pt.Check(); alt-bn128's G1 has cofactor 1, so
 * every curve point is a member).  1 = member, 0 = not, -2 = non-canonical encoding. */
int FN(g1_in_subgroup)(const uint8_t* pt) {
  g1a p, r; if (!g1_read(&p, pt)) return -2;
  if (p.inf) return 1;
  fp l, rr, b; fp_set(&b, CB);
  fp_sqr(&l, &p.y); fp_sqr(&rr, &p.x); fp_mul(&rr, &rr, &p.x); fp_add(&rr, &rr, &b);
  if (!fp_eq(&l, &rr)) return 0;
  g1_mul(&r, &p, ORDER, 256);
  return r.inf ? 1 : 0;
}
