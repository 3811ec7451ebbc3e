/* CPU oracle entry points (curve id -> implementation) -- TEST INFRASTRUCTURE ONLY. */
#include <stdint.h>
#include <stddef.h>
#define DECL(p) \
  int p##_hash_to_g1(const uint8_t*, size_t, uint8_t*); int p##_miller(const uint8_t*, const uint8_t*, uint8_t*); \
  int p##_final_exp(const uint8_t*, uint8_t*); int p##_pairing_product(const uint8_t*, const uint8_t*, size_t, uint8_t*, int, int); \
  int p##_verify_aggregate(const uint8_t*, const uint8_t*, const uint8_t*, const uint64_t*, size_t, int, int, int); \
  int p##_aggregate_points(int, const uint8_t*, size_t, uint8_t*); int p##_verify_multi(const uint8_t*, const uint8_t*, size_t, const uint8_t*, size_t, int); \
  int p##_scale_point(int, const uint8_t*, const uint8_t*, int, uint8_t*); \
  int p##_miller_product(const uint8_t*, const uint8_t*, size_t, uint8_t*, int); int p##_gt_mul(const uint8_t*, const uint8_t*, uint8_t*); \
  int p##_g2_in_subgroup(const uint8_t*); int p##_g1_in_subgroup(const uint8_t*);
DECL(bn) DECL(bls)
#define D(name, ...) (curve == 0 ? bn_##name(__VA_ARGS__) : curve == 1 ? bls_##name(__VA_ARGS__) : -1)
int oracle_hash_to_g1(int curve, const uint8_t* m, size_t l, uint8_t* o) { return D(hash_to_g1, m, l, o); }
int oracle_miller(int curve, const uint8_t* a, const uint8_t* b, uint8_t* o) { return D(miller, a, b, o); }
int oracle_final_exp(int curve, const uint8_t* a, uint8_t* o) { return D(final_exp, a, o); }
int oracle_pairing_product(int curve, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* o, int threads, int faithful) { return D(pairing_product, a, b, n, o, threads, faithful); }
int oracle_verify_aggregate(int curve, const uint8_t* s, const uint8_t* k, const uint8_t* b, const uint64_t* off, size_t n, int dups, int threads, int faithful) { return D(verify_aggregate, s, k, b, off, n, dups, threads, faithful); }
int oracle_aggregate_points(int curve, int g, const uint8_t* p, size_t n, uint8_t* o) { return D(aggregate_points, g, p, n, o); }
int oracle_verify_multi(int curve, const uint8_t* s, const uint8_t* k, size_t n, const uint8_t* m, size_t l, int faithful) { return D(verify_multi, s, k, n, m, l, faithful); }
int oracle_scale_point(int curve, int g, const uint8_t* p, const uint8_t* k, int neg, uint8_t* o) { return D(scale_point, g, p, k, neg, o); }
int oracle_miller_product(int curve, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* o, int threads) { return D(miller_product, a, b, n, o, threads); }
int oracle_gt_mul(int curve, const uint8_t* a, const uint8_t* b, uint8_t* o) { return D(gt_mul, a, b, o); }
int oracle_g2_in_subgroup(int curve, const uint8_t* p) { return D(g2_in_subgroup, p); }
int oracle_g1_in_subgroup(int curve, const uint8_t* p) { return D(g1_in_subgroup, p); }
