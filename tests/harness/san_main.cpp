// Sanitizer run of the device arithmetic headers compiled for the host (CPU tier; SURVEY section 5: "host tests under ASan/UBSan").
// Built by tests/test_host_arith.py with -fsanitize=address,undefined -fno-sanitize-recover=all: any out-of-bounds access of a
// limb array, signed overflow, over-wide shift or misaligned access in fp.hpp / tower.hpp / pairing.hpp / h2c.hpp / rx.hpp /
// rx_pair.hpp / rx_pow.hpp / rx_jac*.hpp aborts the process.  It walks what the Miller kernel k_miller_x60 executes -- the
// lane-pair point steps of a whole Miller loop on both curves AND on alt-bn128's nine-limb 29-bit form (harness curve id 2), the
// consumer's folds / squarings / xi multiples on worst-case limbs -- plus the hash maps, a key sum and the square-root powers.
#include <stdio.h>
#include <vector>
#include "host_harness.cpp"

template <class C>
static void gens(uint8_t* g1, uint8_t* g2) {
  Aff<F1<C>> a = {fp_load<C>(C::G1X), fp_load<C>(C::G1Y), false};
  g1_to_bytes<C>(g1, a);
  Aff<F2<C>> b = {f2_load<C>(C::G2), f2_load<C>(C::G2 + 2 * C::L), false};
  g2_to_bytes<C>(g2, b);
}

#define CHECK(expr)                                        \
  do {                                                     \
    if (!(expr)) {                                         \
      fprintf(stderr, "san_main: failed: %s\n", #expr);    \
      return 1;                                            \
    }                                                      \
  } while (0)

int main() {
  uint8_t g1[2][96], g2[2][192], out[768];
  gens<BN254>(g1[0], g2[0]);
  gens<BLS381>(g1[1], g2[1]);
  // lane-pair point steps over a whole Miller loop against pairing.hpp, every column accumulation checked: ids 0, 1 and 2 (BN254W)
  CHECK(ht_rx_miller(0, g1[0], g2[0]) == 0);
  CHECK(ht_rx_miller(1, g1[1], g2[1]) == 0);
  CHECK(ht_rx_miller(2, g1[0], g2[0]) == 0);
  // the 32-bit Miller loop and the hash maps
  CHECK(ht_miller(0, g1[0], g2[0], out) == 0);
  CHECK(ht_miller(1, g1[1], g2[1], out) == 0);
  const uint8_t msg[5] = {'b', 'g', 'l', 's', 0};
  CHECK(ht_hash_to_g1(0, msg, 5, out) == 0);
  CHECK(ht_hash_to_g1(1, msg, 5, out) == 0);
  // k_bls_sw_jacobi's work item on the carry-free limbs (h2c_x.hpp), both tags, and an all-ones digest
  CHECK(ht_bls_sw_x(msg, 5, 0, out) == 3);
  CHECK(ht_bls_sw_x(msg, 5, 1, out) == 3);
  {
    uint8_t dg[64];
    memset(dg, 0xFF, sizeof(dg));
    CHECK(ht_bls_sw_x_digest(dg, out) == 3);
  }
  // consumer arithmetic on worst-case limbs (every limb at its bound)
  for (int cid = 0; cid < 3; ++cid) {
    const int N = cid == 0 ? 10 : (cid == 1 ? 14 : 9), W = cid == 2 ? 29 : 28;
    std::vector<u32> A(5 * 2 * N), B(5 * 2 * N), o(2 * N);
    for (size_t i = 0; i < A.size(); ++i) {
      const bool top = (int)(i % N) == N - 1;
      A[i] = top ? 1000u : (1u << W) - 1;
      B[i] = top ? 1000u : (1u << W) - 1;
    }
    CHECK(ht_rx_raw(cid, 0, 0, A.data(), B.data(), o.data()) == 0);          // three-term fold
    CHECK(ht_rx_raw(cid, 2, 0, A.data(), B.data(), o.data()) == 0);          // xi multiple
    if (cid == 2) { CHECK(ht_rx_raw(cid, 3, 0, A.data(), B.data(), o.data()) == 0); CHECK(ht_rx_raw(cid, 3, 1, A.data(), B.data(), o.data()) == 0); }      // two-pile squaring, even / odd row
    else CHECK(ht_rx_raw(cid, 1, 2 | (2 << 2) | (2 << 4), A.data(), B.data(), o.data()) == 0);
    uint8_t be[48] = {0};
    be[cid == 1 ? 47 : 31] = 7;
    std::vector<u32> lim(N);
    CHECK(ht_rx_conv(cid, 0, be, lim.data()) == 0);
    CHECK(ht_rx_conv(cid, 1, be, lim.data()) == 0);
    CHECK(be[cid == 1 ? 47 : 31] == 7);
  }
  // key sums on the carry-free limbs (one lane, lane pair) and the square-root powers
  for (int cid = 0; cid < 2; ++cid) {
    const int pb = cid == 0 ? 128 : 192;
    std::vector<uint8_t> pts(3 * pb);
    for (int k = 0; k < 3; ++k) memcpy(pts.data() + k * pb, g2[cid], pb);      // P + P + P: the doubling branch, then an addition
    uint8_t s0[192], s1[192], s2[192];
    CHECK(ht_rx_sum(cid, pts.data(), 3, s0, s1) == 0);
    CHECK(ht_rx_sumpair(cid, pts.data(), 3, s2) == 0);
    CHECK(memcmp(s0, s1, pb) == 0 && memcmp(s0, s2, pb) == 0);
    uint8_t x[48] = {0};
    x[cid == 0 ? 31 : 47] = 4;
    CHECK(ht_rx_pow(cid, 1, x, nullptr) == 0);
  }
  printf("san_main ok\n");
  return 0;
}
