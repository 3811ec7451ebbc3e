// Final exponentiation of a verification on the carry-free limbs (finalx.hpp): product of `count` serialised partials, the
// exponent (p^12 - 1) / r, comparison with one.  One block of FOUR waves: a product occupies two of them, and the powers by the
// curve parameter and the hard part's chain keep two independent products in flight (fx_mul_pair).  Same interface and results as k_final36 (k_tail.inc), which it replaces in
// Engine::finish; own translation unit (both curves).
#include "dev_common.hpp"
#include "finalx.hpp"
#include "launch.hpp"
#include "launch_tail.hpp"

namespace bgls {

template <class C>
__global__ void __launch_bounds__(256) k_finalx(const uint8_t* partials, size_t count, int do_final_exp, uint8_t* gt_out, uint32_t* verdict,
                                               uint32_t* flags, const uint32_t* flags_in, int packed) {
  typedef FX<C> E;
  const int lane = threadIdx.x;
  const int order_pos[6] = {5, 2, 4, 1, 3, 0};
  FX_T(0);
  // packed: verdict[0..2] = {verdict, this stage's flags, the caller's flag word} -- ONE copy back to the host instead of three and
  // no flag word to clear beforehand; else the verdict alone and the flags OR-ed into *flags
  bool bad = false;                                         // (no static LDS here: it would shift the 16-byte alignment of the slots)
  for (size_t k = 0; k < count; ++k) {
    if (lane < 6) {
      const uint8_t* b = partials + k * 12 * C::FP_BYTES + (size_t)(2 * order_pos[lane]) * C::FP_BYTES;
      const Fp<C> im = fp_from_be<C>(b), re = fp_from_be<C>(b + C::FP_BYTES);
      if (fp_geq_p<C>(im) || fp_geq_p<C>(re)) bad = true;
      fx_put<C>(k == 0 ? FE_F : FE_X, lane, X2<C, SX_T>{sx_from_plain<C>(re), sx_from_plain<C>(im)});
    }
    __syncthreads();
    if (k > 0) fx_mul<C>(FE_F, FE_F, FE_X);
  }
  if (do_final_exp) fx_final_exp<C>();
  FX_T(8);
  bool is_one = true;
  if (lane < 6) {
    const X2<C, SX_T> x = fx_ld2<C>(E::coef(FE_F, lane, 0));
    const Fp2<C> v = {sx_to_mont<C>(x.c0), sx_to_mont<C>(x.c1)};
    is_one = lane == 0 ? f2_eq<C>(v, f2_one<C>()) : f2_is_zero<C>(v);
    if (gt_out) {
      uint8_t* o = gt_out + (size_t)(2 * order_pos[lane]) * C::FP_BYTES;
      fp_to_be<C>(o, fp_from_mont<C>(v.c1));
      fp_to_be<C>(o + C::FP_BYTES, fp_from_mont<C>(v.c0));
    }
  }
  const unsigned long long ball = __ballot(is_one);       // wave 0 holds the six coefficients (the other lanes vote "one")
  FX_T(11);
  const unsigned long long bball = __ballot(bad);           // lanes 0..5 of wave 0 parsed the partials
  if (lane == 0) {
    verdict[0] = (ball == ~0ull) ? 1u : 0u;
    const uint32_t fl = bball ? FLAG_ENC : 0u;
    if (packed) {
      verdict[1] = fl;
      verdict[2] = flags_in ? flags_in[0] : 0u;
    } else if (fl) {
      atomicOr(flags, fl);
    }
  }
}

namespace kl {
template <class C>
void finalx(hipStream_t st, const uint8_t* partials, size_t count, int do_final_exp, uint8_t* gt_out, uint32_t* verdict, uint32_t* flags) {
  k_finalx<C><<<1, 256, FX<C>::LDS_BYTES_PAIR, st>>>(partials, count, do_final_exp, gt_out, verdict, flags, nullptr, 0);
}
template <class C>
void finalx_res(hipStream_t st, const uint8_t* partials, size_t count, int do_final_exp, uint8_t* gt_out, uint32_t* res3, const uint32_t* flags_in) {
  k_finalx<C><<<1, 256, FX<C>::LDS_BYTES_PAIR, st>>>(partials, count, do_final_exp, gt_out, res3, nullptr, flags_in, 1);
}
template void finalx_res<BN254>(hipStream_t, const uint8_t*, size_t, int, uint8_t*, uint32_t*, const uint32_t*);
template void finalx_res<BLS381>(hipStream_t, const uint8_t*, size_t, int, uint8_t*, uint32_t*, const uint32_t*);
template void finalx<BN254>(hipStream_t, const uint8_t*, size_t, int, uint8_t*, uint32_t*, uint32_t*);
template void finalx<BLS381>(hipStream_t, const uint8_t*, size_t, int, uint8_t*, uint32_t*, uint32_t*);
}  // namespace kl
}  // namespace bgls
#ifdef FX_DBG
extern "C" int bgls_dbg_fx_dump(unsigned long long* o) { return (int)hipMemcpyFromSymbol(o, HIP_SYMBOL(bgls::g_fx_t), sizeof(bgls::g_fx_t)); }
#endif
