// Types shared by the kernel translation units (k_*.hip) and the host engine (engine.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "pairing.hpp"

namespace bgls {

// n messages: blob + n+1 offsets (host API, bgls.go's [][]byte) or fixed stride (device API)
struct MsgView {
  const uint8_t* base;
  const uint64_t* off;  // n+1 offsets, or nullptr for fixed stride
  size_t len, stride;
  __device__ __forceinline__ const uint8_t* ptr(size_t i) const { return off ? base + off[i] : base + i * stride; }
  __device__ __forceinline__ size_t size(size_t i) const { return off ? (size_t)(off[i + 1] - off[i]) : len; }
};

// status word of a verification (device u32, OR-ed by the kernels; mapped to BGLS_ERR_* / verdict 0 by the engine)
enum : uint32_t { FLAG_DUP = 1u, FLAG_ENC = 2u, FLAG_HASH = 4u, FLAG_SUBGROUP = 8u, FLAG_DEGENERATE = 16u };

inline unsigned nblk(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace bgls
