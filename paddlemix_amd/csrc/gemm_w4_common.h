// Helpers shared by the four-wave tiles (gemm_w4.hip: 16-bit operands; gemm_w4f8.hip: e4m3 operands)
#pragma once
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

namespace sd {
namespace w4 {
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
template <int V>
using ic_t = std::integral_constant<int, V>;

__device__ __forceinline__ void dma(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)lds, 16, voff, soff, 0, 0);
}
// lane index from the execution mask (v_mbcnt), through an opaque asm: recomputed wherever it is needed instead of keeping the
// thread id -- or anything derived from it -- alive across the K loop (kept alive it was spilled, and a scratch reload next to the
// epilogue's stores or the carried prologue is a vmcnt(0))
__device__ __forceinline__ int lane_id() {
  int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(l));
  return l;
}
}  // namespace w4
}  // namespace sd
