// Shared GEMM epilogue (fp8 weight scale / bias / time-embedding row bias / residual / scale / SiLU / GEGLU, bf16 or fp32 stores).
// Accumulator layout: acc[tn][tm] is the "swapped" 16x16 MFMA tile whose lane holds 4 consecutive channels of output row
// m = m_wave + tm*16 + (lane&15). WHICH channels is a free choice -- MFMA row i of n-sub-tile tn is whatever weight row
// the LDS-DMA put into LDS row tn*16+i -- and it is chosen for the stores: sub-tiles are paired so that a lane owns 8
// CONSECUTIVE channels (acc[2h] | acc[2h+1]) = one 16-byte bf16 store, and the four lane groups of a row cover 64
// contiguous bytes per store instruction (scripts/probes/store_probe.hip: 5.8 vs 3.75 TB/s for the 8-byte / 32-byte-
// per-row pattern of the natural mapping). acc_col() is the single definition of the mapping; the DMA source rows
// (w_row_of_lds_row), the split-K partial sums and the epilogues all derive from it.
// Loads (round 3, scripts/gemm_timeline.py): with one dependent load -> wait -> use round per 4-channel group an epilogue costs
// TM x TN memory round trips per wave with ~4 KB in flight per CU -- 12.6 us of the 39.6 us of an 8192 x 1280 x 1280 + R launch,
// 35 us of 65 us at 32768 x 640 x 640 + R (TM x TN = 40), 8.4 us per FF1 tile for nothing but its bias (a load the compiler cannot
// hoist over the stores). Now: (a) the bias is the INITIAL VALUE of the accumulators (GemmArgs::bias_acc, gemm_pipe.hip) -- no
// epilogue load at all; (b) the residual of the register-pipelined tiles is fetched during the last K iterations (EpiPre below);
// (c) everywhere else the operands of a whole row-tile (row bias, gate, residual: up to 3 x TN loads) are issued back to back in
// front of its math (GemmArgs::epi_batch), TM round trips instead of TM x TN.
// Round 1 had tried branch-free batching through zero-sized descriptors (source in the history of the repository, profiles/experiments/ before round 6: r01_gemm_epilogue_batched_loads.h.txt)
// and seen no gain in the step; two lessons kept from it: (1) never give the accumulators two alternative consumer loops (a fast
// path next to a general one, or an e4m3 and a bf16 store loop) -- the register allocator then splits their live ranges and spills
// them INSIDE the K loop; (2) per-channel operands folded into all accumulators up front pin the whole accumulator set in VGPRs.
#pragma once
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace sd {

static __device__ __attribute__((aligned(16))) const unsigned g_zero16[4] = {0u, 0u, 0u, 0u};
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// first channel (relative to the wave's n-range) of the 4 values lane-group lq holds in acc[tn]
template <int TN>
__device__ __forceinline__ int acc_col(int tn, int lq, bool geglu) {
  if (!geglu) return tn < (TN & ~1) ? (tn >> 1) * 32 + lq * 8 + (tn & 1) * 4 : tn * 16 + lq * 4;
  // GEGLU weights are stored in interleaved groups of 32 rows (16 value | 16 gate): acc[2tp] = value, acc[2tp+1] =
  // gate of the same 4 OUTPUT channels ol..ol+3 (output pairs paired again for 16-byte stores)
  const int tp = tn >> 1;
  const int ol = tp < ((TN / 2) & ~1) ? (tp >> 1) * 32 + lq * 8 + (tp & 1) * 4 : tp * 16 + lq * 4;
  return (ol >> 4) * 32 + (ol & 15) + (tn & 1) * 16;
}
// weight row (relative to the block tile) that belongs into LDS row j of the W tile
template <int TN>
__device__ __forceinline__ int w_row_of_lds_row(int j, bool geglu) {
  const int wn = j / (TN * 16), jl = j - wn * (TN * 16);
  const int i = jl & 15;
  return wn * (TN * 16) + acc_col<TN>(jl >> 4, i >> 2, geglu) + (i & 3);
}

// bias / row bias / gate / residual / scale / activation on 4 channels n..n+3 of row m
__device__ __forceinline__ f32x4 act4(const GemmArgs& p, f32x4 v) {
  if (p.silu) {
    v[0] = silu_f(v[0]);
    v[1] = silu_f(v[1]);
    v[2] = silu_f(v[2]);
    v[3] = silu_f(v[3]);
  }
  if (p.gelu_tanh) {
    v[0] = gelu_tanh_f(v[0]);
    v[1] = gelu_tanh_f(v[1]);
    v[2] = gelu_tanh_f(v[2]);
    v[3] = gelu_tanh_f(v[3]);
  }
  return v;
}
__device__ __forceinline__ f32x4 add_r16(f32x4 v, u32x2 raw) {
  const bf16x4 r4 = __builtin_bit_cast(bf16x4, raw);
  v[0] += (float)r4[0];
  v[1] += (float)r4[1];
  v[2] += (float)r4[2];
  v[3] += (float)r4[3];
  return v;
}
// Operands of one 4-channel group fetched ahead of the row-tile's math (GemmArgs::epi_batch). To keep the register cost at 6 per
// group the fp32 operands share one slot: the gate when there is one, else the sum of row bias and fp32 residual; the two
// combinations that would need a second fp32 slot (gate next to a row bias or an fp32 residual -- no program of this library
// emits them) take the group-at-a-time path. 16-bit residuals are kept packed.
struct Epi4Ops {
  f32x4 x;     // gate, or row bias (+ fp32 residual)
  u32x2 r16;   // 16-bit residual
};
__device__ __forceinline__ bool epi_batched(const GemmArgs& p) {
  return p.epi_batch && !p.geglu && !(p.gate && (p.rowbias || (p.R && p.r_f32)));
}
template <bool WS = false, bool BR = false>
__device__ __forceinline__ f32x4 epi4(const GemmArgs& p, f32x4 v, int m, int n, const float* rb, const float* gt, bool batched,
                                      const Epi4Ops& o, const f32x4& wsc, const f32x4& breg = f32x4{0.f, 0.f, 0.f, 0.f}) {
  if constexpr (WS) v *= wsc;   // (the weight-scale instantiations: the scale is in registers, gemm_epilogue below)
  else if (p.wscale) v *= *reinterpret_cast<const f32x4*>(p.wscale + n);
  if constexpr (BR) v += breg;  // (the bias-in-registers instantiation: zero where there is no bias)
  else if (p.bias && !p.bias_acc) v += *reinterpret_cast<const f32x4*>(p.bias + n);
  if (batched) {
    if (gt) v *= o.x;
    else if (rb || (p.R && p.r_f32)) v += o.x;
    if (p.R && !p.r_f32) v = add_r16(v, o.r16);
    return act4(p, v * p.out_scale);
  }
  if (rb) v += *reinterpret_cast<const f32x4*>(rb + n);
  if (gt) v *= *reinterpret_cast<const f32x4*>(gt + n);
  if (p.R) {
    if (p.r_f32) v += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + (size_t)m * p.ldr + n);
    else v = add_r16(v, *reinterpret_cast<const u32x2*>(p.R + (size_t)m * p.ldr + n));
  }
  return act4(p, v * p.out_scale);
}
__device__ __forceinline__ void store4(const GemmArgs& p, size_t off, f32x4 v) {
  if (p.out_f32) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + off) = v;
  } else {
    u32x2 pk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + off) = pk;
  }
}
// 8 consecutive channels; one 16-byte store when the destination allows it (p.c_wide: C, ldc, c_bstride 16-byte aligned)
__device__ __forceinline__ void store8(const GemmArgs& p, size_t off, f32x4 lo, f32x4 hi) {
  if (p.out_f32) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + off) = lo;
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + off + 4) = hi;
  } else if (p.c_wide) {
    u32x4 pk = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16*>(p.C) + off) = pk;
  } else {
    u32x2 p0 = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3])}, p1 = {pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + off) = p0;
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16*>(p.C) + off + 4) = p1;
  }
}
__device__ __forceinline__ f32x4 geglu4(f32x4 h, f32x4 g) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = h[r] * gelu_erf_f(g[r]);
  return o;
}
// stores of one row-tile: GEGLU pairs or plain channels, 8 consecutive channels per lane where sub-tiles pair up.
// fin(tn, n) returns the finished 4 values of acc[tn] (first channel n; the physical weight column for GEGLU).
template <int TN, class F>
__device__ __forceinline__ void store_row(const GemmArgs& p, size_t crow, int n_wave, int lq, F&& fin) {
  if (p.geglu) {
#pragma unroll
    for (int hp = 0; hp < TN / 4; ++hp) {   // two output pairs = 8 consecutive output channels
      const int n_phys = n_wave + acc_col<TN>(4 * hp, lq, true);
      if (n_phys >= p.N) continue;          // N % 32 == 0: the whole interleaved group is inside
      store8(p, crow + (n_wave >> 1) + hp * 32 + lq * 8, geglu4(fin(4 * hp, n_phys), fin(4 * hp + 1, n_phys + 16)),
             geglu4(fin((4 * hp + 2) % TN, n_phys + 4), fin((4 * hp + 3) % TN, n_phys + 20)));
    }
    if constexpr ((TN / 2) & 1) {
      constexpr int tp = TN / 2 - 1;
      const int n_phys = n_wave + tp * 32 + lq * 4;
      if (n_phys < p.N)
        store4(p, crow + (n_wave >> 1) + tp * 16 + lq * 4, geglu4(fin(2 * tp, n_phys), fin((2 * tp + 1) % TN, n_phys + 16)));
    }
  } else {
#pragma unroll
    for (int h = 0; h < TN / 2; ++h) {
      const int n = n_wave + h * 32 + lq * 8;
      if (n >= p.N) continue;
      const f32x4 lo = fin(2 * h, n);
      if (n + 4 < p.N) store8(p, crow + n, lo, fin(2 * h + 1, n + 4));
      else store4(p, crow + n, lo);
    }
    if constexpr (TN & 1) {
      const int n = n_wave + (TN - 1) * 16 + lq * 4;
      if (n < p.N) store4(p, crow + n, fin(TN - 1, n));
    }
  }
}

// WS (round 5): instantiations of their own for launches with a per-channel weight scale (weight-only fp8: the generic e4m3 loop
// and the 16-bit loops on a just-in-time widened matrix). Read where it is used -- the WS = false form, which those launches ran
// on before -- the scale is TM x TN dependent load -> wait -> multiply round trips per wave, loads the compiler cannot hoist over
// the stores: 3-4 us per tile, 59 us of a 32768 x 6144 x 1536 launch (profiles/r05_s17_per_shape_sd3-1024-bs8-fp8w.txt). Here the
// lane's TN scale vectors are fetched once in front of the row-tile loop -- for tiles of up to 128 accumulator registers; the
// 160-register tiles (256 x 320) spill 25-340 registers with 4 TN more live ones, fetched up front or per row-tile, and keep the
// old form (launch_gemm's picker keeps widened matrices off them). Both forms were tried on ALL kernels first: the 256 x 320 kernels
// spilled, and scaling the accumulators up front made nearly every kernel spill (header, lesson 2) -- hence separate
// instantiations; the WS = false code is what it was.
// BR (round 6, the four-wave 256 x 256 tile of gemm_w4.hip): the lane's TN bias vectors are fetched once in front of the row-tile loop
// (that kernel's accumulators start at zero -- its first MFMAs take the constant 0 as C operand, no initialisation pass -- so the bias
// is added here, from registers, instead of one dependent load per 4-channel group).
template <int TM, int TN, bool WS = false, bool BR = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[TN][TM], int m_wave, int n_wave,
                                              int lane) {
  const int lq = lane >> 4;
  constexpr bool WSH = WS && TM * TN <= 32;   // (the 160-register tiles keep the read-where-used form: they spill otherwise)
  f32x4 bsr[BR ? TN : 1];
  if constexpr (BR) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
      bsr[tn] = (p.bias && !p.bias_acc) ? *reinterpret_cast<const f32x4*>(p.bias + min(n_wave + acc_col<TN>(tn, lq, p.geglu), p.N - 4))
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 wsc[WSH ? TN : 1];
  if constexpr (WSH) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
      wsc[tn] = *reinterpret_cast<const f32x4*>(p.wscale + min(n_wave + acc_col<TN>(tn, lq, p.geglu), p.N - 4));
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    const float* rb = p.rowbias ? p.rowbias + (size_t)(m / p.rows_per_batch) * p.ld_rowbias : nullptr;
    const float* gt = p.gate ? p.gate + (size_t)(m / p.rows_per_batch) * p.ld_gate : nullptr;
    const size_t crow = p.c_rpb ? (size_t)(m / p.c_rpb) * p.c_bstride + (size_t)(m % p.c_rpb) * p.ldc : (size_t)m * p.ldc;
    // this row-tile's operands, all loads in flight before the first is used (channels past N: the group's address is clamped
    // into the row, its stores are skipped). Tiles of more than 6 sub-tiles per wave are GEGLU-only (no such operands).
    constexpr int TB = TN <= 6 ? TN : 1;
    Epi4Ops ops[TB];
    const bool batched = TN <= 6 && epi_batched(p);
    if (TN <= 6 && batched) {
      int nc[TB];
#pragma unroll
      for (int tn = 0; tn < TB; ++tn) nc[tn] = min(n_wave + acc_col<TN>(tn, lq, false), p.N - 4);
      const bool r32 = p.R && p.r_f32;
      if (gt) {
#pragma unroll
        for (int tn = 0; tn < TB; ++tn) ops[tn].x = *reinterpret_cast<const f32x4*>(gt + nc[tn]);
      } else if (rb && r32) {
        f32x4 t[TB];
#pragma unroll
        for (int tn = 0; tn < TB; ++tn) {
          ops[tn].x = *reinterpret_cast<const f32x4*>(rb + nc[tn]);
          t[tn] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + (size_t)m * p.ldr + nc[tn]);
        }
#pragma unroll
        for (int tn = 0; tn < TB; ++tn) ops[tn].x += t[tn];
      } else if (rb) {
#pragma unroll
        for (int tn = 0; tn < TB; ++tn) ops[tn].x = *reinterpret_cast<const f32x4*>(rb + nc[tn]);
      } else if (r32) {
#pragma unroll
        for (int tn = 0; tn < TB; ++tn) ops[tn].x = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + (size_t)m * p.ldr + nc[tn]);
      }
      if (p.R && !p.r_f32) {
#pragma unroll
        for (int tn = 0; tn < TB; ++tn) ops[tn].r16 = *reinterpret_cast<const u32x2*>(p.R + (size_t)m * p.ldr + nc[tn]);
      }
    }
    store_row<TN>(p, crow, n_wave, lq, [&](int tn, int n) {
      if (!p.geglu) return epi4<WSH, BR>(p, acc[tn][tm], m, n, rb, gt, batched, ops[TN <= 6 ? tn : 0], wsc[WSH ? tn : 0], bsr[BR ? tn : 0]);
      f32x4 v = acc[tn][tm];   // GEGLU halves: fp8 scale and bias only (launch_gemm rejects the other operands)
      if constexpr (WSH) v *= wsc[tn];
      else if (p.wscale) v *= *reinterpret_cast<const f32x4*>(p.wscale + n);
      if constexpr (BR) v += bsr[tn];
      else if (p.bias && !p.bias_acc) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      return v;
    });
  }
}

// ---- residual fetched EARLY (gemm_pipe.hip, residual launches of the register-pipelined tiles) ---------------------------------
// scripts/gemm_timeline.py on the K = 1280 residual projections (8192 x 1280 x 1280 + R, 192 launches per SDXL step): K loop 22 us,
// epilogue 12.6 us -- every CU leaves its K loop at the same moment and only THEN starts fetching its 80 KB of residual, 20 dependent
// load -> wait -> add -> store rounds per wave with nothing else on the chip to overlap them. The residual values a lane needs are
// known from the start, so they are requested during the last K iterations -- one row-tile per iteration, issued right after that
// iteration's LDS-DMA so that the loop's counted vmcnt (gemm_pipe.hip) lets them stay in flight for two iterations -- and the
// epilogue starts with its operands on chip (the bias is in the accumulators already, GemmArgs::bias_acc): 12.6 -> 4.1 us.
// WHERE they land: in v216 .. v255, registers the compiler never allocates in these kernels (they are built with
// amdgpu_num_vgpr(216); the clobber lists make the kernel's register count include them). Four other forms failed first: plain loads under the run-time row index came out as load -> s_waitcnt -> copy inside the loop (phis of the landing
// registers are not coalesced); the row iterations peeled off the loop made the allocator spill the fresh values behind a
// vmcnt(0); inline-asm loads tied to C variables ("+v") were COPIED right after issue -- to the compiler an asm output is ready
// -- i.e. before the data arrived (NaN in the first hardware run); a-registers as the landing zone made hipcc split the register
// budget 128 + 128 and shuttle the accumulators through v_accvgpr moves. Registers the allocator never sees have none of these
// problems: the loads are asm with the registers named in the text, nothing touches them until gemm_epilogue_pre has waited,
// then a v_mov per register moves each row-tile's 10 registers out right before use. 16 bits per value: 40 registers per lane at
// TM x TN = 4 x 5.
#define SD_PRE_CLOBBERS                                                                                                          \
  "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231",   \
      "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", \
      "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
constexpr int SD_PRE_BASE = 216;   // first landing register; the PRE kernels are compiled with amdgpu_num_vgpr(216)
template <int TM, int TN>
struct EpiPre {
  static constexpr int PAIRS = TN / 2;
  static constexpr int RPR = PAIRS * 4 + (TN & 1) * 2;   // a-registers per row-tile: pair h at 4h .. 4h+3, the unpaired sub-tile last
  static constexpr int LOADS_PER_ROW = PAIRS + (TN & 1);
  static_assert(TM * RPR <= 40, "the landing zone is v216 .. v255");
};

// loads of row-tile TMI. One per-lane offset (the lane's row and first channel), the sub-tile pair as an immediate: rows >= M are
// sent out of the descriptor's range and read as zero; channels past N in a ragged last column tile read whatever follows in the
// row (never stored) or zero beyond the buffer's end.
template <int TMI, int TM, int TN>
__device__ __forceinline__ void epi_prefetch_row(const GemmArgs& p, u32x4 srd, int m_wave, int n_wave, int lane) {
  using P = EpiPre<TM, TN>;
  constexpr unsigned OOB = 0xFFFFF000u;
  const int m = m_wave + TMI * 16 + (lane & 15);
  const unsigned voff = m < p.M ? (unsigned)(((size_t)m * p.ldr + n_wave + (lane >> 4) * 8) * 2) : OOB;
#pragma unroll
  for (int h = 0; h < P::PAIRS; ++h)
    asm volatile("buffer_load_dwordx4 v[%c2:%c3], %0, %1, 0 offen offset:%c4"
                 :
                 : "v"(voff), "s"(srd), "n"(SD_PRE_BASE + TMI * P::RPR + 4 * h), "n"(SD_PRE_BASE + TMI * P::RPR + 4 * h + 3), "n"(h * 64)
                 : "memory", SD_PRE_CLOBBERS);
  if constexpr (TN & 1) {   // the unpaired sub-tile: 4 channels per lane group at (TN-1)*16 + lq*4 = the pair offset - lq*8 bytes
    const unsigned v4 = m < p.M ? voff + (TN - 1) * 32 - (unsigned)(lane >> 4) * 8u : OOB;
    asm volatile("buffer_load_dwordx2 v[%c2:%c3], %0, %1, 0 offen"
                 :
                 : "v"(v4), "s"(srd), "n"(SD_PRE_BASE + TMI * P::RPR + 4 * P::PAIRS), "n"(SD_PRE_BASE + TMI * P::RPR + 4 * P::PAIRS + 1)
                 : "memory", SD_PRE_CLOBBERS);
  }
}
// the descriptor of the residual rows as four SGPR words (base, stride 0, bytes, raw-buffer flags) for the asm loads
__device__ __forceinline__ u32x4 epi_r_srd(const GemmArgs& p) {
  const unsigned long long base = reinterpret_cast<unsigned long long>(p.R);
  return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)base), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xFFFFu),
               (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(((size_t)(p.M - 1) * p.ldr + p.N) * 2)), 0x00020000u};
}
template <int IDX>
__device__ __forceinline__ unsigned epi_pre_read() {
  unsigned v;
  asm volatile("v_mov_b32 %0, v%c1" : "=v"(v) : "n"(SD_PRE_BASE + IDX));
  return v;
}

// gemm_epilogue for a launch whose residual sits in v216.. (bf16 residual, bias in the accumulators, no GEGLU / fp8 scale / gate:
// launch_pipe checks)
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_pre(const GemmArgs& p, f32x4 (&acc)[TN][TM], int m_wave, int n_wave, int lane) {
  using P = EpiPre<TM, TN>;
  const int lq = lane >> 4;
  // every early load has landed (volatile asm keeps its order). `s_mov_b32 m0, m0` is a no-op the compiler never emits: the MARKER by
  // which scripts/check_landing_zone.py recognises, in the built library, every kernel that uses the landing zone -- whatever its name
  asm volatile("s_waitcnt vmcnt(0)\n\ts_mov_b32 m0, m0" ::: "memory", SD_PRE_CLOBBERS);
  auto row = [&](auto tm_tag) {
    constexpr int tm = decltype(tm_tag)::value;
    // this row-tile's residual, 2 registers per 4-channel group (asm volatile: issued after the wait above, in order)
    u32x2 r[TN];
    auto rd = [&](auto tn_tag) {
      constexpr int tn = decltype(tn_tag)::value;
      constexpr int a = tm * P::RPR + ((TN & 1) && tn == TN - 1 ? 4 * P::PAIRS : 4 * (tn >> 1) + 2 * (tn & 1));
      r[tn] = u32x2{epi_pre_read<a>(), epi_pre_read<a + 1>()};
    };
    rd(std::integral_constant<int, 0>{});
    if constexpr (TN > 1) rd(std::integral_constant<int, (TN > 1 ? 1 : 0)>{});
    if constexpr (TN > 2) rd(std::integral_constant<int, (TN > 2 ? 2 : 0)>{});
    if constexpr (TN > 3) rd(std::integral_constant<int, (TN > 3 ? 3 : 0)>{});
    if constexpr (TN > 4) rd(std::integral_constant<int, (TN > 4 ? 4 : 0)>{});
    if constexpr (TN > 5) rd(std::integral_constant<int, (TN > 5 ? 5 : 0)>{});
    static_assert(TN <= 6, "six sub-tiles per wave");
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m >= p.M) return;
    const float* rb = p.rowbias ? p.rowbias + (size_t)(m / p.rows_per_batch) * p.ld_rowbias : nullptr;
    const size_t crow = p.c_rpb ? (size_t)(m / p.c_rpb) * p.c_bstride + (size_t)(m % p.c_rpb) * p.ldc : (size_t)m * p.ldc;
    store_row<TN>(p, crow, n_wave, lq, [&](int tn, int n) {
      f32x4 v = acc[tn][tm];
      if (rb) v += *reinterpret_cast<const f32x4*>(rb + n);
      return act4(p, add_r16(v, r[tn]) * p.out_scale);
    });
  };
  row(std::integral_constant<int, 0>{});
  if constexpr (TM > 1) row(std::integral_constant<int, (TM > 1 ? 1 : 0)>{});
  if constexpr (TM > 2) row(std::integral_constant<int, (TM > 2 ? 2 : 0)>{});
  if constexpr (TM > 3) row(std::integral_constant<int, (TM > 3 ? 3 : 0)>{});
  if constexpr (TM > 4) row(std::integral_constant<int, (TM > 4 ? 4 : 0)>{});
  if constexpr (TM > 5) row(std::integral_constant<int, (TM > 5 ? 5 : 0)>{});
  static_assert(TM <= 6, "six row-tiles per wave");
}

// Epilogue of the LayerNorm-folded projections (mi355x_sd_linear_ln): acc <- rstd[m] * acc - mean[m] * rstd[m] * wsum[n]
// + bias[n], then GEGLU / SiLU / tanh-GELU and the store. A separate function (and separate kernel instantiations,
// template parameter LN) so the register allocation of every other GEMM is unaffected. The row statistics of all TM
// row-tiles are fetched up front with clamped, branch-free addresses.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_ln(const GemmArgs& p, f32x4 (&acc)[TN][TM], int m_wave, int n_wave,
                                                 int lane) {
  const int lq = lane >> 4;
  f32x2 rs[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
    rs[tm] = *reinterpret_cast<const f32x2*>(p.rowstat + 2 * (size_t)min(m_wave + tm * 16 + (lane & 15), p.M - 1));
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    store_row<TN>(p, (size_t)m * p.ldc, n_wave, lq, [&](int tn, int n) {
      f32x4 v = rs[tm][0] * acc[tn][tm] + rs[tm][1] * *reinterpret_cast<const f32x4*>(p.wsum + n);
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      return p.geglu ? v : act4(p, v);
    });
  }
}

// Epilogue of the W8A8 GEMM (mi355x_sd_linear_f8 / _f8_q): acc * ascale[m] * wscale[n] + bias, optional gate / residual /
// tanh-GELU / row-remapped C; bf16 or e4m3 store. Own function + own kernel instantiation (see gemm_epilogue_ln).
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_f8(const GemmArgs& p, f32x4 (&acc)[TN][TM], int m_wave, int n_wave,
                                                 int lane) {
  static_assert(TN % 2 == 0, "fp8 GEMM tiles pair their n sub-tiles");
  const int lq = lane >> 4;
  float as[TM], oinv[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int mc = min(m_wave + tm * 16 + (lane & 15), p.M - 1);
    as[tm] = p.ascale[mc];
    oinv[tm] = 0.f;
    if (p.out_f8) {   // safe row scale for the e4m3 output (1.1: the quantised operands may exceed their fp32 norms slightly)
      const float bound = 1.1f * (p.a_l2[mc] * p.w_norm_max + p.bias_abs_max);
      oinv[tm] = 448.0f / fmaxf(bound, 1e-12f);
      if (n_wave == 0 && lq == 0 && m_wave + tm * 16 + (lane & 15) < p.M) p.oscale[mc] = fmaxf(bound, 1e-12f) * (1.0f / 448.0f);
    }
  }
  // per-channel operands of this lane's TN channel groups, fetched once; per row-tile the gate and the residual of all groups are
  // requested back to back before the first is used (round 5: the loads used to sit inside fin(), TM x TN dependent round trips)
  f32x4 wsc[TN], bs[TN];
  int nc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    nc[tn] = min(n_wave + (tn >> 1) * 32 + lq * 8 + (tn & 1) * 4, p.N - 4);
    wsc[tn] = *reinterpret_cast<const f32x4*>(p.wscale + nc[tn]);
    bs[tn] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nc[tn]) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m_wave + tm * 16 + (lane & 15);
    if (m >= p.M) continue;
    const float* gt = p.gate ? p.gate + (size_t)(m / p.rows_per_batch) * p.ld_gate : nullptr;
    const size_t crow = p.c_rpb ? (size_t)(m / p.c_rpb) * p.c_bstride + (size_t)(m % p.c_rpb) * p.ldc : (size_t)m * p.ldc;
    f32x4 g4[TN];
    u32x2 r4[TN];
    if (gt) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) g4[tn] = *reinterpret_cast<const f32x4*>(gt + nc[tn]);
    }
    if (p.R) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) r4[tn] = *reinterpret_cast<const u32x2*>(p.R + (size_t)m * p.ldr + nc[tn]);
    }
    auto fin = [&](int tn, int) {
      f32x4 v = acc[tn][tm] * as[tm] * wsc[tn];
      if (p.bias) v += bs[tn];
      if (gt) v *= g4[tn];
      if (p.R) v = add_r16(v, r4[tn]);
      return act4(p, v);
    };
    auto q4 = [&](f32x4 v) {   // 4 x e4m3 with the row's output scale
      v *= oinv[tm];
      int q = 0;
      q = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], q, false);
      q = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], q, true);
      return (unsigned)q;
    };
#pragma unroll
    for (int h = 0; h < TN / 2; ++h) {   // one loop for both output types: two loops over the accumulators make the
      const int n = n_wave + h * 32 + lq * 8;   // register allocator split their live ranges and spill in the K loop
      if (n >= p.N) continue;
      const bool two = n + 4 < p.N;
      const f32x4 lo = fin(2 * h, n), hi = two ? fin(2 * h + 1, n + 4) : lo;
      if (p.out_f8) {   // 8 e4m3 bytes per lane (ldc % 8, no gate / residual: launch_gemm_f8)
        unsigned char* dst = reinterpret_cast<unsigned char*>(p.C) + crow + n;
        if (two) *reinterpret_cast<u32x2*>(dst) = u32x2{q4(lo), q4(hi)};
        else *reinterpret_cast<unsigned*>(dst) = q4(lo);
      } else if (two) {
        store8(p, crow + n, lo, hi);
      } else {
        store4(p, crow + n, lo);
      }
    }
  }
}

}  // namespace sd
