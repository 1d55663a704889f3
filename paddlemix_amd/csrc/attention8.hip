// Eight-wave scaled-dot-product attention for head_dim 64 (the SDXL / SD3 self-attention shapes), gfx950.
//
// Same arithmetic and layouts as attention.hip (swapped S^T = K Q^T and O^T = V^T P^T on 32x32x16 MFMA, softmax rows
// lane-local, exp2-domain online softmax with a deferred rescale; reference semantics: attention_processor.py:1167-1249,
// paddle_patch.py:445-461), restructured the way MI355X_MICROARCH.md "Two waves per SIMD" describes:
//
//   * a block is 8 waves x 32 query rows = 256 queries against K/V tiles of 64 keys; the K/V tile is staged ONCE for the
//     256 queries (half the staging work per query of the four-wave kernel) and by LDS-DMA (buffer_load ... lds, 1 KiB per
//     wave-instruction: no staging registers, no ds_write pass) into a ring of four 16-KiB stages, two tiles ahead;
//   * the DMA writes LDS lane-linearly, so the bank-conflict-free images are built on the SOURCE side: lane i of the piece
//     that fills rows 8w..8w+7 fetches 16-byte chunk (i&7) ^ f(row) of row 8w + (i>>3);  K: f = (row>>1)&7 (the 16 rows of a
//     ds_read_b128 lane group then cover all 16 chunk positions of the 256-byte bank row), V: f = ((row>>1)&1)<<2 (the 4 key
//     rows of a ds_read_b64_tr_b16 block land on disjoint quarters of the 64 banks);
//   * the two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run HALF A TILE APART, two barriers per tile: while one
//     group is in its matrix interval [P.V of tile t-1, K.Q^T of tile t] the other is in its VALU interval [softmax of tile t-1],
//     so a SIMD's matrix pipe and its VALU issue slots are claimed by different waves instead of the sum of both by each;
//   * waits: the only VMEM wait in the loop is one vmcnt(0) per tile, two intervals after the pieces were issued and just
//     before the next ones go out; barriers are raw s_barrier (no vmcnt drain).
//
// Hazards. RAW: a wave waits for its pieces of tile t+1 (vmcnt(0)) before barrier B1 of its iteration t; the trailing group's
// B1 of iteration t is the barrier in front of the leading group's K.Q^T of tile t+1, the first reader. WAR: tile t+2 goes to
// stage (t+2)&3, last read as tile t-2; the trailing group's P.V of tile t-2 ends one interval before the leading group issues
// tile t+2, with a barrier in between. Fragment reads of a stage are consumed by MFMAs of the same interval.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemm_cfg.h"        // wait_vmcnt_imm
#include "gemm_epilogue.h"   // lptr_t
#include "kernels.h"

namespace sd {

constexpr int A8_WAVES = 8;
constexpr int A8_THREADS = A8_WAVES * 64;
constexpr int A8_QROWS = 32;                        // per wave
constexpr int A8_QBLK = A8_WAVES * A8_QROWS;        // 256 per block
constexpr int A8_KV = 64;
constexpr int A8_STAGES = 4;
constexpr int A8_TILE = A8_KV * 128;                // bytes of a K (or V) tile: 64 rows x 64 bf16
constexpr int A8_STAGE = 2 * A8_TILE;               // K then V

__device__ __forceinline__ void a8_dma(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)lds, 16, voff, 0, 0, 0);
}
#define A8_BARRIER()                      \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)

// MODE bit 0: static priority for the trailing group (cdna guide T5, static form); bit 1: no group stagger (ablation);
// bit 3: lazy row maximum -- the tile's scores are exponentiated against the RUNNING maximum without looking for the tile's
// own maximum first (that search is ~20 of the ~135 VALU instructions of a tile); the running maximum only has to keep
// exp2 inside the fp32 range, so the exact path (maximum, rescale, exponentiate again) runs on the first tile and whenever a
// lane's partial row sum leaves [0, 2^60] -- P and the accumulators are floating point, a stale reference costs no precision
// bit 4 (needs bit 3): scores are base-2 exponents already (AttnArgs::log2, see attention.hip FL bit 1)
template <int MODE>
__global__ __launch_bounds__(A8_THREADS, 2) void attention8_kernel(const AttnArgs p) {
  constexpr bool LOG2 = (MODE & 16) != 0;
  static_assert(!LOG2 || (MODE & 8), "LOG2 builds on the lazy reference");
  __shared__ __attribute__((aligned(16))) unsigned char smem[A8_STAGES * A8_STAGE];
  constexpr int KS = 4;   // k-steps of K.Q^T (d = 64)
  constexpr int DB = 2;   // 32-row blocks of O^T

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int hi = lane >> 5;
  const int lq = lane & 31;

  const int nqb = (p.Sq + A8_QBLK - 1) / A8_QBLK;
  const int lid = xcd_remap(blockIdx.x, nqb * p.B * p.H);
  const int qb = lid % nqb;
  const int bh = lid / nqb;
  const int b = bh / p.H, h = bh - b * p.H;

  const bf16* Qp = p.Q + (size_t)b * p.q_bs + (size_t)h * 64;
  const bf16* Kp = p.K + (size_t)b * p.k_bs + (size_t)h * 64;
  const bf16* Vp = p.V + (size_t)b * p.v_bs + (size_t)h * 64;
  bf16* Op = p.O + (size_t)b * p.o_bs + (size_t)h * 64;

  // ---- LDS-DMA geometry: this wave owns piece `wave` of every tile = tile rows 8*wave .. 8*wave+7 of K and of V ----
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(Kp), 0, (unsigned)(((size_t)(p.Skv - 1) * p.k_ts + 64) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(Vp), 0, (unsigned)(((size_t)(p.Skv - 1) * p.v_ts + 64) * 2), 0x00020000);
  const int prow = wave * 8 + (lane >> 3);   // tile row this lane fetches
  const int slot = lane & 7;                 // 16-byte LDS slot it lands in
  const unsigned k_voff = (unsigned)((prow * p.k_ts + ((slot ^ ((prow >> 1) & 7)) << 3)) * 2);
  const unsigned v_voff = (unsigned)((prow * p.v_ts + ((slot ^ (((prow >> 1) & 1) << 2)) << 3)) * 2);
  const unsigned k_tstep = (unsigned)(A8_KV * p.k_ts * 2), v_tstep = (unsigned)(A8_KV * p.v_ts * 2);
  // the tile offset rides in the per-lane offset (range-checked by the descriptor): rows >= Skv read as zeros
  auto issue = [&](const int t) {
    unsigned char* st = smem + (t & (A8_STAGES - 1)) * A8_STAGE + wave * 1024;
    a8_dma(k_rsrc, st, k_voff + (unsigned)t * k_tstep);
    a8_dma(v_rsrc, st + A8_TILE, v_voff + (unsigned)t * v_tstep);
  };

  const int ntiles = (p.Skv + A8_KV - 1) / A8_KV;
  const int nfull = p.Skv / A8_KV;
  issue(0);
  if (ntiles > 1) issue(1);

  // Q fragments (MFMA B operand): lane (q = lq, hi) holds d = ks*16 + hi*8 .. +8
  const int q_row = qb * A8_QBLK + wave * A8_QROWS + lq;
  const bool q_ok = q_row < p.Sq;
  bf16x8 qf[KS];
  {
    const bf16* qr = Qp + (size_t)(q_ok ? q_row : 0) * p.q_ts;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (q_ok) v = *reinterpret_cast<const u32x4*>(qr + ks * 16 + hi * 8);
      qf[ks] = *reinterpret_cast<bf16x8*>(&v);
    }
  }

  f32x16 o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY;   // running max of raw scores, both half-waves agree
  float l_run = 0.f;         // this half-wave's partial row sum
  const float c2 = LOG2 ? 1.0f : p.scale * 1.4426950408889634f;
  constexpr float RESCALE_THR = 4.0f;
  f32x16 sinit;              // LOG2: -reference in every element
#pragma unroll
  for (int r = 0; r < 16; ++r) sinit[r] = 0.f;

  // fragment-read geometry
  //   K: lane (lq, hi) reads row sb*32 + lq, chunk ks*2 + hi at slot chunk ^ ((row>>1)&7)        (row>>1 = sb*16 + (lq>>1))
  const int k_row_off = lq * 128;
  const int k_sw = (lq >> 1) & 7;
  //   V (transpose read): 16-lane group (hi, dh) reads the [4 keys][16 d] block; lane address = key row 4*hi + (i16>>2),
  //   d = db*32 + dh*16 + (i16&3)*4 -> chunk db*4 + dh*2 + ((i16&3)>>1), byte (i16&1)*8 inside it; slot = chunk ^ (((row>>1)&1)<<2)
  const int i16 = lane & 15;
  const int dh = (lane >> 4) & 1;
  const int v_r = 4 * hi + (i16 >> 2);                  // + kk*16 (+8): bit 1 of the row = bit 1 of v_r
  const int v_ch = dh * 2 + ((i16 & 3) >> 1);           // + db*4
  const int v_sw = ((v_r >> 1) & 1) << 2;
  const int v_off0 = v_r * 128 + (((v_ch) ^ v_sw) << 4) + (i16 & 1) * 8;          // db = 0
  const int v_off1 = v_r * 128 + (((v_ch + 4) ^ v_sw) << 4) + (i16 & 1) * 8;      // db = 1

  auto xhalf_max = [](float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  };
  auto max3 = [](float a, float b2, float c) { return fmaxf(fmaxf(a, b2), c); };

  // landed: tiles 0 and 1 (prologue), published to every wave
  wait_vmcnt_imm<0>();
  A8_BARRIER();
  if (!(MODE & 2) && grp == 1) A8_BARRIER();                      // the trailing group runs one interval behind
  if ((MODE & 1) && grp == 1) __builtin_amdgcn_s_setprio(1);      // and would lose every VALU arbitration by age otherwise

  f32x16 s[2];
  bf16x8 pf[4];

  auto qk = [&](const int t) {
    const unsigned char* ks_ = smem + (t & (A8_STAGES - 1)) * A8_STAGE;
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      if (LOG2) {
        s[sb] = sinit;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[sb][r] = 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks_ + sb * 4096 + k_row_off + (((ks * 2 + hi) ^ k_sw) << 4));
        s[sb] = mfma_32x32x16(kf, qf[ks], s[sb]);
      }
    }
  };

  auto softmax = [&](const int t, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    auto apply_mask = [&]() {
      const int kv0 = t * A8_KV;
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (kv >= p.Skv) s[sb][r] = -INFINITY;
        }
    };
    if (MASK) apply_mask();
    const float mc0 = m_run * c2;
    bool exact = true;
    float psum = 0.f;
    if ((MODE & 8) && t > 0) {   // lazy: exponentiate against the running maximum, verify afterwards
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const float e = LOG2 ? __builtin_amdgcn_exp2f(s[i >> 4][i & 15]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[i >> 4][i & 15], c2, -mc0));
        psum += e;
        pf[i >> 3][i & 7] = (bf16)e;
      }
      exact = __any(!(psum <= 0x1p60f));   // overflow (or NaN): scores again (the K tile is still staged), exact path
      if (exact) {
        asm volatile("; lazy-maximum overflow guard fired");
        qk(t);
        if (MASK) apply_mask();
      }
    }
    if (exact) {
      if (MODE & 8) asm volatile("; exact softmax path (first tile / overflow guard)");
      float mx[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) mx[i] = max3(s[(3 * i) >> 4][(3 * i) & 15], s[(3 * i + 1) >> 4][(3 * i + 1) & 15],
                                                s[(3 * i + 2) >> 4][(3 * i + 2) & 15]);
      float mloc = max3(max3(mx[0], mx[1], mx[2]), max3(mx[3], mx[4], mx[5]), max3(mx[6], mx[7], mx[8]));
      mloc = max3(mloc, mx[9], fmaxf(s[1][14], s[1][15]));
      mloc = xhalf_max(mloc);
      float shift = 0.f;
      if (LOG2) {   // scores are relative to the reference m_run (0 on the first tile)
        const bool first = t == 0;
        if (first || !__all(mloc <= RESCALE_THR)) {
          asm volatile("; online-softmax rescale (rare)");
          const float mfin = (mloc == -INFINITY) ? 0.f : mloc;
          const float delta = first ? mfin : fmaxf(mfin, 0.f);
          const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
          l_run *= alpha;
          m_run = (first ? 0.f : m_run) + delta;
          shift = delta;
#pragma unroll
          for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) sinit[r] = -m_run;
        }
      } else if (!__all((mloc - m_run) * c2 <= RESCALE_THR)) {
        asm volatile("; online-softmax rescale (rare)");
        const float m_new = fmaxf(m_run, mloc);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * c2);   // m_run = -inf -> 0
        l_run *= alpha;
        m_run = m_use;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      }
      const float mc = LOG2 ? shift : m_run * c2;
      psum = 0.f;
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sb][r], c2, -mc));
          psum += e;
          pf[sb * 2 + (r >> 3)][r & 7] = (bf16)e;
        }
      }
    }
    l_run += psum;
  };

  auto pv = [&](const int t) {
    const unsigned char* vs_ = smem + (t & (A8_STAGES - 1)) * A8_STAGE + A8_TILE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      // B-operand element i of pf[kk] is key kk*16 + (i&3) + 8*(i>>2) + 4*hi -> rows (kk*16 + v_r) and (+8); +8 and +16
      // rows leave bit 1 of the row (the swizzle bit) unchanged
      const unsigned char* vb = vs_ + kk * 2048;
      {
        const bf16x4 lo = ds_read_tr16((lds_bf16x4*)(vb + v_off0));
        const bf16x4 h4 = ds_read_tr16((lds_bf16x4*)(vb + 1024 + v_off0));
        const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], h4[0], h4[1], h4[2], h4[3]};
        o[0] = mfma_32x32x16(vf, pf[kk], o[0]);
      }
      {
        const bf16x4 lo = ds_read_tr16((lds_bf16x4*)(vb + v_off1));
        const bf16x4 h4 = ds_read_tr16((lds_bf16x4*)(vb + 1024 + v_off1));
        const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], h4[0], h4[1], h4[2], h4[3]};
        o[1] = mfma_32x32x16(vf, pf[kk], o[1]);
      }
    }
  };

  // ---- main loop: [K.Q^T(t)] B1 [softmax(t)] B2 [P.V(t) ; K.Q^T(t+1)] B1 ... ----
  for (int t = 0; t < ntiles; ++t) {
    qk(t);
    wait_vmcnt_imm<0>();                      // this wave's pieces of tile t+1 (issued two intervals ago)
    if (t + 2 < ntiles) issue(t + 2);
    A8_BARRIER();                             // B1
    if (t < nfull) softmax(t, std::false_type{});
    else softmax(t, std::true_type{});
    A8_BARRIER();                             // B2
    pv(t);
  }
  if ((MODE & 1) && grp == 1) __builtin_amdgcn_s_setprio(0);

  // ---- finalize: O[q][d] = O^T[d][q] / l ; lane holds d = db*32 + (r&3) + 8*(r>>2) + 4*hi ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv_l = (l_tot > 0.f) ? 1.0f / l_tot : 0.f;
  if (q_ok) {
    bf16* orow = Op + (size_t)q_row * p.o_ts;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int d = db * 32 + 8 * c + 4 * hi;
        float v0 = o[db][4 * c + 0] * inv_l, v1 = o[db][4 * c + 1] * inv_l;
        float v2 = o[db][4 * c + 2] * inv_l, v3 = o[db][4 * c + 3] * inv_l;
        if (p.accum != 0.f) {   // launch-uniform: add to what the first attention call left in O
          const bf16x4 old = __builtin_bit_cast(bf16x4, *reinterpret_cast<const u32x2*>(orow + d));
          v0 = (float)old[0] + p.accum * v0;
          v1 = (float)old[1] + p.accum * v1;
          v2 = (float)old[2] + p.accum * v2;
          v3 = (float)old[3] + p.accum * v3;
        }
        u32x2 pk = {pack_bf16(v0, v1), pack_bf16(v2, v3)};
        *reinterpret_cast<u32x2*>(orow + d) = pk;
      }
  }
}

// D == 64, no additive mask, 16-byte aligned K / V rows (the LDS-DMA moves 16-byte chunks). mode: MI355X_SD_ATTN8 (see launch_attention)
int launch_attention8(const AttnArgs& a, int mode, hipStream_t stream) {
  if (a.D != 64 || a.bias) return SD_ERR_UNSUPPORTED;
  if (a.log2) mode |= 8 | 16;
  if ((a.k_ts & 7) || (a.v_ts & 7) || (a.k_bs & 7) || (a.v_bs & 7) || (reinterpret_cast<uintptr_t>(a.K) & 15) ||
      (reinterpret_cast<uintptr_t>(a.V) & 15))
    return SD_ERR_UNSUPPORTED;
  if (((size_t)(a.Skv - 1) * a.k_ts + 64) * 2 >= 0xFFFFFFF0ull || ((size_t)(a.Skv - 1) * a.v_ts + 64) * 2 >= 0xFFFFFFF0ull)
    return SD_ERR_UNSUPPORTED;   // 32-bit buffer offsets
  const int nqb = (a.Sq + A8_QBLK - 1) / A8_QBLK;
  dim3 grid(nqb * a.B * a.H), block(A8_THREADS);
  if (mode & 16) {
    if (mode & 1) hipLaunchKernelGGL(attention8_kernel<25>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(attention8_kernel<24>, grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
  }
  switch (mode & 11) {
    case 0: hipLaunchKernelGGL(attention8_kernel<0>, grid, block, 0, stream, a); break;
    case 1: hipLaunchKernelGGL(attention8_kernel<1>, grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL(attention8_kernel<2>, grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL(attention8_kernel<3>, grid, block, 0, stream, a); break;
    case 8: hipLaunchKernelGGL(attention8_kernel<8>, grid, block, 0, stream, a); break;
    case 9: hipLaunchKernelGGL(attention8_kernel<9>, grid, block, 0, stream, a); break;
    case 10: hipLaunchKernelGGL(attention8_kernel<10>, grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL(attention8_kernel<11>, grid, block, 0, stream, a); break;
  }
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

}  // namespace sd
