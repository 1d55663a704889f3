// Fused scaled-dot-product attention (flash-style, online softmax) for gfx950, bf16 in / fp32 accumulate.
//
// Replaces the math reached from
//   AttnProcessor.__call__ / Attention.get_attention_scores   ppdiffusers/ppdiffusers/models/attention_processor.py:673-735, 552-586
//   XFormersAttnProcessor.__call__                             attention_processor.py:1167-1249
//   scaled_dot_product_attention_ (math branch)                ppdiffusers/ppdiffusers/patches/paddle_patch.py:445-461
// i.e. out = softmax(q k^T * scale + mask) v, in the sdpa_ layout q [B,Sq,h,d], k/v [B,Skv,h,d] (arbitrary
// token / batch strides so a fused QKV projection buffer is consumed in place).
//
// Design (wave64, MFMA 32x32x16):
//   * block = 4 waves x 32 query rows; K and V tiles of 64 keys are staged global -> registers -> LDS,
//     double buffered, the next tile's loads in flight during the current tile's MFMAs;
//   * S^T = K Q^T ("swapped" QK^T): the 32x32 accumulator then has one query per lane column, so the
//     softmax row statistics are lane-local except for one exchange between the two half-waves;
//   * O^T = V^T P^T: P (bf16) is consumed straight from those registers as the MFMA B operand; the matching
//     V^T A operand comes from ds_read_b64_tr_b16 transpose reads of the row-major V tile;
//   * K rows padded by 16 B and V rows to a stride == 64 (mod 256) B so both fragment reads are bank-conflict free.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace sd {

constexpr int ATTN_LAZY_DEFAULT = 1;
// Lazy-maximum guard (FL bit 0): a lane's partial row sum of the freshly exponentiated tile must stay below this, else the tile
// is redone the exact way. Every P value of the lane is <= that sum, and P is stored in the build's 16-bit type: bf16 shares
// fp32's exponent range (any finite value converts), IEEE half tops out at 65504 -- a score more than ~16 log2 units above the
// stale reference would become +inf in P and NaN in O. 2^15 keeps P finite in the half build with a 2x margin.
#ifdef MI355X_SD_F16
constexpr float LAZY_PSUM_LIMIT = 0x1p15f;
#else
constexpr float LAZY_PSUM_LIMIT = 0x1p60f;
#endif
constexpr int ATT_WAVES = 4;
constexpr int ATT_THREADS = ATT_WAVES * 64;
constexpr int QROWS = 32;                   // per wave
constexpr int QBLK = ATT_WAVES * QROWS;     // per block
constexpr int KVBLK = 64;

template <int DP>
struct AttLds {
  static constexpr int KRS = DP * 2 + 16;                            // K row stride (bytes)
  static constexpr int VRS = (DP == 64) ? 192 : DP * 2;              // V row stride: 192 / 192 / 320 B
  static constexpr int KBYTES = KVBLK * KRS;
  static constexpr int VBYTES = KVBLK * VRS;
  static constexpr int CHUNKS = KVBLK * (DP / 8) / ATT_THREADS;      // 16-B chunks per thread per tile: 2 / 3 / 5
};

// QT: query tiles per block (1; 2 for the short-KV cross-attention launches, see the query-tile loop at the bottom)
// FL bit 0 (LAZY): the tile's scores are exponentiated against the RUNNING row maximum without first searching the tile's own
//   maximum (~20 of the ~135 VALU instructions of a tile); the reference only has to keep exp2 inside the fp32 range -- P and
//   the accumulators are floating point, a stale reference costs no precision -- so the exact path (maximum, rescale,
//   exponentiate again from the intact scores) runs on the first tile and whenever a lane's partial row sum leaves
//   [0, LAZY_PSUM_LIMIT] (2^60; 2^15 in the IEEE-half build, whose P cannot hold more than 65504).
// FL bit 1 (LOG2, needs LAZY, no mask): the caller folded scale * log2(e) into the queries (the UNet builder folds it into
//   the to_q weights), so a score IS the exponent: the S^T accumulators start at -reference instead of 0 (a persistent
//   16-register C operand, rewritten only at a rescale) and exp2 needs no multiply-add per score.
// FL bit 2 (WIDE): O leaves as 16-byte stores (the launcher checked: rows 16-byte aligned, D % 16 == 0, no accumulation): the
//   half-waves hold adjacent 8-byte pieces of a row, one v_permlane32_swap per dword pairs them up (cdna guide T21). Its own
//   instantiation: next to the 8-byte store loop in one kernel it cost 3 registers = the occupancy step at 168.
template <int DP, bool HAS_BIAS, int QT = 1, int FL = 0>
__global__ __launch_bounds__(ATT_THREADS, (DP <= 96) ? ((FL & 4) ? 3 : 2) : 1) void attention_kernel(const AttnArgs p) {
  constexpr bool LAZY = (FL & 1) != 0, LOG2 = (FL & 2) != 0, WIDE = (FL & 4) != 0;
  [[maybe_unused]] unsigned long long clk0 = 0, wall0 = 0;
  if constexpr ((FL & 8) != 0) {   // debug: shader clock over this block's life (see attention_il_kernel ABL bit 6)
    clk0 = __builtin_readcyclecounter();
    wall0 = wall_clock64();
  }
  static_assert(!WIDE || QT == 1, "wide stores: the one-query-tile form");
  static_assert(!LOG2 || (LAZY && !HAS_BIAS), "LOG2 builds on the lazy reference and has no additive mask");
  using L = AttLds<DP>;
  constexpr int NBUF = (DP <= 96) ? 2 : 1;
  constexpr int KS = DP / 16;   // k-steps of QK^T
  constexpr int DB = DP / 32;   // 32-row blocks of O^T
  __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * (L::KBYTES + L::VBYTES)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int lq = lane & 31;

  const int nqb = (p.Sq + QBLK * QT - 1) / (QBLK * QT);
  const int lid = xcd_remap(blockIdx.x, nqb * p.B * p.H);
  const int qb = lid % nqb;
  const int bh = lid / nqb;
  const int b = bh / p.H, h = bh - b * p.H;

  const bf16* Qp = p.Q + (size_t)b * p.q_bs + (size_t)h * p.D;
  const bf16* Kp = p.K + (size_t)b * p.k_bs + (size_t)h * p.D;
  const bf16* Vp = p.V + (size_t)b * p.v_bs + (size_t)h * p.D;
  bf16* Op = p.O + (size_t)b * p.o_bs + (size_t)h * p.D;

  // ---- K/V tile loader: thread owns CHUNKS 16-B chunks of the K tile and of the V tile.  Buffer (SRD) loads with a
  // per-lane 32-bit byte offset = lane part (never changes) + tile part; rows past Skv fall outside the descriptor's
  // range and read as zero in hardware (no branches, no 64-bit address arithmetic in the loop). The tile part is added
  // to the PER-LANE offset, not passed as soffset: the descriptor's range check covers voffset + inst_offset only, so a
  // scalar offset past the end would fetch whatever follows the tensor (NaN there -> NaN out through P.V with P = 0). ----
  u32x4 rk[L::CHUNKS], rv[L::CHUNKS];
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(Kp), 0, (unsigned)(((size_t)(p.Skv - 1) * p.k_ts + p.D) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(Vp), 0, (unsigned)(((size_t)(p.Skv - 1) * p.v_ts + p.D) * 2), 0x00020000);
  unsigned k_off[L::CHUNKS], v_off[L::CHUNKS];
#pragma unroll
  for (int i = 0; i < L::CHUNKS; ++i) {
    const int cid = tid + ATT_THREADS * i;
    const int row = cid / (DP / 8), ch = cid - row * (DP / 8);
    const bool col_ok = ch * 8 < p.D;   // padded head-dim columns: force out of range -> zeros
    k_off[i] = col_ok ? (unsigned)((row * p.k_ts + ch * 8) * 2) : 0xFFFFFFF0u;
    v_off[i] = col_ok ? (unsigned)((row * p.v_ts + ch * 8) * 2) : 0xFFFFFFF0u;
  }
  auto load_kv = [&](int kv0) {
    const unsigned ks_off = (unsigned)(kv0 * p.k_ts * 2), vs_off = (unsigned)(kv0 * p.v_ts * 2);
#pragma unroll
    for (int i = 0; i < L::CHUNKS; ++i) {
      // (padded head-dim columns carry the out-of-range marker: keep it out of range)
      const unsigned ko = (k_off[i] != 0xFFFFFFF0u) ? k_off[i] + ks_off : 0xFFFFFFF0u;
      const unsigned vo = (v_off[i] != 0xFFFFFFF0u) ? v_off[i] + vs_off : 0xFFFFFFF0u;
      rk[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, ko, 0, 0));
      rv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, vo, 0, 0));
    }
  };
  auto store_kv = [&](int buf) {
    unsigned char* ks_ = smem + buf * (L::KBYTES + L::VBYTES);
    unsigned char* vs_ = ks_ + L::KBYTES;
#pragma unroll
    for (int i = 0; i < L::CHUNKS; ++i) {
      const int cid = tid + ATT_THREADS * i;
      const int row = cid / (DP / 8), ch = cid - row * (DP / 8);
      *reinterpret_cast<u32x4*>(ks_ + row * L::KRS + ch * 16) = rk[i];
      *reinterpret_cast<u32x4*>(vs_ + row * L::VRS + ch * 16) = rv[i];
    }
  };

  // per-query-tile state (re-initialised by the q-tile loop at the bottom)
  int q_row = 0;
  bool q_ok = false;
  bool stage_kv = true;   // false on the 2nd.. query tile of a short-KV block: both K/V tiles are already resident in LDS
  bf16x8 qf[KS];
  f32x16 o[DB];
  float m_run = -INFINITY;   // running max of raw scores (q . k [+ bias/scale]), both half-waves agree
  float l_run = 0.f;         // this half-wave's partial row sum
  const float c2 = LOG2 ? 1.0f : p.scale * 1.4426950408889634f;  // exp(x*scale) = exp2(x*c2); LOG2: scores are exponents already
  f32x16 sinit;              // LOG2: -reference in every element (C operand of the first MFMA of each S^T chain)
#pragma unroll
  for (int r = 0; r < 16; ++r) sinit[r] = 0.f;

  // Q fragments (MFMA B operand) of query tile qt: lane (q = lq, hi) holds d = ks*16 + hi*8 .. +8
  auto load_q = [&](const int qt, bf16x8 (&dst)[KS]) {
    const int row = qt * QBLK + wave * QROWS + lq;
    const bool ok = row < p.Sq;
    const bf16* qr = Qp + (size_t)(ok ? row : 0) * p.q_ts;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 16 + hi * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok && d0 < p.D) v = *reinterpret_cast<const u32x4*>(qr + d0);
      dst[ks] = *reinterpret_cast<bf16x8*>(&v);
    }
  };

  const int ntiles = (p.Skv + KVBLK - 1) / KVBLK;
  load_q(qb * QT, qf);   // in flight together with the first K/V tile
  load_kv(0);
  store_kv(0);
  __syncthreads();

  // transpose-read lane geometry: 16-lane group (hi, dh) reads a [4 keys][16 d] block
  const int i16 = lane & 15;
  const int dh = (lane >> 4) & 1;
  const int tr_row = (i16 >> 2);           // + key base
  const int tr_col = dh * 16 + (i16 & 3) * 4;

  // cross-half exchange without touching the LDS pipe: v_permlane32_swap leaves {lo, lo} / {hi, hi} in the pair
  auto xhalf_max = [](float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  };
  constexpr float RESCALE_THR = 4.0f;   // log2 units: skip the O rescale while the row max grew by < 2^4 (P <= 16)

  // one K/V tile; MASK = the (single) ragged last tile, kept out of the steady-state instruction stream
  auto tile_body = [&](const int t, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int buf = (NBUF == 2) ? (t & 1) : 0;
    if (NBUF == 2 && t + 1 < ntiles && stage_kv) load_kv((t + 1) * KVBLK);
    const unsigned char* ks_ = smem + buf * (L::KBYTES + L::VBYTES);
    const unsigned char* vs_ = ks_ + L::KBYTES;

    // ---- S^T = K Q^T : two 32-key blocks (+ mask / bias). A lambda: the lazy softmax overwrites the scores in place and, in
    // the rare case its overflow guard fires, simply computes them again (the K tile is still in LDS) ----
    f32x16 s[2];
    const int kv0 = t * KVBLK;
    auto compute_scores = [&]() {
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
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks_ + (sb * 32 + lq) * L::KRS + (ks * 2 + hi) * 16);
        s[sb] = mfma_32x32x16(kf, qf[ks], s[sb]);
      }
    }

    // ---- mask / bias; s[sb][r] is key kv0 + sb*32 + (r&3) + 8*(r>>2) + 4*hi for query lq ----
    if (HAS_BIAS) {
      const float inv = 1.0f / p.scale;
      const float* bp = p.bias + (size_t)b * p.bias_bs + (size_t)h * p.bias_hs + (size_t)(q_ok ? q_row : 0) * p.bias_qs;
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (!MASK || kv < p.Skv) s[sb][r] += bp[kv] * inv;  // (s + bias/scale)*scale = s*scale + bias
        }
    }
    if (MASK) {
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + sb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (kv >= p.Skv) s[sb][r] = -INFINITY;
        }
    }
    };
    compute_scores();

    // ---- online softmax (one query per lane column) ----
    // 3-input max tree (v_max3_f32).  This file is built with -fno-honor-nans: otherwise hipcc canonicalises every MFMA
    // result before fmaxf (one extra v_max per score).  No inline asm here: an asm reader of MFMA results would need
    // its own MFMA->VALU wait states (cdna guide 5.7).
    auto max3 = [](float a, float b, float c) { return fmaxf(fmaxf(a, b), c); };
    // LOG2: s holds (score - reference) in exponent units; otherwise raw scores against m_run
    bool exact = true;
    float psum = 0.f;
    bf16x8 pf[4];
    if (LAZY && t > 0) {
      const float mc0 = m_run * c2;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const float e = LOG2 ? __builtin_amdgcn_exp2f(s[i >> 4][i & 15]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[i >> 4][i & 15], c2, -mc0));
        psum += e;
        pf[i >> 3][i & 7] = (bf16)e;
      }
      // psum >= 0, so its bit pattern orders like its value and +inf / NaN patterns sit above every finite limit: an integer
      // compare cannot be folded away by -fno-honor-nans (this file's build flag) the way !(psum <= limit) could
      exact = __any(__float_as_uint(psum) > __float_as_uint(LAZY_PSUM_LIMIT));   // too large for P's type / overflow / NaN
      if (exact) {
        asm volatile("; lazy-maximum overflow guard fired: scores again, exact path");
        compute_scores();
      }
    }
    if (exact) {
      if (LAZY) asm volatile("; exact softmax path (first tile / overflow guard)");
      float mx[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) mx[i] = max3(s[(3 * i) >> 4][(3 * i) & 15], s[(3 * i + 1) >> 4][(3 * i + 1) & 15],
                                                s[(3 * i + 2) >> 4][(3 * i + 2) & 15]);
      float mloc = max3(max3(mx[0], mx[1], mx[2]), max3(mx[3], mx[4], mx[5]), max3(mx[6], mx[7], mx[8]));
      mloc = max3(mloc, mx[9], fmaxf(s[1][14], s[1][15]));
      mloc = xhalf_max(mloc);
      float shift = 0.f;   // LOG2: what still has to come off the scores of this tile (they are relative to the OLD reference)
      if (LOG2) {
        // mloc is relative to the reference m_run (0 on the first tile, where m_run is still -inf and o = l = 0)
        const bool first = t == 0;
        if (first || !__all(mloc <= RESCALE_THR)) {
          asm volatile("; online-softmax rescale (rare)");
          const float mfin = (mloc == -INFINITY) ? 0.f : mloc;
          const float delta = first ? mfin : fmaxf(mfin, 0.f);
          const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);   // first tile: o = l = 0, nothing to rescale
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
        asm volatile("; online-softmax rescale (rare; asm volatile keeps it from being speculated)");
        const float m_new = fmaxf(m_run, mloc);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * c2);  // m_run = -inf -> 0
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

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      // B-operand element i of pf[kk] is key kk*16 + (i&3) + 8*(i>>2) + 4*hi -> the A operand must match
      const unsigned char* vrow = vs_ + (kk * 16 + 4 * hi + tr_row) * L::VRS + tr_col * 2;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const bf16x4 lo = ds_read_tr16((lds_bf16x4*)(vrow + db * 64));
        const bf16x4 hi4 = ds_read_tr16((lds_bf16x4*)(vrow + 8 * L::VRS + db * 64));
        const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        o[db] = mfma_32x32x16(vf, pf[kk], o[db]);
      }
    }

    if (NBUF == 2) {
      if (t + 1 < ntiles && stage_kv) store_kv(buf ^ 1);
      __syncthreads();
    } else {
      __syncthreads();
      if (t + 1 < ntiles) {
        load_kv((t + 1) * KVBLK);
        store_kv(0);
      }
      __syncthreads();
    }
  };

  const int nfull = p.Skv / KVBLK;
  // Query-tile loop. Long KV: one tile per block (QT == 1, the loop vanishes). Short KV (cross-attention, Skv <= 2 tiles = the two LDS
  // buffers): a block walks QT query tiles against K/V staged ONCE -- these launches are launch/latency bound
  // (SDXL 8x20x1024x77: 21.6 us for 45 MB), and the staging chain global -> registers -> LDS -> barrier is most of a block's life.
#pragma unroll 1
  for (int qi = 0; qi < QT; ++qi) {
    q_row = (qb * QT + qi) * QBLK + wave * QROWS + lq;
    q_ok = q_row < p.Sq;
    stage_kv = qi == 0;
    if (qi > 0 && (qb * QT + qi) * QBLK >= p.Sq) break;   // block-uniform: no query rows left
    if (qi > 0) load_q(qb * QT + qi, qf);   // (prefetching these under the previous tile cost 27 registers and was slower)
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    m_run = -INFINITY;
    l_run = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sinit[r] = 0.f;

    for (int t = 0; t < nfull; ++t) tile_body(t, std::false_type{});
    if (nfull < ntiles) tile_body(nfull, std::true_type{});

    // ---- finalize: O[q][d] = O^T[d][q] / l ; lane holds d = db*32 + (r&3) + 8*(r>>2) + 4*hi ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv_l = (l_tot > 0.f) ? 1.0f / l_tot : 0.f;
    if constexpr (WIDE) {
      bf16* orow = Op + (size_t)(q_ok ? q_row : 0) * p.o_ts + hi * 8;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
          const unsigned a0 = pack_bf16(o[db][4 * c + 0] * inv_l, o[db][4 * c + 1] * inv_l), a1 = pack_bf16(o[db][4 * c + 2] * inv_l, o[db][4 * c + 3] * inv_l);
          const unsigned b0 = pack_bf16(o[db][4 * c + 4] * inv_l, o[db][4 * c + 5] * inv_l), b1 = pack_bf16(o[db][4 * c + 6] * inv_l, o[db][4 * c + 7] * inv_l);
          const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          const u32x4 pk = {r0[0], r1[0], r0[1], r1[1]};
          if (q_ok && db * 32 + c * 8 + hi * 8 < p.D) *reinterpret_cast<u32x4*>(orow + db * 32 + c * 8) = pk;
        }
    } else if (q_ok) {
      bf16* orow = Op + (size_t)q_row * p.o_ts;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int d = db * 32 + 8 * c + 4 * hi;
          if (d < p.D) {
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
  }
  if constexpr ((FL & 8) != 0) {
    if (tid == 0 && (blockIdx.x == 8 || blockIdx.x == gridDim.x / 2 + 8)) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<bf16*>(p.Q) + (size_t)(p.B - 1) * p.q_bs + (size_t)(p.Sq - 1) * p.q_ts + p.H * 64 - 16) + (blockIdx.x == 8 ? 0 : 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      dbg[0] = __builtin_readcyclecounter() - clk0;
      dbg[1] = wall_clock64() - wall0;
    }
  }
}

// ---- software-pipelined form of the d = 64 self-attention loop (round 5) --------------------------------------------------------
// attention_kernel above runs a wave's tile as three dependent stretches -- 8 QK^T MFMAs, ~80 VALU instructions of softmax, 8 P.V
// MFMAs -- and leaves their overlap to whatever other waves of the SIMD happen to be doing (measured: the matrix pipe 37 % busy, the
// two pipes' busy times ADD, profiles/r02_g_attention_pmc.txt). Here the three streams of ONE wave are independent inside every
// stretch, so its MFMAs issue between its own exponentials. A K/V tile of 64 keys is handled as two halves of 32 (h0, h1); with
// S0 / S1 the score accumulators and P0 / P1 the packed probabilities of the two halves, iteration j runs
//     phase 1:  S0 = K(j+1,h0) Q^T        O^T += V(j,h0)^T P0^T        P1 = exp2(S1)            [S1 = scores of (j, h1)]
//     phase 2:  S1 = K(j+1,h1) Q^T        O^T += V(j,h1)^T P1^T        P0 = exp2(S0)            [S0 = scores of (j+1, h0)]
// -- in each phase the MFMAs write one S and read one P while the VALU reads the OTHER S and writes the OTHER P: no double buffers
// beyond the registers attention_kernel already holds (S 32, P 16). K therefore runs one tile ahead of V in the LDS ring.
// Softmax: the lazy reference of attention_kernel (FL bit 0) -- exact maximum on the first tile only, afterwards scores are
// exponentiated against the running reference and a phase whose partial row sum leaves [0, LAZY_PSUM_LIMIT] is redone the exact
// way (scores recomputed from the K tile still in LDS, reference raised, O / l / the other S rescaled). LOG2 as above.
// Same operand layouts, LDS images, MFMA shapes and per-tile summation order of O as attention_kernel; P of a tile's first half is
// exponentiated against the reference valid BEFORE that tile's second half was looked at (the non-pipelined kernel decides per
// 64-key tile), so results agree to rounding, not bit for bit.
// ABL (timing experiments only, WRONG results; scripts/r05): bit 0 no exponentials, 1 fragments stay in registers (no LDS reads in
// the loop), 2 no global loads / ring stores, 3 no barrier, 4 no MFMAs
template <bool LOG2, int SCHED, int ABL = 0>
__global__ __launch_bounds__(ATT_THREADS, (SCHED != 0 && ABL == 0) ? 3 : 2) void attention_il_kernel(const AttnArgs p) {
  using L = AttLds<64>;
  constexpr int KS = 4, DB = 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (L::KBYTES + L::VBYTES)];
  unsigned char* const kb_ = smem;                     // K ring: [2][KBYTES]
  unsigned char* const vb_ = smem + 2 * L::KBYTES;     // V ring: [2][VBYTES]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int lq = lane & 31;
  [[maybe_unused]] unsigned long long clk0 = 0, wall0 = 0;
  if constexpr ((ABL & 64) != 0) {   // shader-clock measurement (s_memtime ticks per 100-MHz wall tick over this block's life)
    clk0 = __builtin_readcyclecounter();
    wall0 = wall_clock64();
  }

  const int nqb = (p.Sq + QBLK - 1) / QBLK;
  const int lid = xcd_remap(blockIdx.x, nqb * p.B * p.H);
  const int qb = lid % nqb;
  const int bh = lid / nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const bf16* Qp = p.Q + (size_t)b * p.q_bs + (size_t)h * 64;
  const bf16* Kp = p.K + (size_t)b * p.k_bs + (size_t)h * 64;
  const bf16* Vp = p.V + (size_t)b * p.v_bs + (size_t)h * 64;
  bf16* Op = p.O + (size_t)b * p.o_bs + (size_t)h * 64;

  // ---- K / V tile loaders (buffer loads, rows past Skv read as zero; see attention_kernel) ----
  u32x4 rk[2], rv[2];
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(Kp), 0, (unsigned)(((size_t)(p.Skv - 1) * p.k_ts + 64) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(Vp), 0, (unsigned)(((size_t)(p.Skv - 1) * p.v_ts + 64) * 2), 0x00020000);
  const int ld_row = tid >> 3, ld_ch = tid & 7;   // chunk i of this thread: row ld_row + 32 i, 16-byte chunk ld_ch
  const unsigned k_off0 = (unsigned)((ld_row * p.k_ts + ld_ch * 8) * 2), v_off0 = (unsigned)((ld_row * p.v_ts + ld_ch * 8) * 2);
  const unsigned k_step = (unsigned)(32 * p.k_ts * 2), v_step = (unsigned)(32 * p.v_ts * 2);
  auto load_k = [&](const int t) {
    const unsigned o = k_off0 + (unsigned)(t * KVBLK * p.k_ts * 2);
    rk[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, o, 0, 0));
    rk[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, o + k_step, 0, 0));
  };
  auto load_v = [&](const int t) {
    const unsigned o = v_off0 + (unsigned)(t * KVBLK * p.v_ts * 2);
    rv[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, o, 0, 0));
    rv[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, o + v_step, 0, 0));
  };
  auto store_k = [&](const int buf) {
    unsigned char* d = kb_ + buf * L::KBYTES + ld_row * L::KRS + ld_ch * 16;
    *reinterpret_cast<u32x4*>(d) = rk[0];
    *reinterpret_cast<u32x4*>(d + 32 * L::KRS) = rk[1];
  };
  auto store_v = [&](const int buf) {
    unsigned char* d = vb_ + buf * L::VBYTES + ld_row * L::VRS + ld_ch * 16;
    *reinterpret_cast<u32x4*>(d) = rv[0];
    *reinterpret_cast<u32x4*>(d + 32 * L::VRS) = rv[1];
  };

  const int ntiles = (p.Skv + KVBLK - 1) / KVBLK;
  const int q_row = qb * QBLK + wave * QROWS + lq;
  const bool q_ok = q_row < p.Sq;
  bf16x8 qf[KS];
  {
    const bf16* qr = Qp + (size_t)(q_ok ? q_row : 0) * p.q_ts + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qr + ks * 16);
  }
  load_k(0);
  load_v(0);
  store_k(0);
  store_v(0);
  if (ntiles > 1) {
    load_k(1);
    store_k(1);
  }
  __syncthreads();

  const int i16 = lane & 15;
  const int tr_off = (4 * hi + (i16 >> 2)) * L::VRS + (((lane >> 4) & 1) * 16 + (i16 & 3) * 4) * 2;   // transpose-read lane geometry
  const int kf_off = lq * L::KRS + hi * 16;

  f32x16 o[DB], S0, S1, sinit;
  bf16x8 P0[2], P1[2];
  float l_run = 0.f;
  float mref2 = 0.f;   // !LOG2: the reference in exponent units (LOG2 keeps it as -sinit)
  const float c2 = (LOG2 || p.log2) ? 1.0f : p.scale * 1.4426950408889634f;   // (the reference-less form also serves folded queries)
#pragma unroll
  for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = sinit[r] = 0.f;

  // S^T of one half: keys t*64 + hh*32 .. +31 against the wave's 32 queries (C operand: -reference in the LOG2 form)
  bf16x8 kfix[KS], vfix[2][DB];   // ABL bit 1: the fragments every phase multiplies by
  if constexpr ((ABL & 2) != 0) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kfix[ks] = *reinterpret_cast<const bf16x8*>(kb_ + kf_off + ks * 32);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const bf16x4 lo = ds_read_tr16((lds_bf16x4*)(vb_ + tr_off + k2 * 16 * L::VRS + db * 64));
        const bf16x4 hi4 = ds_read_tr16((lds_bf16x4*)(vb_ + tr_off + (k2 * 16 + 8) * L::VRS + db * 64));
        vfix[k2][db] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      }
  }
  auto qk_half = [&](f32x16& S, const int t, const int hh) {
    const unsigned char* kr = kb_ + (t & 1) * L::KBYTES + hh * 32 * L::KRS + kf_off;
    if constexpr ((ABL & 16) == 0) S = sinit;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 kf;
      if constexpr ((ABL & 2) != 0) kf = kfix[ks];
      else kf = *reinterpret_cast<const bf16x8*>(kr + ks * 32);
      if constexpr ((ABL & 16) != 0) asm volatile("" ::"v"(kf));
      else S = mfma_32x32x16(kf, qf[ks], S);
    }
  };
  // O^T += V(t, hh)^T P^T
  auto pv_half = [&](const bf16x8 (&P)[2], const int t, const int hh) {
    const unsigned char* vr = vb_ + (t & 1) * L::VBYTES + hh * 32 * L::VRS + tr_off;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        bf16x8 vf;
        if constexpr ((ABL & 2) != 0) {
          vf = vfix[k2][db];
        } else {
          const bf16x4 lo = ds_read_tr16((lds_bf16x4*)(vr + k2 * 16 * L::VRS + db * 64));
          const bf16x4 hi4 = ds_read_tr16((lds_bf16x4*)(vr + (k2 * 16 + 8) * L::VRS + db * 64));
          vf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        }
        if constexpr ((ABL & 16) != 0) asm volatile("" ::"v"(vf), "v"(P[k2]));
        else o[db] = mfma_32x32x16(vf, P[k2], o[db]);
      }
  };
  // keys past Skv (ragged last tile): s[r] is key kv0 + (r&3) + 8*(r>>2) + 4*hi
  auto mask_half = [&](f32x16& S, const int kv0) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (kv0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.Skv) S[r] = -INFINITY;
  };
  // P = exp2(S - reference), partial row sum
  auto sm_half = [&](const f32x16& S, bf16x8 (&P)[2]) {
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float x = LOG2 ? S[r] : __builtin_fmaf(S[r], c2, -mref2);
      const float e = (ABL & 1) ? x : __builtin_amdgcn_exp2f(x);
      psum += e;
      P[r >> 3][r & 7] = (bf16)e;
    }
    if constexpr ((ABL & 16) != 0) asm volatile("" ::"v"(P[0]), "v"(P[1]));
    return psum;
  };
  auto max3 = [](float a, float b, float c) { return fmaxf(fmaxf(a, b), c); };
  auto rowmax16 = [&](const f32x16& S) {
    float m = max3(S[0], S[1], S[2]);
#pragma unroll
    for (int i = 1; i < 5; ++i) m = max3(m, S[3 * i], max3(S[3 * i + 1], S[3 * i + 2], m));
    return fmaxf(m, S[15]);
  };
  auto xhalf_max = [](float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  };
  // The exact, non-pipelined way for one half: what a wave does from the moment its lazy exponentials left the safe range (rare;
  // see the slow path at the bottom). K fragments straight from global memory -- the ring slot of K(t) may already be refilled by
  // waves that are ahead -- V from its ring slot, which is live during iteration t for every wave.
  auto slow_half = [&](const int t, const int hh) {
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
    const unsigned ko = (unsigned)((((t * KVBLK + hh * 32 + lq) * p.k_ts) + hi * 8) * 2);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 kf = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, ko + ks * 32, 0, 0));
      S = mfma_32x32x16(__builtin_bit_cast(bf16x8, kf), qf[ks], S);
    }
    mask_half(S, t * KVBLK + hh * 32);
    const float mloc = xhalf_max(rowmax16(S));
    const float mx = LOG2 ? mloc : mloc * c2;
    if (!__all(mx <= mref2)) {
      const float mnew = fmaxf(mref2, mx);
      const float alpha = __builtin_amdgcn_exp2f(mref2 - mnew);
      l_run *= alpha;
      mref2 = mnew;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] *= alpha;
        o[1][r] *= alpha;
      }
    }
    bf16x8 P[2];
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], c2, -mref2));
      psum += e;
      P[r >> 3][r & 7] = (bf16)e;
    }
    l_run += psum;
    pv_half(P, t, hh);
  };
  // (the guard stays in the ablation builds -- without it hipcc merges the phases and the timing says nothing about this loop)
  auto guard = [](const float psum) { return (ABL & 17) == 0 && __any(__float_as_uint(psum) > __float_as_uint(LAZY_PSUM_LIMIT)); };

  // ---- prologue: scores of tile 0, exact maximum, P of its first half ----
  qk_half(S0, 0, 0);
  qk_half(S1, 0, 1);
  if (ntiles == 1) {
    mask_half(S0, 0);
    mask_half(S1, 32);
  }
  {
    const float mloc = xhalf_max(fmaxf(rowmax16(S0), rowmax16(S1)));
    const float mfin = (mloc == -INFINITY) ? 0.f : mloc;
    if (LOG2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        S0[r] -= mfin;
        S1[r] -= mfin;
        sinit[r] = -mfin;
      }
    } else {
      mref2 = mfin * c2;
    }
    l_run = sm_half(S0, P0);
  }

  // The pieces of iteration j (NEXT: there is a tile j+1).
  // (loads and ring stores are unconditional: a tile past the end is outside the descriptors' range and reads as zeros into a ring
  // slot nobody reads again. Under `if (there is such a tile)` the compiler cannot match the load's condition with the store's,
  // assumes the staging registers may still be in flight at the loop head, and waits vmcnt(0) -- i.e. for the loads it has just
  // issued -- before the first fragment read reuses one of them.)
  auto top = [&](const int j) {   // global loads of the tiles the NEXT iteration reads: K(j+2), V(j+1)
    if constexpr ((ABL & 4) == 0) {
      load_k(j + 2);
      load_v(j + 1);
    }
  };
  // SCHED == 1: the emitted order of a full phase is written out and fenced (sched_barrier): the first two K fragment reads, two
  // chunks of exponentials while they are in flight, then one MFMA per chunk (2 exp, 2 add, 1 convert) with the remaining
  // fragment reads two MFMAs ahead of their use. Left to itself (SCHED == 0) hipcc interleaves the three streams too, but opens
  // each phase with read -> wait -> MFMA.
  auto phase_pinned = [&](f32x16& Sw, const f32x16& Sr, const bf16x8 (&Pr)[2], bf16x8 (&Pw)[2], const int tn, const int hh, const int tc) {
    const unsigned char* kr = kb_ + (tn & 1) * L::KBYTES + hh * 32 * L::KRS + kf_off;
    const unsigned char* vr = vb_ + (tc & 1) * L::VBYTES + hh * 32 * L::VRS + tr_off;
    float psum = 0.f;
    auto chunk = [&](const int c) {
      const float e0 = __builtin_amdgcn_exp2f(LOG2 ? Sr[2 * c] : __builtin_fmaf(Sr[2 * c], c2, -mref2));
      const float e1 = __builtin_amdgcn_exp2f(LOG2 ? Sr[2 * c + 1] : __builtin_fmaf(Sr[2 * c + 1], c2, -mref2));
      psum += e0;
      psum += e1;
      Pw[c >> 2][(c & 3) * 2] = (bf16)e0;
      Pw[c >> 2][(c & 3) * 2 + 1] = (bf16)e1;
    };
    auto kread = [&](const int ks) { return *reinterpret_cast<const bf16x8*>(kr + ks * 32); };
    auto vread = [&](const int k2, const int db) {
      const bf16x4 lo = ds_read_tr16((lds_bf16x4*)(vr + k2 * 16 * L::VRS + db * 64));
      const bf16x4 hi4 = ds_read_tr16((lds_bf16x4*)(vr + (k2 * 16 + 8) * L::VRS + db * 64));
      return bf16x8{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    };
#define SD_FENCE() __builtin_amdgcn_sched_barrier(0)
    if constexpr (SCHED == 1) {
      // consecutive MFMAs never share an accumulator: S, O0, S, O1, ... A VALU instruction between two MFMAs on the SAME
      // accumulator costs ~43 cycles (MI355X_MICROARCH.md "one EXTRA issue slot ... a cliff"); with another accumulator's MFMA
      // in between the dependent one finds its C operand finished.
      const bf16x8 k0 = kread(0);
      const bf16x8 v00 = vread(0, 0);
      chunk(0);
      chunk(1);
      SD_FENCE();
      Sw = mfma_32x32x16(k0, qf[0], sinit);
      const bf16x8 k1 = kread(1);
      const bf16x8 v01 = vread(0, 1);
      chunk(2);
      SD_FENCE();
      o[0] = mfma_32x32x16(v00, Pr[0], o[0]);
      const bf16x8 k2 = kread(2);
      chunk(3);
      SD_FENCE();
      Sw = mfma_32x32x16(k1, qf[1], Sw);
      const bf16x8 v10 = vread(1, 0);
      chunk(4);
      SD_FENCE();
      o[1] = mfma_32x32x16(v01, Pr[0], o[1]);
      const bf16x8 k3 = kread(3);
      chunk(5);
      SD_FENCE();
      Sw = mfma_32x32x16(k2, qf[2], Sw);
      const bf16x8 v11 = vread(1, 1);
      chunk(6);
      SD_FENCE();
      o[0] = mfma_32x32x16(v10, Pr[1], o[0]);
      chunk(7);
      SD_FENCE();
      Sw = mfma_32x32x16(k3, qf[3], Sw);
      SD_FENCE();
      o[1] = mfma_32x32x16(v11, Pr[1], o[1]);
      SD_FENCE();
    } else {
      // SCHED == 2: the S chain back to back with nothing in between (the forwarding path of a same-accumulator chain), the VALU
      // work in the gaps of the four O MFMAs
      const bf16x8 k0 = kread(0), k1 = kread(1), k2 = kread(2), k3 = kread(3);
      chunk(0);
      chunk(1);
      SD_FENCE();
      Sw = mfma_32x32x16(k0, qf[0], sinit);
      Sw = mfma_32x32x16(k1, qf[1], Sw);
      Sw = mfma_32x32x16(k2, qf[2], Sw);
      Sw = mfma_32x32x16(k3, qf[3], Sw);
      SD_FENCE();
      const bf16x8 v00 = vread(0, 0), v01 = vread(0, 1);
      chunk(2);
      chunk(3);
      SD_FENCE();
      o[0] = mfma_32x32x16(v00, Pr[0], o[0]);
      const bf16x8 v10 = vread(1, 0), v11 = vread(1, 1);
      chunk(4);
      SD_FENCE();
      o[1] = mfma_32x32x16(v01, Pr[0], o[1]);
      chunk(5);
      SD_FENCE();
      o[0] = mfma_32x32x16(v10, Pr[1], o[0]);
      chunk(6);
      SD_FENCE();
      o[1] = mfma_32x32x16(v11, Pr[1], o[1]);
      chunk(7);
      SD_FENCE();
    }
#undef SD_FENCE
    return psum;
  };
  auto phase1 = [&](const int j, auto next_tag) {   // -> the partial row sum of the half it exponentiated
    if constexpr (decltype(next_tag)::value && SCHED != 0) {
      return phase_pinned(S0, S1, P0, P1, j + 1, 0, j);
    } else {
      if constexpr (decltype(next_tag)::value) qk_half(S0, j + 1, 0);
      pv_half(P0, j, 0);
      return sm_half(S1, P1);
    }
  };
  auto phase2 = [&](const int j, auto next_tag) {
    if constexpr (decltype(next_tag)::value && SCHED != 0) {
      return phase_pinned(S1, S0, P1, P0, j + 1, 1, j);
    } else {
      float ps = 0.f;
      if constexpr (decltype(next_tag)::value) qk_half(S1, j + 1, 1);
      pv_half(P1, j, 1);
      if constexpr (decltype(next_tag)::value) ps = sm_half(S0, P0);
      return ps;
    }
  };
  auto bottom = [&](const int j) {   // stage what `top` requested, publish
    if constexpr ((ABL & 4) == 0) {
      store_k(j & 1);
      store_v((j + 1) & 1);
    }
    if constexpr ((ABL & 8) == 0) __syncthreads();
  };
  using T = std::true_type;
  using F = std::false_type;
  // Steady state. Nothing but the two phases touches the loop-carried registers (O, S0, S1, the reference, P0, l): a fired guard
  // only LEAVES the loop -- a first form that repaired the state in place brought the repaired O / S / reference back as phis,
  // which the allocator resolved with 25 register-pair copies per iteration on the hot path. The ragged last tile's scores are
  // masked by the iteration that produces them, at a phase boundary (wave-uniform branch, taken once).
  bool slow = false;
  int g = 0;   // slow path: the first half (2 * tile + half) still to do
  int j = 0;
  for (; j + 1 < ntiles; ++j) {
    top(j);
    const float ps1 = phase1(j, T{});
    if (guard(ps1)) {
      slow = true;
      g = 2 * j + 1;
      break;
    }
    l_run += ps1;
    if (j + 2 == ntiles) mask_half(S0, (j + 1) * KVBLK);
    const float ps2 = phase2(j, T{});
    if (guard(ps2)) {
      slow = true;
      g = 2 * j + 2;
      break;
    }
    l_run += ps2;
    if (j + 2 == ntiles) mask_half(S1, (j + 1) * KVBLK + 32);
    bottom(j);
  }
  if (!slow) {   // the last tile: no scores left to produce
    const float ps1 = phase1(j, F{});
    if (guard(ps1)) {
      slow = true;
      g = 2 * j + 1;
    } else {
      l_run += ps1;
      phase2(j, F{});
    }
  }
  if (slow) {
    // This wave finishes its queries half by half with the exact online softmax. It left the loop inside iteration j (whose global
    // loads it has issued, whose staging + barrier it still owes) and keeps serving the block's schedule -- loads, ring stores and one
    // barrier per iteration -- so the other waves never notice.
    asm volatile("; pipelined attention: lazy-maximum guard fired, exact path for the rest of this wave's keys");
    if (LOG2) mref2 = -sinit[0];
    int t = g >> 1, hh = g & 1;
    for (int i = j; i < ntiles; ++i) {
      if (i != j) top(i);
      if (t == i) {
        for (; hh < 2; ++hh) slow_half(i, hh);
        ++t;
        hh = 0;
      }
      if (i + 1 < ntiles) bottom(i);
    }
  }

  // ---- finalize: O[q][d] = O^T[d][q] / l, 16-byte stores (the launcher checked alignment; see attention_kernel WIDE) ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv_l = (l_tot > 0.f) ? 1.0f / l_tot : 0.f;
  bf16* orow = Op + (size_t)(q_ok ? q_row : 0) * p.o_ts + hi * 8;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int c = 0; c < 4; c += 2) {
      const unsigned a0 = pack_bf16(o[db][4 * c + 0] * inv_l, o[db][4 * c + 1] * inv_l), a1 = pack_bf16(o[db][4 * c + 2] * inv_l, o[db][4 * c + 3] * inv_l);
      const unsigned b0 = pack_bf16(o[db][4 * c + 4] * inv_l, o[db][4 * c + 5] * inv_l), b1 = pack_bf16(o[db][4 * c + 6] * inv_l, o[db][4 * c + 7] * inv_l);
      const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      const u32x4 pk = {r0[0], r1[0], r0[1], r1[1]};
      if (q_ok) *reinterpret_cast<u32x4*>(orow + db * 32 + c * 8) = pk;
    }
  if constexpr ((ABL & 64) != 0) {
    if (tid == 0 && (blockIdx.x == 8 || blockIdx.x == gridDim.x / 2 + 8)) {   // (blocks 8.. land on XCD 0: their query rows are not block 0's)
      // (into the tail of the last query row: an input nobody writes; later launches read 16 odd values there -- a debug build)
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<bf16*>(p.Q) + (size_t)(p.B - 1) * p.q_bs + (size_t)(p.Sq - 1) * p.q_ts + p.H * 64 - 16) + (blockIdx.x == 8 ? 0 : 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      dbg[0] = __builtin_readcyclecounter() - clk0;
      dbg[1] = wall_clock64() - wall0;
    }
  }
}

// ---- two query tiles per wave, one wave per SIMD (round 5) ----------------------------------------------------------------------
// attention_il_kernel moved nothing (profiles/r05_s1 .. s3): with one 32-query tile per wave every MFMA is fed by its own LDS fragment
// read, and every 16 MFMAs pay a K/V tile's staging (4 global loads, 4 ring stores, a barrier). Here a wave owns TWO query tiles A and B
// (64 queries, a block 256) and the whole 512-entry register file of its SIMD (one block of four waves per CU): every K and V
// fragment read from LDS feeds two MFMAs, a staged tile feeds 32 per wave. The two tiles run half an iteration apart,
//     phase 1:  O_A += V(j)^T P_A^T    S_A = K(j+1) Q_A^T        P_B = exp2(S_B)     [S_B = scores of (B, j)]
//     phase 2:  O_B += V(j)^T P_B^T    S_B = K(j+1) Q_B^T        P_A = exp2(S_A)     [S_A = scores of (A, j+1)]
// so that inside a phase the 16 MFMAs (tile X) and the ~80 VALU instructions (tile Y) are independent streams of one wave -- there is
// no second wave on the SIMD to fill its stalls -- and nothing is double-buffered; the fragments of K(j+1) and V(j) are read once, in
// phase 1, and stay in registers for phase 2. Softmax, guard and exact path as in attention_il_kernel (whole 64-key tiles here).
template <bool LOG2, int SCHED = 1>
__global__ __launch_bounds__(ATT_THREADS, 1) void attention_q2_kernel(const AttnArgs p) {
  using L = AttLds<64>;
  constexpr int KS = 4, DB = 2, QB2 = 2 * QBLK;   // 256 queries per block
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (L::KBYTES + L::VBYTES)];
  unsigned char* const kb_ = smem;
  unsigned char* const vb_ = smem + 2 * L::KBYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int lq = lane & 31;

  const int nqb = (p.Sq + QB2 - 1) / QB2;
  const int lid = xcd_remap(blockIdx.x, nqb * p.B * p.H);
  const int qb = lid % nqb;
  const int bh = lid / nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const bf16* Qp = p.Q + (size_t)b * p.q_bs + (size_t)h * 64;
  const bf16* Kp = p.K + (size_t)b * p.k_bs + (size_t)h * 64;
  const bf16* Vp = p.V + (size_t)b * p.v_bs + (size_t)h * 64;
  bf16* Op = p.O + (size_t)b * p.o_bs + (size_t)h * 64;

  u32x4 rk[2], rv[2];
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(Kp), 0, (unsigned)(((size_t)(p.Skv - 1) * p.k_ts + 64) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(Vp), 0, (unsigned)(((size_t)(p.Skv - 1) * p.v_ts + 64) * 2), 0x00020000);
  const int ld_row = tid >> 3, ld_ch = tid & 7;
  const unsigned k_off0 = (unsigned)((ld_row * p.k_ts + ld_ch * 8) * 2), v_off0 = (unsigned)((ld_row * p.v_ts + ld_ch * 8) * 2);
  const unsigned k_step = (unsigned)(32 * p.k_ts * 2), v_step = (unsigned)(32 * p.v_ts * 2);
  auto load_k = [&](const int t) {
    const unsigned o = k_off0 + (unsigned)(t * KVBLK * p.k_ts * 2);
    rk[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, o, 0, 0));
    rk[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, o + k_step, 0, 0));
  };
  auto load_v = [&](const int t) {
    const unsigned o = v_off0 + (unsigned)(t * KVBLK * p.v_ts * 2);
    rv[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, o, 0, 0));
    rv[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, o + v_step, 0, 0));
  };
  auto store_k = [&](const int buf) {
    unsigned char* d = kb_ + buf * L::KBYTES + ld_row * L::KRS + ld_ch * 16;
    *reinterpret_cast<u32x4*>(d) = rk[0];
    *reinterpret_cast<u32x4*>(d + 32 * L::KRS) = rk[1];
  };
  auto store_v = [&](const int buf) {
    unsigned char* d = vb_ + buf * L::VBYTES + ld_row * L::VRS + ld_ch * 16;
    *reinterpret_cast<u32x4*>(d) = rv[0];
    *reinterpret_cast<u32x4*>(d + 32 * L::VRS) = rv[1];
  };

  const int ntiles = (p.Skv + KVBLK - 1) / KVBLK;
  // query tile x (0 = A, 1 = B): rows qb*256 + wave*64 + x*32 + lq
  int q_row[2];
  bool q_ok[2];
  bf16x8 qf[2][KS];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    q_row[x] = qb * QB2 + wave * 64 + x * 32 + lq;
    q_ok[x] = q_row[x] < p.Sq;
    const bf16* qr = Qp + (size_t)(q_ok[x] ? q_row[x] : 0) * p.q_ts + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[x][ks] = *reinterpret_cast<const bf16x8*>(qr + ks * 16);
  }
  load_k(0);
  load_v(0);
  store_k(0);
  store_v(0);
  if (ntiles > 1) {
    load_k(1);
    store_k(1);
  }
  __syncthreads();

  const int i16 = lane & 15;
  const int tr_off = (4 * hi + (i16 >> 2)) * L::VRS + (((lane >> 4) & 1) * 16 + (i16 & 3) * 4) * 2;
  const int kf_off = lq * L::KRS + hi * 16;

  f32x16 o[2][DB], S[2][2], sinit[2];   // [query tile][...]
  bf16x8 P[2][4];
  float l_run[2] = {0.f, 0.f}, mref2[2] = {0.f, 0.f};
  const float c2 = (LOG2 || p.log2) ? 1.0f : p.scale * 1.4426950408889634f;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[x][0][r] = o[x][1][r] = sinit[x][r] = 0.f;

  bf16x8 kf[2][KS], vf[4][DB];   // fragments of K(j+1) and V(j): read in phase 1, used by both phases
  auto read_k = [&](const int t) {
    const unsigned char* kr = kb_ + (t & 1) * L::KBYTES + kf_off;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kf[hh][ks] = *reinterpret_cast<const bf16x8*>(kr + hh * 32 * L::KRS + ks * 32);
  };
  auto read_v = [&](const int t) {
    const unsigned char* vr = vb_ + (t & 1) * L::VBYTES + tr_off;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const bf16x4 lo = ds_read_tr16((lds_bf16x4*)(vr + kk * 16 * L::VRS + db * 64));
        const bf16x4 hi4 = ds_read_tr16((lds_bf16x4*)(vr + (kk * 16 + 8) * L::VRS + db * 64));
        vf[kk][db] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
      }
  };
  auto qk_tile = [&](const int x) {   // S[x] = kf . Q_x^T (+ the -reference C operand)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      S[x][hh] = sinit[x];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) S[x][hh] = mfma_32x32x16(kf[hh][ks], qf[x][ks], S[x][hh]);
    }
  };
  auto pv_tile = [&](const int x) {   // O_x^T += vf^T P_x^T
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int db = 0; db < DB; ++db) o[x][db] = mfma_32x32x16(vf[kk][db], P[x][kk], o[x][db]);
  };
  auto mask_tile = [&](const int x, const int kv0) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kv0 + hh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.Skv) S[x][hh][r] = -INFINITY;
  };
  auto sm_tile = [&](const int x) {   // P_x = exp2(S_x - reference); -> partial row sum
    float psum = 0.f;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(LOG2 ? S[x][hh][r] : __builtin_fmaf(S[x][hh][r], c2, -mref2[x]));
        psum += e;
        P[x][hh * 2 + (r >> 3)][r & 7] = (bf16)e;
      }
    return psum;
  };
  // One phase in a pinned order (SCHED == 1; with one wave per SIMD nobody else fills this wave's stalls, and left to itself hipcc
  // issues ~45 VALU instructions before the phase's first MFMA and its last ten MFMAs back to back): the MFMAs of query tile X --
  // O0, S0, O1, S1 in turn, so that consecutive ones never share an accumulator -- one per chunk (2 exp, 2 add, 1 convert) of
  // tile Y's exponentials; READS: phase 1 issues all 24 fragment reads first and covers their latency with three chunks.
  auto phase_pinned = [&](auto x_tag, auto reads_tag, const int tk, const int tv) {
    constexpr int X = decltype(x_tag)::value, Y = X ^ 1;
    float psum = 0.f;
    auto chunk = [&](const int c) {   // scores 2c, 2c+1 of tile Y (c < 8: first half)
      const int hh = c >> 3, r = (c & 7) * 2;
      const float e0 = __builtin_amdgcn_exp2f(LOG2 ? S[Y][hh][r] : __builtin_fmaf(S[Y][hh][r], c2, -mref2[Y]));
      const float e1 = __builtin_amdgcn_exp2f(LOG2 ? S[Y][hh][r + 1] : __builtin_fmaf(S[Y][hh][r + 1], c2, -mref2[Y]));
      psum += e0;
      psum += e1;
      P[Y][hh * 2 + (r >> 3)][r & 7] = (bf16)e0;
      P[Y][hh * 2 + (r >> 3)][(r & 7) + 1] = (bf16)e1;
    };
    f32x16 sw0 = sinit[X], sw1 = sinit[X];
#define SD_FENCE() __builtin_amdgcn_sched_barrier(0)
    if constexpr (decltype(reads_tag)::value) {
      read_k(tk);
      read_v(tv);
      chunk(0);
      chunk(1);
      chunk(2);
      SD_FENCE();
    } else {
      chunk(0);
      chunk(1);
      chunk(2);
      SD_FENCE();
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {   // group g: key block g of P.V, k-step g of K.Q^T
      o[X][0] = mfma_32x32x16(vf[g][0], P[X][g], o[X][0]);
      chunk(3 + 3 * g);
      SD_FENCE();
      sw0 = mfma_32x32x16(kf[0][g], qf[X][g], sw0);
      chunk(4 + 3 * g);
      SD_FENCE();
      o[X][1] = mfma_32x32x16(vf[g][1], P[X][g], o[X][1]);
      chunk(5 + 3 * g);
      SD_FENCE();
      sw1 = mfma_32x32x16(kf[1][g], qf[X][g], sw1);
      if (g == 3) chunk(15);
      SD_FENCE();
    }
#undef SD_FENCE
    S[X][0] = sw0;
    S[X][1] = sw1;
    return psum;
  };
  auto max3 = [](float a, float b, float c) { return fmaxf(fmaxf(a, b), c); };
  auto rowmax16 = [&](const f32x16& s) {
    float m = max3(s[0], s[1], s[2]);
#pragma unroll
    for (int i = 1; i < 5; ++i) m = max3(m, s[3 * i], max3(s[3 * i + 1], s[3 * i + 2], m));
    return fmaxf(m, s[15]);
  };
  auto xhalf_max = [](float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  };
  auto guard = [](const float psum) { return __any(__float_as_uint(psum) > __float_as_uint(LAZY_PSUM_LIMIT)); };
  // exact, non-pipelined tile for query tile x (the slow path below): K fragments from global memory, V from its ring slot
  auto slow_tile = [&](const int x, const int t) {
    f32x16 s2[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s2[hh][r] = 0.f;
      const unsigned ko = (unsigned)((((t * KVBLK + hh * 32 + lq) * p.k_ts) + hi * 8) * 2);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 kg = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, ko + ks * 32, 0, 0));
        s2[hh] = mfma_32x32x16(__builtin_bit_cast(bf16x8, kg), qf[x][ks], s2[hh]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (t * KVBLK + hh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.Skv) s2[hh][r] = -INFINITY;
    }
    const float mloc = xhalf_max(fmaxf(rowmax16(s2[0]), rowmax16(s2[1])));
    const float mx = LOG2 ? mloc : mloc * c2;
    if (!__all(mx <= mref2[x])) {
      const float mnew = fmaxf(mref2[x], mx);
      const float alpha = __builtin_amdgcn_exp2f(mref2[x] - mnew);
      l_run[x] *= alpha;
      mref2[x] = mnew;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[x][0][r] *= alpha;
        o[x][1][r] *= alpha;
      }
    }
    bf16x8 pp[4];
    float psum = 0.f;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s2[hh][r], c2, -mref2[x]));
        psum += e;
        pp[hh * 2 + (r >> 3)][r & 7] = (bf16)e;
      }
    l_run[x] += psum;
    const unsigned char* vr = vb_ + (t & 1) * L::VBYTES + tr_off;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const bf16x4 lo = ds_read_tr16((lds_bf16x4*)(vr + kk * 16 * L::VRS + db * 64));
        const bf16x4 hi4 = ds_read_tr16((lds_bf16x4*)(vr + (kk * 16 + 8) * L::VRS + db * 64));
        o[x][db] = mfma_32x32x16(bf16x8{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]}, pp[kk], o[x][db]);
      }
  };

  // ---- prologue: scores of tile 0 for both query tiles, exact maxima, P_A of tile 0 ----
  read_k(0);
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    qk_tile(x);
    if (ntiles == 1) mask_tile(x, 0);
    const float mloc = xhalf_max(fmaxf(rowmax16(S[x][0]), rowmax16(S[x][1])));
    const float mfin = (mloc == -INFINITY) ? 0.f : mloc;
    if (LOG2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        S[x][0][r] -= mfin;
        S[x][1][r] -= mfin;
        sinit[x][r] = -mfin;
      }
    } else {
      mref2[x] = mfin * c2;
    }
  }
  l_run[0] = sm_tile(0);

  auto top = [&](const int j) {
    load_k(j + 2);
    load_v(j + 1);
  };
  auto bottom = [&](const int j) {
    store_k(j & 1);
    store_v((j + 1) & 1);
    __syncthreads();
  };
  bool slow = false;
  int tA = 0, tB = 0;   // slow path: the first tile each query tile still has to do
  int j = 0;
  for (; j + 1 < ntiles; ++j) {
    top(j);
    // phase 1
    float psB;
    if constexpr (SCHED == 1) {
      psB = phase_pinned(std::integral_constant<int, 0>{}, std::true_type{}, j + 1, j);
    } else {
      read_k(j + 1);
      read_v(j);
      pv_tile(0);
      qk_tile(0);
      psB = sm_tile(1);
    }
    if (guard(psB)) {
      slow = true;
      tA = j + 1;
      tB = j;
      break;
    }
    l_run[1] += psB;
    if (j + 2 == ntiles) mask_tile(0, (j + 1) * KVBLK);
    // phase 2
    float psA;
    if constexpr (SCHED == 1) {
      psA = phase_pinned(std::integral_constant<int, 1>{}, std::false_type{}, 0, 0);
    } else {
      pv_tile(1);
      qk_tile(1);
      psA = sm_tile(0);
    }
    if (guard(psA)) {
      slow = true;
      tA = j + 1;
      tB = j + 1;
      break;
    }
    l_run[0] += psA;
    if (j + 2 == ntiles) mask_tile(1, (j + 1) * KVBLK);
    bottom(j);
  }
  if (!slow) {   // last tile
    read_v(j);
    pv_tile(0);
    const float psB = sm_tile(1);
    if (guard(psB)) {
      slow = true;
      tA = j + 1;
      tB = j;
    } else {
      l_run[1] += psB;
      pv_tile(1);
    }
  }
  if (slow) {
    asm volatile("; two-tile attention: lazy-maximum guard fired, exact path for the rest of this wave's keys");
    if (LOG2) {
      mref2[0] = -sinit[0][0];
      mref2[1] = -sinit[1][0];
    }
    for (int i = j; i < ntiles; ++i) {
      if (i != j) top(i);
      if (tA <= i) slow_tile(0, i);
      if (tB <= i) slow_tile(1, i);
      if (i + 1 < ntiles) bottom(i);
    }
  }

  // ---- finalize both query tiles ----
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const float l_tot = l_run[x] + __shfl_xor(l_run[x], 32, 64);
    const float inv_l = (l_tot > 0.f) ? 1.0f / l_tot : 0.f;
    bf16* orow = Op + (size_t)(q_ok[x] ? q_row[x] : 0) * p.o_ts + hi * 8;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int c = 0; c < 4; c += 2) {
        const unsigned a0 = pack_bf16(o[x][db][4 * c + 0] * inv_l, o[x][db][4 * c + 1] * inv_l), a1 = pack_bf16(o[x][db][4 * c + 2] * inv_l, o[x][db][4 * c + 3] * inv_l);
        const unsigned b0 = pack_bf16(o[x][db][4 * c + 4] * inv_l, o[x][db][4 * c + 5] * inv_l), b1 = pack_bf16(o[x][db][4 * c + 6] * inv_l, o[x][db][4 * c + 7] * inv_l);
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        const u32x4 pk = {r0[0], r1[0], r0[1], r1[1]};
        if (q_ok[x]) *reinterpret_cast<u32x4*>(orow + db * 32 + c * 8) = pk;
      }
  }
}

// ---- short key sequences (cross-attention against 77 text tokens), head dim 64 ---------------------------------------------
// The SDXL step runs 70 such launches (8x20x1024x77, 8x10x4096x77): 45-90 MB of Q in / O out against 20 KB of K/V per (batch,
// head), i.e. a streaming problem -- which the flash kernel above turns into a chain of latencies: it walks 77 keys as two 64-key
// tiles of an online softmax (128 key columns of MFMA and exp work for 77), fetches each query tile only after the previous one
// is stored, and stores 8 bytes per lane. This kernel is cut for the shape instead:
//   * all keys at once: NSB <= 4 blocks of 32 keys (77 -> 3) resident in LDS for the block's life, ONE exact softmax pass per
//     query tile -- no running maximum, no rescale, no second tile;
//   * a block walks `qtpb` query tiles of its (batch, head); the next tile's Q fragments are requested before the current tile's
//     MFMAs, so the only exposed load latency is the first;
//   * O leaves as 16-byte stores: the two half-waves hold adjacent 8-byte pieces of a row, one v_permlane32_swap per dword pairs
//     them up (cdna guide T21) -- 4 store instructions per tile instead of 8.
// Same operand layouts, LDS images and MFMA order as attention_kernel (S^T = K Q^T, O^T = V^T P^T); results differ from it only by
// the softmax being exact in one pass (no deferred rescale).
template <int NSB, bool LOG2>
__global__ __launch_bounds__(ATT_THREADS, 2) void attention_short_kernel(const AttnArgs p, const int qtpb) {
  using L = AttLds<64>;
  constexpr int KS = 4, DB = 2, ROWS = NSB * 32;
  __shared__ __attribute__((aligned(16))) unsigned char smem[ROWS * (L::KRS + L::VRS)];
  unsigned char* ks_ = smem;
  unsigned char* vs_ = smem + ROWS * L::KRS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int lq = lane & 31;

  const int ntq = (p.Sq + QBLK - 1) / QBLK;          // query tiles per (batch, head)
  const int nqb = (ntq + qtpb - 1) / qtpb;            // blocks per (batch, head)
  const int lid = xcd_remap(blockIdx.x, nqb * p.B * p.H);
  const int qb = lid % nqb;
  const int bh = lid / nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const bf16* Qp = p.Q + (size_t)b * p.q_bs + (size_t)h * 64;
  const bf16* Kp = p.K + (size_t)b * p.k_bs + (size_t)h * 64;
  const bf16* Vp = p.V + (size_t)b * p.v_bs + (size_t)h * 64;
  bf16* Op = p.O + (size_t)b * p.o_bs + (size_t)h * 64;

  // Q fragments (MFMA B operand) of query tile qt: lane (q = lq, hi) holds d = ks*16 + hi*8 .. +8. Rows past Sq: row 0 (never stored)
  auto load_q = [&](const int qt, bf16x8 (&dst)[KS]) {
    const int row = qt * QBLK + wave * QROWS + lq;
    const bf16* qr = Qp + (size_t)(row < p.Sq ? row : 0) * p.q_ts + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) dst[ks] = *reinterpret_cast<const bf16x8*>(qr + ks * 16);
  };
  bf16x8 qf[KS], qn[KS];
  const int qt0 = qb * qtpb;
  load_q(qt0, qf);   // in flight together with K / V

  // ---- K and V of this (batch, head): rows past Skv are outside the descriptor's range and read as zero ----
  {
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(Kp), 0, (unsigned)(((size_t)(p.Skv - 1) * p.k_ts + 64) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(Vp), 0, (unsigned)(((size_t)(p.Skv - 1) * p.v_ts + 64) * 2), 0x00020000);
    u32x4 rk[NSB], rv[NSB];
#pragma unroll
    for (int i = 0; i < NSB; ++i) {   // 16-byte chunk cid: row cid / 8, chunk cid % 8
      const int cid = tid + ATT_THREADS * i;
      const int row = cid >> 3, ch = cid & 7;
      rk[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, (unsigned)((row * p.k_ts + ch * 8) * 2), 0, 0));
      rv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, (unsigned)((row * p.v_ts + ch * 8) * 2), 0, 0));
    }
#pragma unroll
    for (int i = 0; i < NSB; ++i) {
      const int cid = tid + ATT_THREADS * i;
      const int row = cid >> 3, ch = cid & 7;
      *reinterpret_cast<u32x4*>(ks_ + row * L::KRS + ch * 16) = rk[i];
      *reinterpret_cast<u32x4*>(vs_ + row * L::VRS + ch * 16) = rv[i];
    }
  }
  __syncthreads();

  const int i16 = lane & 15;
  const int dh = (lane >> 4) & 1;
  const int tr_row = (i16 >> 2);
  const int tr_col = dh * 16 + (i16 & 3) * 4;
  const float c2 = LOG2 ? 1.0f : p.scale * 1.4426950408889634f;
  auto max3 = [](float a, float b, float c) { return fmaxf(fmaxf(a, b), c); };

#pragma unroll 1
  for (int qi = 0; qi < qtpb; ++qi) {
    const int qt = qt0 + qi;
    if (qt >= ntq) break;                              // block-uniform
    const bool more = qi + 1 < qtpb && qt + 1 < ntq;
    if (more) load_q(qt + 1, qn);                      // lands under this tile's MFMAs and exponentials

    // ---- S^T = K Q^T: lane (lq, hi) gets key sb*32 + (r&3) + 8*(r>>2) + 4*hi of query lq in s[sb][r] ----
    f32x16 s[NSB];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[sb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks_ + (sb * 32 + lq) * L::KRS + (ks * 2 + hi) * 16);
        s[sb] = mfma_32x32x16(kf, qf[ks], s[sb]);
      }
    }
    // keys past Skv (they sit in the last 32-key block only: the launcher picks NSB = ceil(Skv / 32))
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kv = (NSB - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (kv >= p.Skv) s[NSB - 1][r] = -INFINITY;
    }
    // ---- exact softmax in one pass (key 0 always exists: the maximum is finite) ----
    float mloc = s[0][0];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
#pragma unroll
      for (int i = 0; i < 5; ++i) mloc = max3(mloc, s[sb][3 * i], max3(s[sb][3 * i + 1], s[sb][3 * i + 2], mloc));
      mloc = fmaxf(mloc, s[sb][15]);
    }
    {
      const unsigned u = __float_as_uint(mloc);
      const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
      mloc = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    const float mc = mloc * c2;
    float psum = 0.f;
    bf16x8 pf[2 * NSB];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(LOG2 ? s[sb][r] - mc : __builtin_fmaf(s[sb][r], c2, -mc));
        psum += e;
        pf[sb * 2 + (r >> 3)][r & 7] = (bf16)e;
      }
    }
    // ---- O^T = V^T P^T ----
    f32x16 o[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2 * NSB; ++kk) {
      const unsigned char* vrow = vs_ + (kk * 16 + 4 * hi + tr_row) * L::VRS + tr_col * 2;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const bf16x4 lo = ds_read_tr16((lds_bf16x4*)(vrow + db * 64));
        const bf16x4 hi4 = ds_read_tr16((lds_bf16x4*)(vrow + 8 * L::VRS + db * 64));
        const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        o[db] = mfma_32x32x16(vf, pf[kk], o[db]);
      }
    }
    // ---- O[q][d] = O^T[d][q] / l: lane holds d = db*32 + 8c + 4*hi .. +3; the half-waves' pieces paired into 16-byte stores ----
    const float l_tot = psum + __shfl_xor(psum, 32, 64);
    const float inv_l = 1.0f / l_tot;
    const int q_row = qt * QBLK + wave * QROWS + lq;
    bf16* orow = Op + (size_t)(q_row < p.Sq ? q_row : 0) * p.o_ts + hi * 8;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int c = 0; c < 4; c += 2) {
        unsigned a0 = pack_bf16(o[db][4 * c + 0] * inv_l, o[db][4 * c + 1] * inv_l), a1 = pack_bf16(o[db][4 * c + 2] * inv_l, o[db][4 * c + 3] * inv_l);
        unsigned b0 = pack_bf16(o[db][4 * c + 4] * inv_l, o[db][4 * c + 5] * inv_l), b1 = pack_bf16(o[db][4 * c + 6] * inv_l, o[db][4 * c + 7] * inv_l);
        // lower half keeps its piece c and takes the upper half's piece c (channels 8c .. 8c+7); the upper half takes the lower
        // half's piece c+1 next to its own (channels 8c+8 .. 8c+15)
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        const u32x4 pk = {r0[0], r1[0], r0[1], r1[1]};
        if (q_row < p.Sq) *reinterpret_cast<u32x4*>(orow + db * 32 + c * 8) = pk;
      }
    if (more) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
    }
  }
}

// lazy row maximum (on: profiles/r02_attention.txt)
static bool attn_lazy() { return ATTN_LAZY_DEFAULT != 0; }

template <int DP>
static int launch_dp(const AttnArgs& a0, hipStream_t stream) {
  const AttnArgs& a = a0;
  // lazy only where it costs no occupancy step: the d <= 64 kernel without mask (158 / 168 VGPRs: still 3 waves per SIMD)
  const bool lazy = DP == 64 && !a.bias && (attn_lazy() || a.log2);
  if (a.log2 && (DP != 64 || a.bias)) return SD_ERR_UNSUPPORTED;
  // short KV (both tiles stay resident in the two LDS buffers) and enough query tiles to keep every CU busy: two query
  // tiles per block
  // short key sequences at head dim 64: the single-pass streaming kernel (MI355X_SD_ATTN_NO_SHORT: A/B switch). 16-byte O stores
  // need 16-byte-aligned rows; masks and the accumulating (IP-Adapter) form stay on the flash kernel
  static const bool no_short = getenv("MI355X_SD_ATTN_NO_SHORT") != nullptr;
  if (DP == 64 && a.D == 64 && a.Skv <= 128 && !a.bias && a.accum == 0.f && !no_short && !(a.o_ts & 7) && !(a.o_bs & 7) &&
      !(reinterpret_cast<uintptr_t>(a.O) & 15)) {
    const int ntq = (a.Sq + QBLK - 1) / QBLK;
    // query tiles per block: ONE. Measured per launch at 8x20x1024x77 / 8x10x4096x77 (profiles/r03_s5_attn_short.txt): 1 / 2 / 4 / 8 tiles
    // per block = 14.8 / 15.2 / 17.1 / 19.5 us and 21.3 / 22.3 / 22.5 / 26.3 us (flash kernel: 21.2 / 31.6 us) -- re-staging 20 KB
    // of L2-resident K / V per tile costs less than the parallelism lost to longer blocks
    const int qtpb = 1;
    const int nqb = (ntq + qtpb - 1) / qtpb;
    dim3 grid(nqb * a.B * a.H), block(ATT_THREADS);
    const int nsb = (a.Skv + 31) / 32;
#define SD_SHORT(N)                                                                                                  \
    do {                                                                                                               \
      if (a.log2) hipLaunchKernelGGL((attention_short_kernel<N, true>), grid, block, 0, stream, a, qtpb);              \
      else hipLaunchKernelGGL((attention_short_kernel<N, false>), grid, block, 0, stream, a, qtpb);                    \
    } while (0)
    if (nsb == 1) SD_SHORT(1);
    else if (nsb == 2) SD_SHORT(2);
    else if (nsb == 3) SD_SHORT(3);
    else SD_SHORT(4);
#undef SD_SHORT
    return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
  }
  static const bool no_qt = getenv("MI355X_SD_ATTN_NO_QT") != nullptr;
  const long qtiles = (long)((a.Sq + QBLK - 1) / QBLK) * a.B * a.H;
  if (DP == 64 && a.Skv <= 2 * KVBLK && qtiles >= 1024 && !a.bias && !no_qt) {
    constexpr int QT = 2;
    constexpr int Q = (DP == 64 ? QT : 1);
    const int nqb = (a.Sq + QBLK * QT - 1) / (QBLK * QT);
    dim3 grid(nqb * a.B * a.H), block(ATT_THREADS);
    // (the two-query-tile kernel keeps the exact maximum: lazy would cost it an occupancy step, 168 -> 199 VGPRs)
    if (DP == 64 && a.log2) hipLaunchKernelGGL((attention_kernel<DP, false, Q, (DP == 64 ? 3 : 0)>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((attention_kernel<DP, false, Q, 0>), grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
  }
  const int nqb = (a.Sq + QBLK - 1) / QBLK;
  dim3 grid(nqb * a.B * a.H), block(ATT_THREADS);
  static const bool no_wide = getenv("MI355X_SD_ATTN_NO_WIDE") != nullptr;   // A/B switch
  const bool wide = !no_wide && a.accum == 0.f && !(a.o_ts & 7) && !(a.o_bs & 7) && !(reinterpret_cast<uintptr_t>(a.O) & 15) && !(a.D & 15);
  // d = 64 without mask, 16-byte O rows: the software-pipelined loop (MI355X_SD_ATTN_NO_IL: A/B switch)
  // MI355X_SD_ATTN_IL (A/B switch): 0 the non-pipelined kernel; 1 pipelined (default); 2 pipelined, pinned order; +2: the log2 form
  // without its C-operand reference (one more VALU per score, 16 fewer registers)
  static const int il = [] {
    const char* e = getenv("MI355X_SD_ATTN_IL");
    return e ? atoi(e) : 1;
  }();
  if (DP == 64 && a.D == 64 && !a.bias && wide && lazy && il > 0) {
    const bool l2 = a.log2 && (il <= 2 || il == 5);
    const bool pin = il == 2 || il == 4;
    static const int abl = [] {
      const char* e = getenv("MI355X_SD_ATTN_ABL");
      return e ? atoi(e) : 0;
    }();
    if (abl && a.log2) {   // timing experiments (wrong results): one switch per ablated part
#define SD_ABL(N) if (abl == N) { hipLaunchKernelGGL((attention_il_kernel<true, 0, N>), grid, block, 0, stream, a); return SD_OK; }
      SD_ABL(64)
#undef SD_ABL
    }
    if (il == 6 || il == 7) {   // two query tiles per wave, one wave per SIMD
      const int nq2 = (a.Sq + 2 * QBLK - 1) / (2 * QBLK);
      dim3 grid2(nq2 * a.B * a.H);
      if (a.log2 && il == 6) hipLaunchKernelGGL((attention_q2_kernel<true, 1>), grid2, block, 0, stream, a);
      else if (a.log2) hipLaunchKernelGGL((attention_q2_kernel<true, 0>), grid2, block, 0, stream, a);
      else hipLaunchKernelGGL((attention_q2_kernel<false, 1>), grid2, block, 0, stream, a);
      return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
    }
    if (il == 5 && a.log2) {
      hipLaunchKernelGGL((attention_il_kernel<true, 2>), grid, block, 0, stream, a);
      return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
    }
    if (l2 && pin) hipLaunchKernelGGL((attention_il_kernel<true, 1>), grid, block, 0, stream, a);
    else if (l2) hipLaunchKernelGGL((attention_il_kernel<true, 0>), grid, block, 0, stream, a);
    else if (pin) hipLaunchKernelGGL((attention_il_kernel<false, 1>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((attention_il_kernel<false, 0>), grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
  }
  if (a.bias) {
    hipLaunchKernelGGL((attention_kernel<DP, true, 1, 0>), grid, block, 0, stream, a);
  } else if (DP == 64 && a.log2) {
    static const bool stamp = getenv("MI355X_SD_ATTN_ABL") && atoi(getenv("MI355X_SD_ATTN_ABL")) == 64;
    if (wide && stamp) hipLaunchKernelGGL((attention_kernel<DP, false, 1, (DP == 64 ? 15 : 0)>), grid, block, 0, stream, a);
    else if (wide) hipLaunchKernelGGL((attention_kernel<DP, false, 1, (DP == 64 ? 7 : 0)>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((attention_kernel<DP, false, 1, (DP == 64 ? 3 : 0)>), grid, block, 0, stream, a);
  } else if (lazy) {
    if (wide) hipLaunchKernelGGL((attention_kernel<DP, false, 1, (DP == 64 ? 5 : 0)>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((attention_kernel<DP, false, 1, (DP == 64 ? 1 : 0)>), grid, block, 0, stream, a);
  } else {
    hipLaunchKernelGGL((attention_kernel<DP, false, 1, 0>), grid, block, 0, stream, a);
  }
  return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
}

int launch_attention(const AttnArgs& a, hipStream_t stream) {
  if (a.B <= 0 || a.H <= 0 || a.Sq <= 0 || a.Skv <= 0 || a.D <= 0) return SD_ERR_INVALID;
  if ((a.D & 7) || a.D > 160) return SD_ERR_UNSUPPORTED;
  if ((a.q_ts & 7) || (a.k_ts & 7) || (a.v_ts & 7) || (a.o_ts & 3) || (a.q_bs & 7) || (a.k_bs & 7) || (a.v_bs & 7) ||
      (a.o_bs & 3))
    return SD_ERR_UNSUPPORTED;
  if (!(a.scale > 0.f)) return SD_ERR_INVALID;
  if (a.D <= 64) return launch_dp<64>(a, stream);
  if (a.D <= 96) return launch_dp<96>(a, stream);
  return launch_dp<160>(a, stream);
}

}  // namespace sd
