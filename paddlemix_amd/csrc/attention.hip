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
}

// ---- d = 64 attention on v_mfma_f32_16x16x32 (round 5) -------------------------------------------------------------------------------
// Why another MFMA shape. The d = 64 launches of attention_kernel are not stall-bound but CLOCK-bound: the kernel stamps s_memtime
// against the 100-MHz wall clock and reads 1.5 - 1.6 GHz while the SDXL S = 4096 launch runs (profiles/r05_s6_attn_clock.txt), and a
// loop that needs 5 % fewer cycles (the software-pipelined attention_il_kernel, its source is in the history of the repository: profiles/experiments/ before round 6) gets the same
// TIME at a lower clock. A dense stream of v_mfma_f32_32x32x16_bf16 on random operands is itself held to 1.22 - 1.75 PFLOP/s by this
// board (1.64 - 1.82 GHz, at only ~1 kW of the ~1.34 kW cap: a current limit, not the power cap), the same stream of
// v_mfma_f32_16x16x32_bf16 sustains 2.02 PFLOP/s at 2.1 GHz (scripts/probes/mfma_shape_energy_probe.hip,
// profiles/r05_s7_mfma_shape_energy.txt) -- the shape the GEMM kernels use. So: the same flash loop on the 16x16x32 shape.
// Per wave 32 queries = 2 query blocks of 16; a K/V tile of 64 keys = 4 key blocks of 16:
//   * S^T = K Q^T: tile (kb, qb) = 2 MFMAs over d (k-steps of 32); lane (query l15 of block qb, g = lane >> 4) gets keys kb*16 + 4g + r;
//   * P^T feeds O^T = V^T P^T straight from those registers: the B operand of k-step kk takes its 8 key slots as
//     [keys of block 2kk | keys of block 2kk+1] of the lane's group g -- the key ORDER of a contraction is free, the V^T A operand is
//     read (ds_read_b64_tr_b16, two [4 keys][16 d] blocks) in the same order; O^T tile (db, qb): lane holds d = db*16 + 4g + r;
//   * row statistics: a lane's partial row sum covers its 16 keys of a tile; the four lane groups are summed once, at the end (lazy
//     reference as in attention_kernel FL bit 0 / 1: no per-tile maximum, hence no per-tile cross-lane traffic at all);
//   * LDS images for these fragment shapes: K rows unpadded (128 B) with the 16-byte chunk index XOR (row >> 1) & 7 -- the 16 lanes
//     of a ds_read_b128 group then cover the 64 banks exactly once; V rows at a stride of 160 B -- the 8 rows x 32 B a half-wave's
//     transposing read touches tile the 256-byte bank row. 18 KB per stage, 36 KB per block.
// STAMP (debug build, MI355X_SD_ATTN_STAMP): two blocks leave {s_memtime ticks, 100-MHz wall ticks} of their life in the tail of the
// last query row (an input; scripts/c/attn_probe.c prints the shader clock from them -- the launch is clock-bound, see above).
template <bool LOG2, bool STAMP = false>
__global__ __launch_bounds__(ATT_THREADS, 3) void attention16_kernel(const AttnArgs p) {
  constexpr int KROW = 128, VROW = 160, KB16 = KVBLK * KROW, VB16 = KVBLK * VROW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (KB16 + VB16)];
  [[maybe_unused]] unsigned long long clk0 = 0, wall0 = 0;
  if constexpr (STAMP) {
    clk0 = __builtin_readcyclecounter();
    wall0 = wall_clock64();
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int g = lane >> 4;

  const int nqb = (p.Sq + QBLK - 1) / QBLK;
  const int lid = xcd_remap(blockIdx.x, nqb * p.B * p.H);
  const int qb = lid % nqb;
  const int bh = lid / nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const bf16* Qp = p.Q + (size_t)b * p.q_bs + (size_t)h * 64;
  const bf16* Kp = p.K + (size_t)b * p.k_bs + (size_t)h * 64;
  const bf16* Vp = p.V + (size_t)b * p.v_bs + (size_t)h * 64;
  bf16* Op = p.O + (size_t)b * p.o_bs + (size_t)h * 64;

  // ---- K / V tile loader (buffer loads: rows past Skv read as zero; see attention_kernel) ----
  u32x4 rk[2], rv[2];
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(Kp), 0, (unsigned)(((size_t)(p.Skv - 1) * p.k_ts + 64) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(Vp), 0, (unsigned)(((size_t)(p.Skv - 1) * p.v_ts + 64) * 2), 0x00020000);
  const int ld_row = tid >> 3, ld_ch = tid & 7;   // chunk i of this thread: row ld_row + 32 i, 16-byte chunk ld_ch
  const unsigned k_off0 = (unsigned)((ld_row * p.k_ts + ld_ch * 8) * 2), v_off0 = (unsigned)((ld_row * p.v_ts + ld_ch * 8) * 2);
  const unsigned k_step = (unsigned)(32 * p.k_ts * 2), v_step = (unsigned)(32 * p.v_ts * 2);
  auto load_kv = [&](const int t) {
    const unsigned ko = k_off0 + (unsigned)(t * KVBLK * p.k_ts * 2), vo = v_off0 + (unsigned)(t * KVBLK * p.v_ts * 2);
    rk[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, ko, 0, 0));
    rk[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, ko + k_step, 0, 0));
    rv[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, vo, 0, 0));
    rv[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, vo + v_step, 0, 0));
  };
  // (rows ld_row and ld_row + 32 share (row >> 1) & 7: one swizzled chunk position for both)
  const int k_st = ld_row * KROW + ((ld_ch ^ ((ld_row >> 1) & 7)) * 16), v_st = ld_row * VROW + ld_ch * 16;
  auto store_kv = [&](const int buf) {
    unsigned char* kd = smem + buf * (KB16 + VB16) + k_st;
    unsigned char* vd = smem + buf * (KB16 + VB16) + KB16 + v_st;
    *reinterpret_cast<u32x4*>(kd) = rk[0];
    *reinterpret_cast<u32x4*>(kd + 32 * KROW) = rk[1];
    *reinterpret_cast<u32x4*>(vd) = rv[0];
    *reinterpret_cast<u32x4*>(vd + 32 * VROW) = rv[1];
  };

  const int ntiles = (p.Skv + KVBLK - 1) / KVBLK;
  const int nfull = p.Skv / KVBLK;
  // Q fragments (B operand): lane (query q0 + qx*16 + l15, g) holds d = ks*32 + g*8 .. +7
  int q_row[2];
  bool q_ok[2];
  bf16x8 qf[2][2];
#pragma unroll
  for (int qx = 0; qx < 2; ++qx) {
    q_row[qx] = qb * QBLK + wave * QROWS + qx * 16 + l15;
    q_ok[qx] = q_row[qx] < p.Sq;
    const bf16* qr = Qp + (size_t)(q_ok[qx] ? q_row[qx] : 0) * p.q_ts + g * 8;
    qf[qx][0] = *reinterpret_cast<const bf16x8*>(qr);
    qf[qx][1] = *reinterpret_cast<const bf16x8*>(qr + 32);
  }
  load_kv(0);
  store_kv(0);
  __syncthreads();

  // fragment read offsets of this lane
  const int swz = (l15 >> 1) & 7;
  const int kfo0 = l15 * KROW + ((g ^ swz) * 16), kfo1 = l15 * KROW + (((4 + g) ^ swz) * 16);   // k-step 0 / 1 of key block 0
  const int vfo = (g * 4 + (l15 >> 2)) * VROW + (l15 & 3) * 8;                                   // d block 0 of key block 0

  f32x4 o[4][2], sinit[2];
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  const float c2 = LOG2 ? 1.0f : p.scale * 1.4426950408889634f;
#pragma unroll
  for (int qx = 0; qx < 2; ++qx) {
    sinit[qx] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db][qx] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto gmax = [](float x) {   // maximum over the four lane groups that share a query
    x = fmaxf(x, __shfl_xor(x, 16, 64));
    return fmaxf(x, __shfl_xor(x, 32, 64));
  };
  constexpr float RESCALE_THR = 4.0f;

  auto tile_body = [&](const int t, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int buf = t & 1;
    if (t + 1 < ntiles) load_kv(t + 1);
    const unsigned char* ks_ = smem + buf * (KB16 + VB16);
    const unsigned char* vs_ = ks_ + KB16;
    const int kv0 = t * KVBLK;

    // ---- S^T tiles: s[kb][qx][r] = score of key kv0 + kb*16 + 4g + r for query qx*16 + l15 (minus the reference in the LOG2 form) ----
    f32x4 s[4][2];
    auto compute_scores = [&]() {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(ks_ + kb * 16 * KROW + kfo0);
        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(ks_ + kb * 16 * KROW + kfo1);
#pragma unroll
        for (int qx = 0; qx < 2; ++qx) {
          s[kb][qx] = mfma_16x16x32(k0, qf[qx][0], LOG2 ? sinit[qx] : f32x4{0.f, 0.f, 0.f, 0.f});
          s[kb][qx] = mfma_16x16x32(k1, qf[qx][1], s[kb][qx]);
        }
      }
      if (MASK) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kv0 + kb * 16 + g * 4 + r >= p.Skv) s[kb][0][r] = s[kb][1][r] = -INFINITY;
      }
    };
    compute_scores();

    // ---- softmax: lazy reference (exact maximum on the first tile and whenever the guard fires; attention_kernel FL bit 0) ----
    bf16x8 pf[2][2];   // [k-step of P.V][query block]
    float psum[2];
    auto exponentiate = [&](const float (&mc)[2]) {
#pragma unroll
      for (int qx = 0; qx < 2; ++qx) {
        psum[qx] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(LOG2 ? (s[kb][qx][r] - mc[qx]) : __builtin_fmaf(s[kb][qx][r], c2, -mc[qx]));
            psum[qx] += e;
            pf[kb >> 1][qx][(kb & 1) * 4 + r] = (bf16)e;
          }
      }
    };
    auto exponentiate_lazy = [&]() {
#pragma unroll
      for (int qx = 0; qx < 2; ++qx) {
        const float mc0 = m_run[qx] * c2;
        psum[qx] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = LOG2 ? __builtin_amdgcn_exp2f(s[kb][qx][r]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][qx][r], c2, -mc0));
            psum[qx] += e;
            pf[kb >> 1][qx][(kb & 1) * 4 + r] = (bf16)e;
          }
      }
    };
    bool exact = true;
    if (t > 0) {
      exponentiate_lazy();
      exact = __any(__float_as_uint(psum[0]) > __float_as_uint(LAZY_PSUM_LIMIT) || __float_as_uint(psum[1]) > __float_as_uint(LAZY_PSUM_LIMIT));
      if (exact) {
        asm volatile("; lazy-maximum overflow guard fired: scores again, exact path");
        compute_scores();
      }
    }
    if (exact) {
      asm volatile("; exact softmax path (first tile / overflow guard)");
      float mc[2];
#pragma unroll
      for (int qx = 0; qx < 2; ++qx) {
        float mloc = fmaxf(fmaxf(s[0][qx][0], s[0][qx][1]), fmaxf(s[0][qx][2], s[0][qx][3]));
#pragma unroll
        for (int kb = 1; kb < 4; ++kb) mloc = fmaxf(mloc, fmaxf(fmaxf(s[kb][qx][0], s[kb][qx][1]), fmaxf(s[kb][qx][2], s[kb][qx][3])));
        mloc = gmax(mloc);
        if (LOG2) {
          // mloc is relative to the reference m_run (0 on the first tile, where o = l = 0)
          const bool first = t == 0;
          float shift = 0.f;
          if (first || mloc > RESCALE_THR) {
            const float mfin = (mloc == -INFINITY) ? 0.f : mloc;
            const float delta = first ? mfin : fmaxf(mfin, 0.f);
            const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
            l_run[qx] *= alpha;
            m_run[qx] = (first ? 0.f : m_run[qx]) + delta;
            shift = delta;
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db][qx] *= alpha;
            sinit[qx] = f32x4{-m_run[qx], -m_run[qx], -m_run[qx], -m_run[qx]};
          }
          mc[qx] = shift;
        } else {
          if ((mloc - m_run[qx]) * c2 > RESCALE_THR) {
            const float m_new = fmaxf(m_run[qx], mloc);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f((m_run[qx] - m_use) * c2);   // m_run = -inf -> 0
            l_run[qx] *= alpha;
            m_run[qx] = m_use;
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db][qx] *= alpha;
          }
          mc[qx] = m_run[qx] * c2;
        }
      }
      exponentiate(mc);
    }
    l_run[0] += psum[0];
    l_run[1] += psum[1];

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const unsigned char* vp = vs_ + kk * 32 * VROW + vfo + db * 32;
        const bf16x4 lo = ds_read_tr16((lds_bf16x4*)vp);
        const bf16x4 hi4 = ds_read_tr16((lds_bf16x4*)(vp + 16 * VROW));
        const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        o[db][0] = mfma_16x16x32(vf, pf[kk][0], o[db][0]);
        o[db][1] = mfma_16x16x32(vf, pf[kk][1], o[db][1]);
      }

    if (t + 1 < ntiles) store_kv(buf ^ 1);
    __syncthreads();
  };

  for (int t = 0; t < nfull; ++t) tile_body(t, std::false_type{});
  if (nfull < ntiles) tile_body(nfull, std::true_type{});

  // ---- finalize: O[q][d] = O^T[d][q] / l; lane holds d = db*16 + 4g .. +3 of its query; lane groups g, g ^ 1 hold adjacent 8-byte
  // pieces of a row: one v_permlane16_swap per dword pairs them into 16-byte stores (cdna guide T21 on row-of-16 swaps) ----
#pragma unroll
  for (int qx = 0; qx < 2; ++qx) {
    float l_tot = l_run[qx] + __shfl_xor(l_run[qx], 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    const float inv_l = (l_tot > 0.f) ? 1.0f / l_tot : 0.f;
    bf16* orow = Op + (size_t)(q_ok[qx] ? q_row[qx] : 0) * p.o_ts;
#pragma unroll
    for (int db = 0; db < 4; db += 2) {
      // pieces X = (db, this g) and Y = (db + 1, this g). After the swaps an even group holds d = db*16 + 4g .. 4g+7 (its X and its
      // odd neighbour's X), an odd group d = (db+1)*16 + 4(g-1) .. +7 (its neighbour's Y and its own Y).
      unsigned x0 = pack_bf16(o[db][qx][0] * inv_l, o[db][qx][1] * inv_l), x1 = pack_bf16(o[db][qx][2] * inv_l, o[db][qx][3] * inv_l);
      unsigned y0 = pack_bf16(o[db + 1][qx][0] * inv_l, o[db + 1][qx][1] * inv_l), y1 = pack_bf16(o[db + 1][qx][2] * inv_l, o[db + 1][qx][3] * inv_l);
      const auto r0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
      const auto r1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
      const u32x4 pk = {r0[0], r1[0], r0[1], r1[1]};
      const int d0 = (g & 1) ? (db + 1) * 16 + (g - 1) * 4 : db * 16 + g * 4;
      if (q_ok[qx]) *reinterpret_cast<u32x4*>(orow + d0) = pk;
    }
  }
  if constexpr (STAMP) {
    if (tid == 0 && (blockIdx.x == 8 || blockIdx.x == gridDim.x / 2 + 8)) {
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<bf16*>(p.Q) + (size_t)(p.B - 1) * p.q_bs + (size_t)(p.Sq - 1) * p.q_ts + p.H * 64 - 16) + (blockIdx.x == 8 ? 0 : 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      dbg[0] = __builtin_readcyclecounter() - clk0;
      dbg[1] = wall_clock64() - wall0;
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
  static const bool no_short = sd_switch("MI355X_SD_ATTN_NO_SHORT") != nullptr;
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
  static const bool no_qt = sd_switch("MI355X_SD_ATTN_NO_QT") != nullptr;
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
  static const bool no_wide = sd_switch("MI355X_SD_ATTN_NO_WIDE") != nullptr;   // A/B switch
  const bool wide = !no_wide && a.accum == 0.f && !(a.o_ts & 7) && !(a.o_bs & 7) && !(reinterpret_cast<uintptr_t>(a.O) & 15) && !(a.D & 15);
  // d = 64 without mask, 16-byte O rows: the software-pipelined loop (MI355X_SD_ATTN_NO_IL: A/B switch)
  // d = 64 without mask, 16-byte O rows: the loop on the 16x16x32 MFMA shape (MI355X_SD_ATTN_NO_M16: A/B switch of the debug build)
  static const bool no_m16 = sd_switch("MI355X_SD_ATTN_NO_M16") != nullptr;
  if (DP == 64 && a.D == 64 && !a.bias && wide && lazy && !no_m16) {
#ifdef MI355X_SD_DEBUG_SWITCHES
    static const bool stamp = sd_switch("MI355X_SD_ATTN_STAMP") != nullptr;
    if (stamp && a.log2) {
      hipLaunchKernelGGL((attention16_kernel<true, true>), grid, block, 0, stream, a);
      return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
    }
#endif
    if (a.log2) hipLaunchKernelGGL((attention16_kernel<true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((attention16_kernel<false>), grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? SD_OK : SD_ERR_HIP;
  }
  if (a.bias) {
    hipLaunchKernelGGL((attention_kernel<DP, true, 1, 0>), grid, block, 0, stream, a);
  } else if (DP == 64 && a.log2) {
    if (wide) hipLaunchKernelGGL((attention_kernel<DP, false, 1, (DP == 64 ? 7 : 0)>), grid, block, 0, stream, a);
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
