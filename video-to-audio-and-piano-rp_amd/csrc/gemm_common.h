// Shared pieces of the MFMA GEMM kernels (gemm.hip, gemm_8phase.hip): kernel parameter block, register staging helpers
// of the v1 kernel, the direct and the LDS-staged fused epilogues.  Everything lives in an anonymous namespace: each
// translation unit gets its own copy.
#pragma once
#include "v2a_common.h"
#include <stdlib.h>
#include <type_traits>

namespace v2a_detail {

struct GemmParams {
  const void* a[3];
  int64_t lda[3];
  int32_t kend[3];  // cumulative K end of each segment
  int32_t nseg;
  const void* w;
  int64_t ldw;
  const float* bias;
  int32_t M, N, K;
  void* out;
  int64_t ldo;
  bf16_t* out2;      // optional bf16 shadow of an fp32 output (operand of a later GEMM)
  int64_t ldo2;
  const float* resid;
  int64_t ldr;
  const float* gate;
  const int32_t* step;
  int64_t gss, gbs;
  int32_t rpb;
  const float* rope;     // cos/sin table [pos][32][2] or null: rotate interleaved pairs of columns < rope_cols (STORE epilogue)
  int32_t rope_cols, rope_pos_off;
  int32_t vec_epi;  // all epilogue pointers / strides allow 16-byte row pieces
  int32_t relu;     // clamp the result at zero (conv + BatchNorm + ReLU blocks of the Video2Roll encoder)
  // implicit-GEMM convolution (DMA kernel only): A row m starts at a[0] + a_rowoff[m], K tile kt adds a_koff[kt] (elements);
  // out / resid / out2 row m starts at base + o_rowoff[m] instead of m * ld
  const int32_t* a_rowoff;
  const int32_t* a_koff;
  const int32_t* o_rowoff;
  int32_t krot;     // start each M band's K walk at a different K tile (see the kernel)
  // RMSNorm folded into its neighbours (v2a_gemm_args: norm_gamma .. row_norm_dim).  Producer: out2 = bf16(out * gamma),
  // ssq[m][n / 32] = sum of squares of 32 output columns.  Consumer: acc row m is scaled by rnorm / max(sqrt(sum_j rssq[m][j]), eps)
  const float* ngam;
  int64_t ngss, ngbs;
  int32_t nsw_row, nsw_off;
  float* ssq;
  int64_t ssq_ld;
  const float* rssq;
  int64_t rssq_ld;
  int32_t rssq_parts;
  float rnorm;
  int32_t xcd_gm, xcd_gn;   // the 8 XCDs of the launch as a gm x gn grid over the tile space (gm * gn == 8): see tile_of_block
  // the rectangles of the tile order, filled on the host once the tile shape is known (fill_tile_map): first tile, height and
  // tile count of each, and 1 / height so that the workgroup's tile costs one multiply and a correction instead of ~40 integer
  // divisions at the head of every kernel
  struct Rect { int32_t m_lo, n_lo, cnt, hm; float rcp; } rect[8];
  int32_t nrect, tiles_m, tiles_n;
  // split (hi | lo plane) outputs of the bf16x3 mode: out2 = the shadow (lo plane N columns after hi), out = the GEGLU hidden
  // (lo plane N / 2 columns after hi)
  int32_t out2_split, out_split;
  int64_t alo[3];           // split operands: elements from a row's hi plane to its lo plane, per segment (default: the segment's K extent)
  int64_t out2_lo;          // split shadow: elements from the hi plane to the lo plane of a shadow row (default: N)
  // 8-phase kernel, split (hi | lo plane) operands: s3_kl = the LOGICAL K (sum of the segments' extents), K = 3 * s3_kl, and the K loop
  // walks the logical K three times: pass 0 = A_hi x W_hi, pass 1 = A_hi x W_lo, pass 2 = A_lo x W_hi -- every pass over all (up to
  // three) logical segments, whose rows are [hi k | lo k] (lda >= 2k), against weight rows [W_hi (s3_kl) | W_lo (s3_kl)].  0 = plain operands
  int32_t s3_kl;
  int32_t dbg;              // probe builds only (-DV2A_GEMM_PROBE, scripts/probes/kloop_probe.py): K-loop parts switched off by bit
};

// tile-shape selectors of the LDS-DMA bf16 kernels (v2a_set_tuning)
struct GemmTuning {
  int force_tile;        // -1 = by shape
  int krot;              // K rotation (see gemm_bf16_dma_kernel)
  int use_8phase;        // 256x256 phase-interleaved kernel for wide outputs: 0 off, 1 staggered wave rows, 2 lock-step
  int min_tiles_8phase;  // ... when the problem has at least this many 256x256 tiles
  int xcd_grid;          // 1: XCD rectangle grid chosen per shape, 0: always 1 x 8 (every XCD walks all M of its column strip)
  int dbg;               // probe builds only: v2a_tuning.reserved[0]
};
extern GemmTuning g_gemm_tuning;
extern int g_attn_one_group_from;    // v2a_attention: workgroup count from which one wave group per workgroup is used
extern int g_dwconv_rows_per_wave;   // 4 or 8 (v2a_set_tuning)
extern int g_probe_dbg;              // v2a_tuning.reserved[0] (probe builds)
extern int g_dwconv_stream;          // 0: never use the streaming depthwise conv (v2a_tuning.dwconv_rows_per_wave = -1)

// 256x256 8-phase kernel (gemm_8phase.hip)
int launch_gemm_8phase(const GemmParams& p, int epilogue, int out_dtype, hipStream_t stream);

// host: the gm x gn rectangles of the tile space for a BM x BN tile shape (xcd_gm / xcd_gn chosen by v2a_gemm)
inline void fill_tile_map(GemmParams& p, int BM, int BN) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const int gm = p.xcd_gm, gn = p.xcd_gn;
  p.nrect = gm * gn;
  for (int r = 0; r < p.nrect; ++r) {
    const int xm = r / gn, xn = r % gn;
    const int m_lo = xm * p.tiles_m / gm, m_hi = (xm + 1) * p.tiles_m / gm;
    const int n_lo = xn * p.tiles_n / gn, n_hi = (xn + 1) * p.tiles_n / gn;
    const int hm = m_hi - m_lo;
    p.rect[r].m_lo = m_lo;
    p.rect[r].n_lo = n_lo;
    p.rect[r].cnt = hm * (n_hi - n_lo);
    p.rect[r].hm = hm > 0 ? hm : 1;
    p.rect[r].rcp = 1.0f / (float)(hm > 0 ? hm : 1);
  }
}

}  // namespace v2a_detail

namespace {

using v2a_detail::GemmParams;

// Workgroup -> output tile, XCD-aware.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the L2 a block
// uses), every XCD has its own 4 MB L2, and an operand panel shared by workgroups on different XCDs is fetched once per XCD.
// The tile space (tiles_m x tiles_n) is therefore cut into a gm x gn grid of rectangles, one per XCD label: the A rows of a
// rectangle are then pulled by gn XCDs and the W rows by gm XCDs, and the host picks (gm, gn) to minimise
// A_bytes * gn + W_bytes * gm (wide outputs: 1 x 8, W read once; narrow ones with long K: 4 x 2 or 2 x 4).  Tiles are
// numbered rectangle by rectangle (M-fastest inside one, so consecutive workgroups of an XCD share a W panel) and label x takes
// the x-th contiguous chunk of that order; chunk and rectangle sizes differ by at most a few tiles, which costs locality only.
// position L of the linear tile order (rectangle by rectangle, M-fastest inside one) -> tile coordinates
__device__ __forceinline__ void tile_of_index(const GemmParams& p, int L, int& tm, int& tn) {
  tm = tn = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if (r < p.nrect) {
      const int cnt = p.rect[r].cnt;
      if (L >= 0 && L < cnt) {
        const int hm = p.rect[r].hm;
        int q = (int)((float)L * p.rect[r].rcp);           // L / hm within one for L < 2^24 (checked on the host), corrected below
        int rem = L - q * hm;
        if (rem < 0) { --q; rem += hm; }
        else if (rem >= hm) { ++q; rem -= hm; }
        tn = p.rect[r].n_lo + q;
        tm = p.rect[r].m_lo + rem;
      }
      L -= cnt;          // negative from the containing rectangle on: no later one matches
    }
  }
}
__device__ __forceinline__ void tile_of_block(const GemmParams& p, int bid, int& tm, int& tn) {
  const int nwg = p.tiles_m * p.tiles_n;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);      // bijective: position in the linear order
  tile_of_index(p, L, tm, tn);
}

template <typename T> struct TileCfg;
template <> struct TileCfg<bf16_t> {
  static constexpr int BK = 64;
  static constexpr int LDS_ROW = 64;  // elements per LDS row (128 B)
};
template <> struct TileCfg<float> {
  static constexpr int BK = 16;
  static constexpr int LDS_ROW = 20;  // padded: 80 B rows keep float4 writes aligned, 2-way max on reads
};

// ---- global -> register staging ----------------------------------------------------------
// bf16 tile: ROWS x 64 bf16; thread t covers 16-B chunk (t & 7) of rows (t >> 3) + 32*i.
template <int ROWS, bool SRC_F32> struct StageBf16 {
  static constexpr int N = ROWS / 32;
  bf16x8 r[N];
  __device__ __forceinline__ void load(const void* base, int64_t ld, int row0, int rows_total, int k0, int tid) {
    const int chunk = tid & 7;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      int row = row0 + (tid >> 3) + 32 * i;
      row = row < rows_total ? row : rows_total - 1;
      if constexpr (SRC_F32) {
        const float* p = reinterpret_cast<const float*>(base) + (int64_t)row * ld + k0 + chunk * 8;
        f32x4 lo = *reinterpret_cast<const f32x4*>(p);
        f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = (bf16_t)lo[j];
          v[4 + j] = (bf16_t)hi[j];
        }
        r[i] = v;
      } else {
        const bf16_t* p = reinterpret_cast<const bf16_t*>(base) + (int64_t)row * ld + k0 + chunk * 8;
        r[i] = *reinterpret_cast<const bf16x8*>(p);
      }
    }
  }
  __device__ __forceinline__ void store(bf16_t* lds, int tid) const {
    const int chunk = tid & 7;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      int row = (tid >> 3) + 32 * i;
      *reinterpret_cast<bf16x8*>(lds + row * 64 + ((chunk ^ (row & 7)) << 3)) = r[i];
    }
  }
};

// fp32 tile: ROWS x 16 floats; thread t covers float4 (t & 3) of rows (t >> 2) + 64*i.
template <int ROWS> struct StageF32 {
  static constexpr int N = ROWS / 64;
  f32x4 r[N];
  __device__ __forceinline__ void load(const void* base, int64_t ld, int row0, int rows_total, int k0, int tid) {
    const int c4 = tid & 3;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      int row = row0 + (tid >> 2) + 64 * i;
      row = row < rows_total ? row : rows_total - 1;
      const float* p = reinterpret_cast<const float*>(base) + (int64_t)row * ld + k0 + c4 * 4;
      r[i] = *reinterpret_cast<const f32x4*>(p);
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid) const {
    const int c4 = tid & 3;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      int row = (tid >> 2) + 64 * i;
      *reinterpret_cast<f32x4*>(lds + row * 20 + c4 * 4) = r[i];
    }
  }
};

template <typename T, bool A_F32, int ROWS> struct StageSel;
template <bool A_F32, int ROWS> struct StageSel<bf16_t, A_F32, ROWS> { using type = StageBf16<ROWS, A_F32>; };
template <bool A_F32, int ROWS> struct StageSel<float, A_F32, ROWS> { using type = StageF32<ROWS>; };

// ---- shared epilogue: C/D layout col = lane & 15, row = (lane >> 4) * 4 + j ------------------
template <int EPI, typename OutT, int TM, int TN, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                              int lr, int lq) {
  OutT* out = reinterpret_cast<OutT*>(p.out);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int m = m0 + wm * WM + i * 16 + lq * 4 + jj;
      if (m >= p.M) continue;
      const float* gvec = nullptr;
      if constexpr (EPI == V2A_EPI_GATE_RESID) gvec = step_vec(p.gate, p.step, p.gss, p.gbs, p.gbs ? m / p.rpb : 0);
      if constexpr (EPI == V2A_EPI_GEGLU) {
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
          const int n = n0 + wn * WN + j * 16 + lr;  // packed row index of the value
          if (n >= p.N) continue;
          float v = acc[i][j][jj], g = acc[i][j + 1][jj];
          if (p.bias) { v += p.bias[n]; g += p.bias[n + 16]; }
          const int oc = ((n0 + wn * WN) >> 1) + (j >> 1) * 16 + lr;
          const float ge = sizeof(OutT) == 2 ? gelu_fast_f(g) : gelu_erf_f(g);
          out[(int64_t)m * p.ldo + oc] = from_f32<OutT>(v * ge);
        }
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * WN + j * 16 + lr;
          if (n >= p.N) continue;
          float v = acc[i][j][jj];
          if (p.bias) v += p.bias[n];
          if constexpr (EPI == V2A_EPI_SIGMOID) v = sigmoid_f(v);
          if constexpr (EPI == V2A_EPI_RESID) v += p.resid[(int64_t)m * p.ldr + n];
          if constexpr (EPI == V2A_EPI_GATE_RESID) v = p.resid[(int64_t)m * p.ldr + n] + gvec[n] * v;
          if (p.relu) v = fmaxf(v, 0.f);
          out[(int64_t)m * p.ldo + n] = from_f32<OutT>(v);
          if constexpr (sizeof(OutT) == 4) {
            if (p.out2) p.out2[(int64_t)m * p.ldo2 + n] = (bf16_t)v;
          }
        }
      }
    }
  }
}


// ---- LDS-staged epilogue (DMA kernels) -----------------------------------------------------
// The MFMA C layout gives a lane 4 ROWS of one column, so direct stores are 2-4 byte pieces.
// Instead every wave parks its WM x WN fp32 tile in a private LDS region (the K-loop stages are
// dead by then), and reads it back row-major 4 columns per lane: bias / GELU / gate / residual run
// on float4s and every global access is a 16-byte (fp32) or 8-byte (bf16) piece of a contiguous row.
// Residual / gate pieces of the vector epilogue, fetched at kernel start so their L2/HBM latency hides under the K loop
// instead of being paid once per 16-row slab at the tail of every workgroup (same lane -> (row, 4 columns) map as below).
template <int EPI, int TM, int WN, bool ON> struct EpiPrefetch {
  static constexpr int LPR = WN / 4, RPI = 64 / LPR, RPS = 16 / RPI;   // lanes per row, rows per pass, rows per slab and lane
  static constexpr bool RES = ON && (EPI == V2A_EPI_RESID || EPI == V2A_EPI_GATE_RESID);
  static constexpr bool GATE = ON && EPI == V2A_EPI_GATE_RESID;
  f32x4 rs[RES ? TM : 1][RES ? RPS : 1];
  f32x4 gt[GATE ? TM : 1][GATE ? RPS : 1];
  static constexpr bool SCAT = RES && !GATE;   // offset tables come with STORE / RESID only (checked on the host)
  int32_t ro[SCAT ? TM : 1][SCAT ? RPS : 1];  // out / resid row offsets when rows are scattered (o_rowoff)
  static constexpr bool ROPE = ON && EPI == V2A_EPI_STORE;
  f32x4 cs[ROPE ? TM : 1][ROPE ? RPS : 1];     // (cos, sin) of the two column pairs a lane rotates
  __device__ __forceinline__ void load(const GemmParams& p, int m_base, int n_base, int lane) {
    if constexpr (ROPE) {
      const int c4 = (lane % LPR) * 4, r0 = lane / LPR;
      const int n = n_base + c4;
      if (p.rope && n < p.rope_cols) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < RPS; ++q) {
            int m = m_base + i * 16 + r0 + q * RPI;
            m = m < p.M ? m : p.M - 1;
            const int pos = p.rope_pos_off + m % p.rpb;
            cs[i][q] = *reinterpret_cast<const f32x4*>(p.rope + ((int64_t)pos * 32 + ((n & 63) >> 1)) * 2);
          }
      }
    }
    if constexpr (RES) {
      const int c4 = (lane % LPR) * 4, r0 = lane / LPR;
      const int n = n_base + c4;
      // kernel arguments copied once: the loops below stay free of scalar re-loads and branches
      const int M = p.M;
      const bool full = n + 3 < p.N;
      const bool has_res = p.resid != nullptr, has_gate = p.gate != nullptr;     // wave-uniform
      const float* resid = p.resid + n;
      const int64_t ldr = p.ldr;
      const int32_t* orow = SCAT ? p.o_rowoff : nullptr;
      // scattered rows (implicit-GEMM convolution): all row offsets are requested before the first residual row depends on one
      if constexpr (SCAT) {
        if (orow) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < RPS; ++q) {
              int m = m_base + i * 16 + r0 + q * RPI;
              m = m < M ? m : M - 1;
              ro[i][q] = orow[m];
            }
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < RPS; ++q) {
          const int m = m_base + i * 16 + r0 + q * RPI;
          if (m < M && full) {
            int64_t off = (int64_t)m * ldr;
            if constexpr (SCAT) { if (orow) off = ro[i][q]; }
            // absent operands read as 0 / 1 (resid + 1 * acc is exactly the RESID epilogue, 0 + 1 * acc the plain store)
            rs[i][q] = has_res ? *reinterpret_cast<const f32x4*>(resid + off) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (GATE) gt[i][q] = has_gate ? *reinterpret_cast<const f32x4*>(step_vec(p.gate, p.step, p.gss, p.gbs, p.gbs ? m / p.rpb : 0) + n) : f32x4{1.f, 1.f, 1.f, 1.f};
          }
        }
    }
  }
};

// Folded RMSNorm, consumer side: 1 / rms of A row m (F.normalize: x / max(||x||, 1e-12), times sqrt(d)) from the sums of squares
// the producer of that row left per 32 columns.  One thread per tile row requests the row's partial sums when the kernel starts
// (before the first operand DMA, so waiting for them drains nothing), adds them up in a fixed order -- the scale does not depend
// on the tile shape -- and leaves the scale in LDS behind the ring for the epilogue.
constexpr int kRowSsqMax = 40;                // partial sums per row: d <= 1280
struct RowScaleLoad {
  f32x4 v[kRowSsqMax / 4];
};
__device__ __forceinline__ void rowscale_load(const GemmParams& p, int m, RowScaleLoad& r) {
  m = m < p.M ? m : p.M - 1;
  const float* q = p.rssq + (int64_t)m * p.rssq_ld;
#pragma unroll
  for (int k = 0; k < kRowSsqMax / 4; ++k) {
    r.v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (4 * k < p.rssq_parts) r.v[k] = *reinterpret_cast<const f32x4*>(q + 4 * k);
  }
}
__device__ __forceinline__ float rowscale_finish(const GemmParams& p, const RowScaleLoad& r) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kRowSsqMax / 4; ++k) s += (r.v[k][0] + r.v[k][1]) + (r.v[k][2] + r.v[k][3]);
  return p.rnorm / fmaxf(sqrtf(s), 1e-12f);
}

template <int EPI, typename OutT, int TM, int TN, int WM, int WN, bool PF>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmParams& p, f32x4 (&acc)[TM][TN], float* tile /* wave-private, 16 x (WN+4) floats */,
                                                  int m_base, int n_base, int lane, const EpiPrefetch<EPI, TM, WN, PF>& pf,
                                                  const float* rs_row = nullptr /* LDS: row scales of this wave's rows, or null */) {
  constexpr int LD = WN + 4;                   // 16-B aligned rows, <= 2-way write conflicts
  const int lr = lane & 15, lq = lane >> 4;
  OutT* out = reinterpret_cast<OutT*>(p.out);
  // kernel arguments copied once (the slab loops below otherwise re-load them from the argument segment per row)
  const int M = p.M;
  const bool relu = p.relu != 0;
  const int32_t* orow = p.o_rowoff;           // scattered rows (implicit-GEMM convolution into a bordered map) or null
  const int64_t ldo = p.ldo, ldr = p.ldr, ldo2 = p.ldo2;
  bf16_t* out2 = p.out2;
  const float* resid = p.resid;
  // folded RMSNorm, consumer side: one scale per row this lane touches, all requested before the first slab is staged
  constexpr int LPRX = (EPI == V2A_EPI_GEGLU ? WN / 2 : WN) / 4;   // lanes per row of the store loops below
  constexpr int RPSX = 16 / (64 / LPRX);                           // rows per lane per slab
  // folded RMSNorm, producer side: the gamma pieces of this lane's four columns, loaded once (they depend on the row only through
  // the switch row, or through the batch when every clip has its own time -- then they are fetched per row below)
  f32x4 gmA = {1.f, 1.f, 1.f, 1.f}, gmB = gmA;
  if constexpr ((EPI == V2A_EPI_RESID || EPI == V2A_EPI_GATE_RESID) && sizeof(OutT) == 4) {
    const int nn = n_base + (lane % (WN / 4)) * 4;
    if (out2 && p.ngam && p.ngbs == 0 && nn + 3 < p.N) {
      const float* gb = step_vec(p.ngam, p.step, p.ngss, 0, 0) + nn;
      gmA = *reinterpret_cast<const f32x4*>(gb);
      gmB = *reinterpret_cast<const f32x4*>(gb + p.nsw_off);
    }
  }
  float rsc[TM][RPSX];
  const bool scaled = rs_row != nullptr;                           // wave-uniform
  // GEGLU with 16-byte stores (below): needs 16-byte aligned output rows and planes
  const bool geglu_wide = EPI == V2A_EPI_GEGLU && sizeof(OutT) == 2 && ((uintptr_t)p.out & 15) == 0 && (p.ldo & 7) == 0 && ((p.N >> 1) & 7) == 0 &&
                          !(p.dbg & 128);       // v2a_tuning.reserved[0] bit 7: the four-column form (A/B)
  if (scaled) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < RPSX; ++q) rsc[i][q] = rs_row[i * 16 + lane / LPRX + q * (64 / LPRX)];
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (m_base + i * 16 >= M) continue;          // a slab wholly behind the last row (wave-uniform): nothing to stage or store
    // one 16-row slab of the wave tile at a time: the staging area of a workgroup is a few KB of one ring stage
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) tile[(lq * 4 + jj) * LD + j * 16 + lr] = acc[i][j][jj];
    // same wave wrote and reads: LDS operations of one wave complete in order, no barrier needed
    if constexpr (EPI == V2A_EPI_GEGLU && sizeof(OutT) == 2 && (WN / 2) % 32 == 0) {
      // bf16 / hi | lo outputs of a wave tile with >= 32 output columns: EIGHT columns per lane, so that every global store is 16 bytes (the
      // 8-byte pieces of the four-column form below made the 128 KB of a 256x256 tile's planes a store-issue-bound tail); taken when the
      // rows allow it (wave-uniform test), the four-column form otherwise
      if (geglu_wide) {
        constexpr int OC = WN / 2, LPR = OC / 8, RPI = 64 / LPR;
        static_assert(RPI <= 16, "one slab holds 16 rows");
        const int c8 = (lane % LPR) * 8, r0 = lane / LPR;
        const int lc = (c8 >> 4) * 32 + (c8 & 15);            // LDS column of the first value; the gates lie 16 further
        const int n = n_base + lc;
        f32x4 bv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, bg[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (p.bias && n < p.N) {
          bv[0] = *reinterpret_cast<const f32x4*>(p.bias + n);
          bv[1] = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
          bg[0] = *reinterpret_cast<const f32x4*>(p.bias + n + 16);
          bg[1] = *reinterpret_cast<const f32x4*>(p.bias + n + 20);
        }
#pragma unroll
        for (int q = 0; q < 16 / RPI; ++q) {
          const int r = r0 + q * RPI;
          const int m = m_base + i * 16 + r;
          if (m >= p.M || n >= p.N) continue;
          const float rs = scaled ? rs_row[i * 16 + r] : 1.f;
          bf16x8 hi, lo;
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            f32x4 v = *reinterpret_cast<const f32x4*>(tile + r * LD + lc + 4 * h2);
            f32x4 g = *reinterpret_cast<const f32x4*>(tile + r * LD + lc + 16 + 4 * h2);
            if (scaled) {          // (the same expressions as the four-column form: equal bit for bit)
              v *= rs;
              g *= rs;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float o32 = (v[e] + bv[h2][e]) * gelu_fast_f(g[e] + bg[h2][e]);
              hi[4 * h2 + e] = (bf16_t)o32;
              lo[4 * h2 + e] = (bf16_t)(o32 - (float)hi[4 * h2 + e]);
            }
          }
          OutT* dst = out + (int64_t)m * p.ldo + (n_base >> 1) + c8;
          *reinterpret_cast<bf16x8*>(dst) = hi;
          if (p.out_split) *reinterpret_cast<bf16x8*>(dst + (p.N >> 1)) = lo;
        }
        continue;
      }
    }
    if constexpr (EPI == V2A_EPI_GEGLU) {
      constexpr int OC = WN / 2;                 // output columns of this wave
      constexpr int LPR = OC / 4;                // lanes per row
      constexpr int RPI = 64 / LPR;              // rows per pass
      const int c4 = (lane % LPR) * 4, r0 = lane / LPR;
      const int lc = (c4 >> 4) * 32 + (c4 & 15); // LDS column of the value; gate is 16 further
      const int n = n_base + lc;                 // packed W row of the value
      f32x4 bv = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
      if (p.bias && n < p.N) {
        bv = *reinterpret_cast<const f32x4*>(p.bias + n);
        bg = *reinterpret_cast<const f32x4*>(p.bias + n + 16);
      }
#pragma unroll
      for (int q = 0; q < 16 / RPI; ++q) {
        const int r = r0 + q * RPI;
        const int m = m_base + i * 16 + r;
        if (m >= p.M || n >= p.N) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(tile + r * LD + lc);
        f32x4 g = *reinterpret_cast<const f32x4*>(tile + r * LD + lc + 16);
        if (scaled) {
          v *= rsc[i][q];
          g *= rsc[i][q];
        }
        OutT* dst = out + (int64_t)m * p.ldo + (n_base >> 1) + c4;
        if constexpr (sizeof(OutT) == 2) {
          if (p.out_split) {
            // bf16x3 mode: fp32 GELU stored as hi | lo planes (lo plane N / 2 columns further).  erf by Abramowitz-Stegun 7.1.26
            // (|error| <= 1.5e-7, two orders below the 2^-17 the planes keep): libm's branchy erff cost ~20 us per 256x256 tile round
            // here against ~6 (64 values per lane), a fifth of a feed-forward launch at 8 clips per GPU
            bf16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float o32 = (v[e] + bv[e]) * gelu_fast_f(g[e] + bg[e]);
              hi[e] = (bf16_t)o32;
              lo[e] = (bf16_t)(o32 - (float)hi[e]);
            }
            *reinterpret_cast<bf16x4*>(dst) = hi;
            *reinterpret_cast<bf16x4*>(dst + (p.N >> 1)) = lo;
          } else {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)((v[e] + bv[e]) * gelu_fast_f(g[e] + bg[e]));
            *reinterpret_cast<bf16x4*>(dst) = o;
          }
        } else {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (v[e] + bv[e]) * gelu_erf_f(g[e] + bg[e]);
          *reinterpret_cast<f32x4*>(dst) = o;
        }
      }
    } else {
      constexpr int LPR = WN / 4;
      constexpr int RPI = 64 / LPR;
      const int c4 = (lane % LPR) * 4, r0 = lane / LPR;
      const int n = n_base + c4;
      const bool full = n + 3 < p.N;             // N is a multiple of 4 for every vector-eligible call (checked on the host)
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias && full) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
      for (int q = 0; q < 16 / RPI; ++q) {
        const int r = r0 + q * RPI;
        const int m = m_base + i * 16 + r;
        if (m >= M || !full) continue;
        int64_t o_out = (int64_t)m * ldo, o_res = (int64_t)m * ldr, o_out2 = (int64_t)m * ldo2;
        if constexpr (EPI == V2A_EPI_STORE || EPI == V2A_EPI_RESID) {     // offset tables come with STORE / RESID only
          if (orow) {
            int64_t ro;
            if constexpr (PF && EPI == V2A_EPI_RESID) ro = pf.ro[i][q];
            else ro = orow[m];
            o_out = o_res = o_out2 = ro;
          }
        }
        f32x4 v = *reinterpret_cast<const f32x4*>(tile + r * LD + c4);
        if (scaled) v *= rsc[i][q];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bv[e];
        if constexpr (EPI == V2A_EPI_SIGMOID) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = sigmoid_f(v[e]);
        }
        if constexpr (EPI == V2A_EPI_STORE) {
          if (p.rope && n < p.rope_cols) {
            // interleaved RoPE (A6): columns (n, n+1) and (n+2, n+3) are pairs (n & 63) / 2 and +1 of this head
            f32x4 cs;
            if constexpr (PF) cs = pf.cs[i][q];
            else cs = *reinterpret_cast<const f32x4*>(p.rope + ((int64_t)(p.rope_pos_off + m % p.rpb) * 32 + ((n & 63) >> 1)) * 2);
            const float a0 = v[0], b0 = v[1], a1 = v[2], b1 = v[3];
            // one fixed contraction: left to -ffp-contract the two kernels' instantiations picked different multiply-add
            // pairings and the rotated columns differed in the last bit between tile shapes
            v[0] = fmaf(a0, cs[0], -(b0 * cs[1]));
            v[1] = fmaf(b0, cs[0], a0 * cs[1]);
            v[2] = fmaf(a1, cs[2], -(b1 * cs[3]));
            v[3] = fmaf(b1, cs[2], a1 * cs[3]);
          }
        }
        if constexpr (EPI == V2A_EPI_RESID || EPI == V2A_EPI_GATE_RESID) {
          f32x4 rs, gt = {1.f, 1.f, 1.f, 1.f};
          if constexpr (PF) {
            rs = pf.rs[i][q];
            if constexpr (EPI == V2A_EPI_GATE_RESID) gt = pf.gt[i][q];
          } else {
            rs = resid ? *reinterpret_cast<const f32x4*>(resid + o_res + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == V2A_EPI_GATE_RESID) {
              if (p.gate) gt = *reinterpret_cast<const f32x4*>(step_vec(p.gate, p.step, p.gss, p.gbs, p.gbs ? m / p.rpb : 0) + n);
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rs[e] + gt[e] * v[e];
        }
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (sizeof(OutT) == 2) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
          *reinterpret_cast<bf16x4*>(out + o_out + n) = o;
        } else {
          *reinterpret_cast<f32x4*>(out + o_out + n) = v;
          if (out2) {
            // folded RMSNorm, producer side: the shadow carries the norm's gamma (the consumer applies 1 / rms per row) and
            // the row's sum of squares is left per 32 columns (8 lanes x 4 columns: a butterfly inside the lane octet)
            f32x4 gm = m >= p.nsw_row ? gmB : gmA;
            if (p.ngam && p.ngbs) gm = *reinterpret_cast<const f32x4*>(step_vec(p.ngam, p.step, p.ngss, p.ngbs, m / p.rpb) + (m >= p.nsw_row ? p.nsw_off : 0) + n);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(v[e] * gm[e]);
            *reinterpret_cast<bf16x4*>(out2 + o_out2 + n) = o;
            if (p.out2_split) {      // bf16x3 mode: the lo plane of the (gamma-scaled) row, N columns further
              bf16x4 lo;
#pragma unroll
              for (int e = 0; e < 4; ++e) lo[e] = (bf16_t)(v[e] * gm[e] - (float)o[e]);
              *reinterpret_cast<bf16x4*>(out2 + o_out2 + p.out2_lo + n) = lo;
            }
            if (p.ssq) {
              const float ss = octet_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
              if ((lane & 7) == 0) p.ssq[(int64_t)m * p.ssq_ld + (n >> 5)] = ss;
            }
          }
        }
      }
    }
  }
}

}  // namespace
