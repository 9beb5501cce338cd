// Cross-attention of the audio stream as ONE launch: q-projection GEMM (+ folded RMSNorm, bias, RoPE) -> QK^T over the nc <= 64
// context keys of the clip -> soft clamp, softmax -> PV -> per-head sigmoid gate -> bf16 output rows for the out-projection.
// Replaces, at the sizes where the two launches were latency bound (one or two clips per GPU), v2a_gemm(STORE, rope) into the
// [q | gate] buffer followed by v2a_attention: xt Attention.forward with context, x3:1126 (call site), attend.py via x3:881-914.
//
// A workgroup = 64 consecutive tokens of ONE sequence x ONE head (64 q columns): 4 waves, each 16 tokens x all 64 head
// dimensions, so everything after the K loop is wave-private except the shared K / V tile.
//   K loop     the 64x64 LDS-DMA ring of gemm.hip (3 stages, counted vmcnt, one barrier per K tile), wave tile 16x64, plus ONE
//              extra MFMA per 32-k step for the head's gate logit: the gate row of W (row H*64 + h) is preloaded into LDS
//              (2 KB, by LDS-DMA ahead of the ring) and read as a B fragment whose 16 columns all hold that row -- the
//              accumulator then carries the logit of each of the wave's 16 tokens with the arithmetic of the plain GEMM.
//   epilogue   acc -> wave-private fp32 slab -> row pieces: x row scale, + bias, RoPE, bf16 (the expressions of
//              gemm_epilogue_lds, so q equals the unfused launch's bit for bit) -> wave-private swizzled q tile in LDS;
//              K / V rows of the head (requested when the kernel starts) -> shared swizzled tiles; then the single-tile body of
//              attn_mfma_kernel (swapped product: the query sits on the lane) and 8-byte stores of O * sigmoid(gate) / l.
// Results equal the two-launch path bit for bit (tests/test_kernels_gpu.py::test_qproj_xattn_equals_two_launches).
// S3 = the bf16x3 mode's arithmetic: hi | lo bf16 planes of A, W, q, K, V and P, three MFMAs per product (the split ring of gemm.hip,
// the body of attn_mfma_split_kernel), fp32 K / V rows and gate logit, output rows as hi | lo planes for the out-projection's split GEMM.
#include "gemm_common.h"

namespace {

struct QxParams {
  GemmParams g;     // a[0] / lda[0] / K, w / ldw (rows 0 .. H*64-1 = q, rows H*64 .. H*64+H-1 = gates), bias, M, rpb, rope*, rssq*
  const void* k;        // bf16 rows, or fp32 rows (S3)
  const void* v;
  bf16_t* out;          // bf16 rows; S3: hi | lo planes, the lo plane H*64 columns further
  int64_t krs, vrs, ors, kbs, vbs, obs;
  const int32_t* kv_len;
  const int32_t* q_len;
  float scale, clamp;
  int32_t H, Nk, tiles_per_seq, nseq;
};

template <int J, int LPW>
__device__ __forceinline__ void qx_ring_wait(int younger) {
  if constexpr (J == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    if (younger >= J) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(J * LPW) : "memory");
    else qx_ring_wait<J - 1, LPW>(younger);
  }
}

__device__ __forceinline__ void qx_split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bf16_t ha = (bf16_t)a[e], hb = (bf16_t)b[e];
    hi[e] = ha;
    hi[4 + e] = hb;
    lo[e] = (bf16_t)(a[e] - (float)ha);
    lo[4 + e] = (bf16_t)(b[e] - (float)hb);
  }
}

template <int CLAMP, bool S3>
__global__ __launch_bounds__(256) void qproj_xattn_kernel(QxParams P) {
  constexpr int BM = 64, BN = 64, NW = 4, NST = 3, TN = 4;
  constexpr int PLANE_BYTES = (BM + BN) * 128;
  constexpr int STAGE_BYTES = PLANE_BYTES * (S3 ? 2 : 1);     // S3: [A_hi | W_hi | A_lo | W_lo]
  constexpr int GA = BM / 8;
  constexpr int LPW = 4;                        // DMA instructions per wave per K tile and plane: groups wave, wave+4 (A), wave+8, wave+12 (W)
  constexpr int SLAB = 16 * (BN + 4) * 4;       // bytes of a wave's fp32 staging slab
  constexpr int TILE = 64 * 128;                // one bf16 64x64 tile
  constexpr int NPL = S3 ? 2 : 1;               // planes of q, K and V
  constexpr int Q_OFF = NW * SLAB, K_OFF = Q_OFF + NPL * TILE, V_OFF = K_OFF + NPL * TILE, G_OFF = V_OFF + NPL * TILE;
  static_assert(G_OFF + 64 * 4 <= NST * STAGE_BYTES, "post-loop regions alias the ring");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const GemmParams& p = P.g;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;

  // workgroup -> (row tile, head): XCD x (blockIdx & 7) takes a contiguous chunk of the row-tile-fastest order, i.e. ~2 heads'
  // W panels and every A panel (3.7 MB for one clip) per L2
  const int tiles_m = P.nseq * P.tiles_per_seq;
  int tm, h;
  {
    const int nwg = tiles_m * P.H;
    const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + ((int)blockIdx.x >> 3);
    h = L / tiles_m;
    tm = L - h * tiles_m;
  }
  const int b = tm / P.tiles_per_seq, t = tm - b * P.tiles_per_seq;
  const int rpb = p.rpb;
  const int m0 = b * rpb + t * BM;              // first row of the tile; rows of the tile beyond the sequence are clamped / dropped
  const int m_end = min((b + 1) * rpb, p.M);
  const int n0 = h * BN;

  // ---- K / V rows of this (sequence, head): requested first, parked in registers until the ring memory is free
  const int kchunk = tid & 7, krow = tid >> 3;
  struct Raw { f32x4 a, b; };
  bf16x8 kreg[S3 ? 1 : 2], vreg[S3 ? 1 : 2];
  Raw kraw[S3 ? 2 : 1], vraw[S3 ? 2 : 1];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int key = krow + 32 * i;
    key = key < P.Nk ? key : P.Nk - 1;
    if constexpr (S3) {
      const float* kpz = reinterpret_cast<const float*>(P.k) + b * P.kbs + h * 64 + (int64_t)key * P.krs + kchunk * 8;
      const float* vpz = reinterpret_cast<const float*>(P.v) + b * P.vbs + h * 64 + (int64_t)key * P.vrs + kchunk * 8;
      kraw[i].a = *reinterpret_cast<const f32x4*>(kpz);
      kraw[i].b = *reinterpret_cast<const f32x4*>(kpz + 4);
      vraw[i].a = *reinterpret_cast<const f32x4*>(vpz);
      vraw[i].b = *reinterpret_cast<const f32x4*>(vpz + 4);
    } else {
      kreg[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(P.k) + b * P.kbs + h * 64 + (int64_t)key * P.krs + kchunk * 8);
      vreg[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(P.v) + b * P.vbs + h * 64 + (int64_t)key * P.vrs + kchunk * 8);
    }
  }

  f32x4 acc[TN], accg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- LDS-DMA source geometry (gemm.hip): lane -> (row of the 8-row group, physical 16-B chunk), swizzle on the SOURCE chunk
  const int srow = lane >> 3;
  const int schunk = ((lane & 7) ^ srow) << 3;
  uint32_t goff[LPW];
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int g = wave + i * NW;
    if (g < GA) {
      int r = m0 + g * 8 + srow;
      r = r < m_end ? r : m_end - 1;
      goff[i] = (uint32_t)(((int64_t)r * p.lda[0] + schunk) * 2);
    } else {
      const int r = n0 + (g - GA) * 8 + srow;                   // always a q row: n0 + 63 < H * 64
      goff[i] = (uint32_t)(((int64_t)r * p.ldw + schunk) * 2);
    }
  }
  const char* a_run = reinterpret_cast<const char*>(p.a[0]);
  const char* w_run = reinterpret_cast<const char*>(p.w);
  const int64_t lo_bytes = (int64_t)p.K * 2;
  auto issue = [&](int stage) {
    char* st = smem_raw + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      const int g = wave + i * NW;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((g < GA ? a_run : w_run) + goff[i]),
                                       (__attribute__((address_space(3))) void*)(st + g * 1024), 16, 0, 0);
    }
    if constexpr (S3) {          // the lo planes: same rows, K elements further in the A row and in the W row
#pragma unroll
      for (int i = 0; i < LPW; ++i) {
        const int g = wave + i * NW;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((g < GA ? a_run : w_run) + lo_bytes + goff[i]),
                                         (__attribute__((address_space(3))) void*)(st + PLANE_BYTES + g * 1024), 16, 0, 0);
      }
    }
    a_run += 128;
    w_run += 128;
  };

  // ---- the head's gate row of W, linear in LDS behind the ring: K * 2 bytes = K / 512 wave-instructions.  Issued ahead of the
  // ring's DMAs, so every counted wait of the K loop retires them first.
  char* gate_lds = smem_raw + NST * STAGE_BYTES;
  {
    // S3: the row is [W_hi | W_lo], 2 K contiguous elements: the same linear copy, twice as long
    const char* gsrc = reinterpret_cast<const char*>(p.w) + ((int64_t)(P.H * 64 + h) * p.ldw) * 2 + lane * 16;
    const int ngi = (p.K >> 9) * NPL;
    for (int i = wave; i < ngi; i += NW)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + i * 1024),
                                       (__attribute__((address_space(3))) void*)(gate_lds + i * 1024), 16, 0, 0);
  }
  // folded RMSNorm (consumer): one thread per tile row
  float* rs_lds = reinterpret_cast<float*>(gate_lds + p.K * 2 * NPL);
  const bool scaled = p.rssq != nullptr;
  RowScaleLoad rsl;
  if (scaled && tid < BM) rowscale_load(p, min(m0 + tid, m_end - 1), rsl);
  const int nk = p.K / 64;
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nk) issue(s0);
  if (scaled && tid < BM) rs_lds[tid] = rowscale_finish(p, rsl);

  auto tile = [&](auto stage_c, int kt) {
    constexpr int STAGE = decltype(stage_c)::value;
    qx_ring_wait<NST - 2, LPW * NPL>(nk - 1 - kt);
    __builtin_amdgcn_s_barrier();
    const bf16_t* As = reinterpret_cast<const bf16_t*>(smem_raw + STAGE * STAGE_BYTES);
    const bf16_t* Ws = As + BM * 64;
    bf16x8 af[2], bf[2][TN], gf[2];
    const int row = wave * 16 + lr;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      af[kk] = *reinterpret_cast<const bf16x8*>(As + row * 64 + (((kk * 4 + lq) ^ (row & 7)) << 3));
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int wrow = j * 16 + lr;
        bf[kk][j] = *reinterpret_cast<const bf16x8*>(Ws + wrow * 64 + (((kk * 4 + lq) ^ (wrow & 7)) << 3));
      }
      gf[kk] = *reinterpret_cast<const bf16x8*>(gate_lds + (kt * 64 + kk * 32 + lq * 8) * 2);     // the same row for all 16 columns
    }
    if constexpr (S3) {
      const bf16_t* Al = As + PLANE_BYTES / 2;
      const bf16_t* Wl = Al + BM * 64;
      bf16x8 afl[2], bfl[2][TN], gfl[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        afl[kk] = *reinterpret_cast<const bf16x8*>(Al + row * 64 + (((kk * 4 + lq) ^ (row & 7)) << 3));
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int wrow = j * 16 + lr;
          bfl[kk][j] = *reinterpret_cast<const bf16x8*>(Wl + wrow * 64 + (((kk * 4 + lq) ^ (wrow & 7)) << 3));
        }
        gfl[kk] = *reinterpret_cast<const bf16x8*>(gate_lds + (p.K + kt * 64 + kk * 32 + lq * 8) * 2);
      }
      if (kt + NST - 1 < nk) issue((STAGE + NST - 1) % NST);
      // three products per fragment pair, small terms first (the order of gemm_bf16_dma_kernel<.., S3>)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afl[kk], bf[kk][j], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk], bfl[kk][j], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk], bf[kk][j], acc[j], 0, 0, 0);
        }
        accg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afl[kk], gf[kk], accg, 0, 0, 0);
        accg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk], gfl[kk], accg, 0, 0, 0);
        accg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk], gf[kk], accg, 0, 0, 0);
      }
    } else {
      if (kt + NST - 1 < nk) issue((STAGE + NST - 1) % NST);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk], bf[kk][j], acc[j], 0, 0, 0);
        accg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk], gf[kk], accg, 0, 0, 0);
      }
    }
  };
  for (int kt = 0; kt < nk; kt += NST) {
    tile(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) tile(std::integral_constant<int, 1>{}, kt + 1);
    if (kt + 2 < nk) tile(std::integral_constant<int, 2>{}, kt + 2);
  }
  __builtin_amdgcn_s_barrier();       // every wave is done reading the ring; all DMAs were retired by the last counted wait

  // ---- K / V tiles into the dead ring (row-major [key][64], 16-B chunk XOR-swizzled by key & 7, as in attn_mfma_kernel)
  bf16_t* ks = reinterpret_cast<bf16_t*>(smem_raw + K_OFF);      // S3: hi plane, the lo plane one tile further
  bf16_t* vt = reinterpret_cast<bf16_t*>(smem_raw + V_OFF);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = krow + 32 * i;
    const int off = row * 64 + ((kchunk ^ (row & 7)) << 3);
    if constexpr (S3) {
      bf16x8 hi, lo;
      qx_split8(kraw[i].a, kraw[i].b, hi, lo);
      *reinterpret_cast<bf16x8*>(ks + off) = hi;
      *reinterpret_cast<bf16x8*>(ks + 64 * 64 + off) = lo;
      qx_split8(vraw[i].a, vraw[i].b, hi, lo);
      *reinterpret_cast<bf16x8*>(vt + off) = hi;
      *reinterpret_cast<bf16x8*>(vt + 64 * 64 + off) = lo;
    } else {
      *reinterpret_cast<bf16x8*>(ks + off) = kreg[i];
      *reinterpret_cast<bf16x8*>(vt + off) = vreg[i];
    }
  }
  // ---- q rows of this wave: the STORE epilogue of gemm_epilogue_lds (row scale, bias, RoPE, bf16) into a swizzled LDS tile
  float* slab = reinterpret_cast<float*>(smem_raw + wave * SLAB);
  bf16_t* qt = reinterpret_cast<bf16_t*>(smem_raw + Q_OFF);
  float* gl = reinterpret_cast<float*>(smem_raw + G_OFF);
  constexpr int LD = BN + 4;
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) slab[(lq * 4 + jj) * LD + j * 16 + lr] = acc[j][jj];
  if (lr == 0) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) gl[wave * 16 + lq * 4 + jj] = accg[jj];
  }
  {
    const int c4 = (lane & 15) * 4, r0 = lane >> 4;
    const int n = n0 + c4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
    const bool rope = p.rope && n < p.rope_cols;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + q * 4;
      const int row = wave * 16 + r;
      int m = m0 + row;
      m = m < m_end ? m : m_end - 1;
      f32x4 v = *reinterpret_cast<const f32x4*>(slab + r * LD + c4);
      if (scaled) v *= rs_lds[row];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bv[e];
      if (rope) {
        const f32x4 cs = *reinterpret_cast<const f32x4*>(p.rope + ((int64_t)(p.rope_pos_off + m % rpb) * 32 + ((n & 63) >> 1)) * 2);
        const float a0 = v[0], b0 = v[1], a1 = v[2], b1 = v[3];
        v[0] = fmaf(a0, cs[0], -(b0 * cs[1]));
        v[1] = fmaf(b0, cs[0], a0 * cs[1]);
        v[2] = fmaf(a1, cs[2], -(b1 * cs[3]));
        v[3] = fmaf(b1, cs[2], a1 * cs[3]);
      }
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
      bf16_t* qd = qt + row * 64 + (((c4 >> 3) ^ (row & 7)) << 3) + (c4 & 7);
      *reinterpret_cast<bf16x4*>(qd) = o;
      if constexpr (S3) {        // the fp32 q of the unfused launch, split as attn_mfma_split_kernel splits it
        bf16x4 lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) lo[e] = (bf16_t)(v[e] - (float)o[e]);
        *reinterpret_cast<bf16x4*>(qd + 64 * 64) = lo;
      }
    }
  }
  __syncthreads();                     // K / V tiles, q tile and gate logits visible

  // ---- single-tile attention of the wave's 16 queries (attn_mfma_kernel<1, CLAMP> / attn_mfma_split_kernel<1, CLAMP> with one key tile)
  const int g = lq;
  const int qrow = wave * 16 + lr;                 // tile row of this lane's query
  const int qi = t * BM + qrow;                    // its index in the sequence
  bf16x8 qf[2], ql[S3 ? 2 : 1];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int off = qrow * 64 + (((kk * 4 + g) ^ (qrow & 7)) << 3);
    qf[kk] = *reinterpret_cast<const bf16x8*>(qt + off);
    if constexpr (S3) ql[kk] = *reinterpret_cast<const bf16x8*>(qt + 64 * 64 + off);
  }
  const int kvn = P.kv_len ? min(P.kv_len[b], P.Nk) : P.Nk;
  constexpr float LOG2E = 1.4426950408889634f;
  const float zc = P.clamp > 0.f ? 2.0f * LOG2E * P.scale / P.clamp : P.scale * LOG2E;
  const float c2 = P.clamp * LOG2E;
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
  if (kvn > 0) {
    f32x4 s[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      s[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = 16 * tt + lr;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int off = row * 64 + (((kk * 4 + g) ^ (row & 7)) << 3);
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + off);
        if constexpr (S3) {
          const bf16x8 kl = *reinterpret_cast<const bf16x8*>(ks + 64 * 64 + off);
          s[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qf[kk], s[tt], 0, 0, 0);
          s[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, ql[kk], s[tt], 0, 0, 0);
        }
        s[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[tt], 0, 0, 0);
      }
    }
    // soft clamp, key mask, softmax weights (fp32) in s
    if constexpr (CLAMP == 2) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float e = __builtin_amdgcn_exp2f(s[tt][j] * zc);
          s[tt][j] = __builtin_amdgcn_exp2f(fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f * c2, c2));
        }
      if (64 > kvn) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (16 * tt + 4 * g + j >= kvn) s[tt][j] = 0.f;
      }
    } else {
      float tmax = -INFINITY;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v;
          if constexpr (CLAMP == 1) {
            const float e = __builtin_amdgcn_exp2f(s[tt][j] * zc);
            v = fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f * c2, c2);
          } else {
            v = s[tt][j] * zc;
          }
          s[tt][j] = v;
        }
      if (64 > kvn) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (16 * tt + 4 * g + j >= kvn) s[tt][j] = -INFINITY;
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int j = 0; j < 4; ++j) tmax = fmaxf(tmax, s[tt][j]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[tt][j] = __builtin_amdgcn_exp2f(s[tt][j] - tmax);
    }
    // row sum and the P^T operand (S3: hi | lo planes of the fp32 weights)
    bf16x8 pf[2], pl[S3 ? 2 : 1];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pv = s[tt][j];
        l += pv;
        const bf16_t hv = (bf16_t)pv;
        pf[tt >> 1][(tt & 1) * 4 + j] = hv;
        if constexpr (S3) pl[tt >> 1][(tt & 1) * 4 + j] = (bf16_t)(pv - (float)hv);
      }
    // O^T += V^T P^T, V^T fragments by the transposing LDS read (see attn_mfma_kernel)
    const int vq = lr >> 2, vp = lr & 3;
    auto vt_read = [&](const bf16_t* vs, int key0, int dt) {
      const int key = key0 + 4 * g + vq;
      const int chunk = 2 * dt + (vp >> 1);
      const bf16_t* ad = vs + key * 64 + ((chunk ^ (key & 7)) << 3) + 4 * (vp & 1);
      return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)ad);
    };
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const bf16x4 lo = vt_read(vt, 32 * ks2, dt), hi = vt_read(vt, 32 * ks2 + 16, dt);
        bf16x8 vf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vf[j] = lo[j];
          vf[4 + j] = hi[j];
        }
        if constexpr (S3) {
          const bf16x4 lo2 = vt_read(vt + 64 * 64, 32 * ks2, dt), hi2 = vt_read(vt + 64 * 64, 32 * ks2 + 16, dt);
          bf16x8 vfl;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vfl[j] = lo2[j];
            vfl[4 + j] = hi2[j];
          }
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfl, pf[ks2], o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pl[ks2], o[dt], 0, 0, 0);
        }
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ks2], o[dt], 0, 0, 0);
      }
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (m0 + qrow >= m_end) return;
  // gate logit of this query: the GEMM's STORE epilogue (row scale, bias; bf16 mode: the bf16 rounding of the [q | gate] buffer), then sigmoid
  float gv = gl[qrow];
  if (scaled) gv *= rs_lds[qrow];
  if (p.bias) gv += p.bias[P.H * 64 + h];
  const float gt = S3 ? sigmoid_f(gv) : sigmoid_f((float)(bf16_t)gv);
  const int qn = P.q_len ? min(P.q_len[b], rpb) : rpb;
  const float f = (qi < qn && l > 0.f) ? gt / l : 0.f;
  bf16_t* op = P.out + b * P.obs + (int64_t)qi * P.ors + h * 64 + 4 * g;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    if constexpr (S3) {
      const f32x4 v = o[dt] * f;
      bf16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (bf16_t)v[e];
        lo[e] = (bf16_t)(v[e] - (float)hi[e]);
      }
      *reinterpret_cast<bf16x4*>(op + 16 * dt) = hi;
      *reinterpret_cast<bf16x4*>(op + P.H * 64 + 16 * dt) = lo;
    } else {
      bf16x4 ov;
#pragma unroll
      for (int j = 0; j < 4; ++j) ov[j] = (bf16_t)(o[dt][j] * f);
      *reinterpret_cast<bf16x4*>(op + 16 * dt) = ov;
    }
  }
}

template <int CLAMP, bool S3>
int launch_qx(const QxParams& P, hipStream_t s) {
  const size_t smem = 3 * (size_t)(64 + 64) * 128 * (S3 ? 2 : 1) + (size_t)P.g.K * 2 * (S3 ? 2 : 1) + 64 * 4;
  // the dynamic LDS size depends on K (the preloaded gate row) while the opt-in above 64 KB is set ONCE per (kernel, device): it is set
  // for the largest K the entry point accepts (4096), so a later launch with a larger K than the first one is not refused
  constexpr size_t smem_max = 3 * (size_t)(64 + 64) * 128 * (S3 ? 2 : 1) + (size_t)4096 * 2 * (S3 ? 2 : 1) + 64 * 4;
  static_assert(smem_max <= 160 * 1024, "qproj_xattn: LDS");
  auto kern = qproj_xattn_kernel<CLAMP, S3>;
  static std::atomic<uint64_t> lds_set{0};
  if (int rc = v2a_enable_lds(reinterpret_cast<const void*>(kern), smem_max, lds_set, "v2a_qproj_xattn")) return rc;
  hipLaunchKernelGGL(kern, dim3(P.nseq * P.tiles_per_seq * P.H), dim3(256), smem, s, P);
  return v2a_check_launch("v2a_qproj_xattn");
}

}  // namespace

extern "C" int v2a_qproj_xattn(const v2a_gemm_args* a, const v2a_attn_args* at, v2a_stream_t stream) {
  V2A_REQUIRE(a != nullptr && at != nullptr, "v2a_qproj_xattn: null args");
  const bool s3 = a->a_dtype == V2A_BF16_SPLIT;          // the bf16x3 mode: split operands, fp32 K / V, hi | lo output planes
  V2A_REQUIRE(a->nseg == 1 && a->a[0] && (a->a_dtype == V2A_BF16 || s3) && a->compute_dtype == V2A_BF16 && a->epilogue == V2A_EPI_STORE,
              "v2a_qproj_xattn: one bf16 (or split bf16) A segment, bf16 compute, STORE epilogue");
  V2A_REQUIRE(s3 ? (at->dtype == V2A_BF16_SPLIT && at->out_split) : (at->dtype == V2A_BF16 && !at->out_split),
              "v2a_qproj_xattn: bf16 operands go with dtype V2A_BF16, split operands with V2A_BF16_SPLIT and out_split");
  V2A_REQUIRE(at->H > 0 && at->B > 0 && at->Nq > 0 && at->Nk > 0 && at->Nk <= 64, "v2a_qproj_xattn: H=%d B=%d Nq=%d Nk=%d (Nk <= 64)", at->H, at->B, at->Nq, at->Nk);
  const int K = a->ka[0];
  V2A_REQUIRE(K > 0 && K % 512 == 0 && K <= 4096, "v2a_qproj_xattn: K=%d must be a multiple of 512, <= 4096", K);
  V2A_REQUIRE(a->N >= at->H * 65 && a->M == at->B * at->Nq && a->rows_per_batch == at->Nq,
              "v2a_qproj_xattn: W holds H*64 q rows + H gate rows (N=%d, H=%d); M=%d rows = B*Nq = %d*%d, rows_per_batch=%d", a->N, at->H, a->M,
              at->B, at->Nq, a->rows_per_batch);
  const int pl = s3 ? 2 : 1;
  V2A_REQUIRE(((uintptr_t)a->a[0] & 15) == 0 && a->lda[0] % 8 == 0 && a->lda[0] >= (int64_t)pl * K && a->w && ((uintptr_t)a->w & 15) == 0 &&
                  a->ldw % 8 == 0 && a->ldw >= (int64_t)pl * K,
              "v2a_qproj_xattn: A / W rows must be 16-byte aligned and hold K (split: 2 K) elements");
  V2A_REQUIRE(!a->bias || ((uintptr_t)a->bias & 15) == 0, "v2a_qproj_xattn: bias not 16-byte aligned");
  V2A_REQUIRE(!a->a_row_offset && !a->a_ktile_offset && !a->out_row_offset && !a->relu && !a->out_bf16 && !a->norm_gamma && !a->norm_ssq,
              "v2a_qproj_xattn: no offset tables, relu, shadow or norm producer");
  V2A_REQUIRE(at->k && at->v && at->out && (((uintptr_t)at->k | (uintptr_t)at->v) & 15) == 0 && ((uintptr_t)at->out & 7) == 0 &&
                  at->k_row_stride % 8 == 0 && at->v_row_stride % 8 == 0 && at->k_batch_stride % 8 == 0 && at->v_batch_stride % 8 == 0 &&
                  at->out_row_stride % 4 == 0 && at->out_batch_stride % 4 == 0 && (!s3 || at->out_row_stride >= 2 * (int64_t)at->H * 64),
              "v2a_qproj_xattn: k / v head slices must be 16-byte aligned, out rows 8-byte aligned (split: 2 * H * 64 bf16 per row)");
  QxParams P{};
  GemmParams& p = P.g;
  p.a[0] = a->a[0];
  p.lda[0] = a->lda[0];
  p.kend[0] = K;
  p.nseg = 1;
  p.w = a->w;
  p.ldw = a->ldw;
  p.bias = a->bias;
  p.M = a->M;
  p.N = at->H * 64;
  p.K = K;
  p.rpb = a->rows_per_batch;
  p.rope = a->rope_table;
  p.rope_cols = a->rope_cols;
  p.rope_pos_off = a->rope_pos_offset;
  if (a->rope_table)
    V2A_REQUIRE(a->rope_cols % 64 == 0 && a->rope_cols <= at->H * 64 && ((uintptr_t)a->rope_table & 15) == 0, "v2a_qproj_xattn: rope_cols %d", a->rope_cols);
  p.rssq = a->row_ssq;
  p.rssq_ld = a->ld_row_ssq;
  p.rssq_parts = a->row_ssq_parts;
  p.rnorm = sqrtf((float)a->row_norm_dim);
  if (a->row_ssq)
    V2A_REQUIRE(a->row_ssq_parts > 0 && a->row_ssq_parts <= 40 && a->row_norm_dim > 0 && a->ld_row_ssq >= (a->row_ssq_parts + 3) / 4 * 4 &&
                    a->ld_row_ssq % 4 == 0 && ((uintptr_t)a->row_ssq & 15) == 0,
                "v2a_qproj_xattn: row_ssq needs row_ssq_parts <= 40 (%d), row_norm_dim (%d) and rows of whole float4", a->row_ssq_parts, a->row_norm_dim);
  P.k = at->k;
  P.v = at->v;
  P.out = reinterpret_cast<bf16_t*>(at->out);
  P.krs = at->k_row_stride; P.vrs = at->v_row_stride; P.ors = at->out_row_stride;
  P.kbs = at->k_batch_stride; P.vbs = at->v_batch_stride; P.obs = at->out_batch_stride;
  P.kv_len = at->kv_len;
  P.q_len = at->q_len;
  P.scale = at->scale;
  P.clamp = at->softclamp;
  P.H = at->H;
  P.Nk = at->Nk;
  P.nseq = at->B;
  P.tiles_per_seq = (at->Nq + 63) / 64;
  hipStream_t s = (hipStream_t)stream;
  // the clamp mode v2a_attention would pick for these keys (attention.hip: bounded weights while clamp * log2 e + log2 Nk <= 90)
  int cl = 0;
  if (at->softclamp > 0.f) cl = at->softclamp * 1.4426950408889634f + log2f((float)(at->Nk > 1 ? at->Nk : 1)) <= 90.f ? 2 : 1;
  if (s3) {
    if (cl == 2) return launch_qx<2, true>(P, s);
    if (cl == 1) return launch_qx<1, true>(P, s);
    return launch_qx<0, true>(P, s);
  }
  if (cl == 2) return launch_qx<2, false>(P, s);
  if (cl == 1) return launch_qx<1, false>(P, s);
  return launch_qx<0, false>(P, s);
}
