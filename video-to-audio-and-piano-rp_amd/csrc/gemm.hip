// MFMA GEMM with fused epilogues for gfx950:  acc[m][n] = sum_k A[m][k] * W[n][k].
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights [N][K]), which
// is exactly the MFMA A/B fragment order, so no transposes are needed anywhere.
//   bf16 : v_mfma_f32_16x16x32_bf16, BK = 64, LDS rows of 128 B with the 16-B chunk index
//          XOR-swizzled by (row & 7) -> conflict-free ds_read_b128 fragment reads.
//   fp32 : v_mfma_f32_16x16x4_f32 (exact fp32 == fmaf chain), BK = 16, rows padded to 20
//          floats -- the parity mode, 1/16 of the bf16 rate by design.
// Block = 256 threads = 4 waves in a 2x2 grid; wave tile (BM/2)x(BN/2) in 16x16 MFMA tiles.
// Register-staged global->LDS with a 2-deep LDS ring and one barrier per K tile.
#include "v2a_common.h"

namespace {

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
  const float* resid;
  int64_t ldr;
  const float* gate;
  const int32_t* step;
  int64_t gss, gbs;
  int32_t rpb;
};

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

// ---- the kernel ---------------------------------------------------------------------------
template <typename T, bool A_F32, int EPI, typename OutT, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  constexpr int BK = TileCfg<T>::BK;
  constexpr int LR = TileCfg<T>::LDS_ROW;
  constexpr int WM = BM / 2, WN = BN / 2;   // wave tile
  constexpr int TM = WM / 16, TN = WN / 16; // MFMA tiles per wave
  static_assert(EPI != V2A_EPI_GEGLU || (TN % 2 == 0), "GEGLU needs value/gate tile pairs");

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  // ring: [2][ A tile (BM rows) | W tile (BN rows) ]
  constexpr int STAGE_ELEMS = (BM + BN) * LR;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lq = lane >> 4;

  // XCD-aware tile order: blocks that share an XCD (same blockIdx % 8) get consecutive
  // tiles of the same M-row band, so the A band and neighbouring W panels stay in that L2.
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  typename StageSel<T, A_F32, BM>::type sa;
  typename StageSel<T, false, BN>::type sw;

  const int nk = p.K / BK;

  auto seg_load = [&](int kt) {
    const int k0 = kt * BK;
    int s = 0, kbeg = 0;
    if (p.nseg > 1 && k0 >= p.kend[0]) { s = 1; kbeg = p.kend[0]; }
    if (p.nseg > 2 && k0 >= p.kend[1]) { s = 2; kbeg = p.kend[1]; }
    sa.load(p.a[s], p.lda[s], m0, p.M, k0 - kbeg, tid);
    sw.load(p.w, p.ldw, n0, p.N, k0, tid);
  };

  seg_load(0);
  sa.store(smem, tid);
  sw.store(smem + BM * LR, tid);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const T* As = smem + (kt & 1) * STAGE_ELEMS;
    const T* Ws = As + BM * LR;
    if (kt + 1 < nk) seg_load(kt + 1);

    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wm * WM + i * 16 + lr;
          af[i] = *reinterpret_cast<const bf16x8*>(As + row * 64 + (((kk * 4 + lq) ^ (row & 7)) << 3));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wn * WN + j * 16 + lr;
          bf[j] = *reinterpret_cast<const bf16x8*>(Ws + row * 64 + (((kk * 4 + lq) ^ (row & 7)) << 3));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = As[(wm * WM + i * 16 + lr) * 20 + kk * 4 + lq];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = Ws[(wn * WN + j * 16 + lr) * 20 + kk * 4 + lq];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }

    if (kt + 1 < nk) {
      T* An = smem + ((kt + 1) & 1) * STAGE_ELEMS;
      sa.store(An, tid);
      sw.store(An + BM * LR, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane & 15, row = (lane >> 4) * 4 + j -------------------
  OutT* out = reinterpret_cast<OutT*>(p.out);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int m = m0 + wm * WM + i * 16 + lq * 4 + jj;
      if (m >= p.M) continue;
      const float* gvec = nullptr;
      if constexpr (EPI == V2A_EPI_GATE_RESID) gvec = step_vec(p.gate, p.step, p.gss, p.gbs, m / p.rpb);
      if constexpr (EPI == V2A_EPI_GEGLU) {
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
          const int n = n0 + wn * WN + j * 16 + lr;  // packed row index of the value
          if (n >= p.N) continue;
          float v = acc[i][j][jj], g = acc[i][j + 1][jj];
          if (p.bias) { v += p.bias[n]; g += p.bias[n + 16]; }
          const int oc = ((n0 + wn * WN) >> 1) + (j >> 1) * 16 + lr;
          out[(int64_t)m * p.ldo + oc] = from_f32<OutT>(v * gelu_erf_f(g));
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
          out[(int64_t)m * p.ldo + n] = from_f32<OutT>(v);
        }
      }
    }
  }
}

template <typename T, bool A_F32, int EPI, typename OutT, int BM, int BN>
int launch(const GemmParams& p, hipStream_t s) {
  constexpr int LR = TileCfg<T>::LDS_ROW;
  constexpr size_t smem = 2 * (size_t)(BM + BN) * LR * sizeof(T);
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  auto kern = gemm_kernel<T, A_F32, EPI, OutT, BM, BN>;
  static bool attr_set = false;  // > 64 KB dynamic LDS is opt-in
  if (!attr_set && smem > 48 * 1024) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), smem, s, p);
  return v2a_check_launch("v2a_gemm");
}

template <typename T, bool A_F32, int BM, int BN>
int dispatch_epi(const v2a_gemm_args* a, const GemmParams& p, hipStream_t s) {
  const bool out_f32 = a->out_dtype == V2A_F32;
  switch (a->epilogue) {
    case V2A_EPI_STORE:
      if (out_f32) return launch<T, A_F32, V2A_EPI_STORE, float, BM, BN>(p, s);
      if constexpr (sizeof(T) == 2) return launch<T, A_F32, V2A_EPI_STORE, bf16_t, BM, BN>(p, s);
      break;
    case V2A_EPI_SIGMOID:
      if (out_f32) return launch<T, A_F32, V2A_EPI_SIGMOID, float, BM, BN>(p, s);
      break;
    case V2A_EPI_GEGLU:
      if (out_f32) {
        if constexpr (sizeof(T) == 4) return launch<T, A_F32, V2A_EPI_GEGLU, float, BM, BN>(p, s);
      } else {
        if constexpr (sizeof(T) == 2) return launch<T, A_F32, V2A_EPI_GEGLU, bf16_t, BM, BN>(p, s);
      }
      break;
    case V2A_EPI_RESID:
      if (out_f32) return launch<T, A_F32, V2A_EPI_RESID, float, BM, BN>(p, s);
      break;
    case V2A_EPI_GATE_RESID:
      if (out_f32) return launch<T, A_F32, V2A_EPI_GATE_RESID, float, BM, BN>(p, s);
      break;
  }
  return v2a_fail(V2A_ERR_ARG, "v2a_gemm: unsupported epilogue %d / out_dtype %d for compute dtype %d", a->epilogue,
                  a->out_dtype, a->compute_dtype);
}

}  // namespace

extern "C" int v2a_gemm(const v2a_gemm_args* a, v2a_stream_t stream) {
  V2A_REQUIRE(a != nullptr, "v2a_gemm: null args");
  V2A_REQUIRE(a->nseg >= 1 && a->nseg <= 3, "v2a_gemm: nseg=%d", a->nseg);
  V2A_REQUIRE(a->M > 0 && a->N > 0, "v2a_gemm: M=%d N=%d", a->M, a->N);
  V2A_REQUIRE(a->compute_dtype == V2A_F32 || a->compute_dtype == V2A_BF16, "v2a_gemm: compute dtype %d", a->compute_dtype);
  V2A_REQUIRE(a->a_dtype == a->compute_dtype || a->a_dtype == V2A_F32, "v2a_gemm: A dtype %d with compute dtype %d",
              a->a_dtype, a->compute_dtype);
  const int bk = a->compute_dtype == V2A_BF16 ? 64 : 16;
  const int a_vec = a->a_dtype == V2A_BF16 ? 8 : 4;  // elements per 16-byte load
  GemmParams p{};
  int K = 0;
  for (int s = 0; s < a->nseg; ++s) {
    V2A_REQUIRE(a->a[s] != nullptr, "v2a_gemm: segment %d null", s);
    V2A_REQUIRE(a->ka[s] > 0 && a->ka[s] % bk == 0, "v2a_gemm: segment %d K=%d not a multiple of %d", s, a->ka[s], bk);
    V2A_REQUIRE(a->lda[s] % a_vec == 0 && ((uintptr_t)a->a[s] & 15) == 0, "v2a_gemm: segment %d not 16-byte aligned", s);
    p.a[s] = a->a[s];
    p.lda[s] = a->lda[s];
    K += a->ka[s];
    p.kend[s] = K;
  }
  p.nseg = a->nseg;
  V2A_REQUIRE(a->w != nullptr && ((uintptr_t)a->w & 15) == 0 && a->ldw % (a->compute_dtype == V2A_BF16 ? 8 : 4) == 0 && a->ldw >= K,
              "v2a_gemm: W pointer/ldw (%lld) misaligned or < K=%d", (long long)a->ldw, K);
  V2A_REQUIRE(a->out != nullptr, "v2a_gemm: out null");
  p.w = a->w;
  p.ldw = a->ldw;
  p.bias = a->bias;
  p.M = a->M;
  p.N = a->N;
  p.K = K;
  p.out = a->out;
  p.ldo = a->ldo;
  p.resid = a->resid;
  p.ldr = a->ldr;
  p.gate = a->gate;
  p.step = a->step;
  p.gss = a->gate_step_stride;
  p.gbs = a->gate_batch_stride;
  p.rpb = a->rows_per_batch > 0 ? a->rows_per_batch : a->M;
  if (a->epilogue == V2A_EPI_RESID || a->epilogue == V2A_EPI_GATE_RESID)
    V2A_REQUIRE(a->resid != nullptr, "v2a_gemm: epilogue %d needs resid", a->epilogue);
  if (a->epilogue == V2A_EPI_GATE_RESID) V2A_REQUIRE(a->gate != nullptr, "v2a_gemm: GATE_RESID needs gate");
  if (a->epilogue == V2A_EPI_GEGLU) V2A_REQUIRE(a->N % 32 == 0, "v2a_gemm: GEGLU needs N %% 32 == 0 (N=%d)", a->N);
  hipStream_t s = (hipStream_t)stream;
  if (a->compute_dtype == V2A_F32) return dispatch_epi<float, false, 128, 128>(a, p, s);
  if (a->a_dtype == V2A_F32) return dispatch_epi<bf16_t, true, 128, 128>(a, p, s);
  return dispatch_epi<bf16_t, false, 128, 128>(a, p, s);
}
