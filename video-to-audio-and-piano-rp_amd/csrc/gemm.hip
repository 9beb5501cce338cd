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
#include "gemm_common.h"

namespace {


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

  gemm_epilogue<EPI, OutT, TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, lr, lq);
}

template <typename T, bool A_F32, int EPI, typename OutT, int BM, int BN>
int launch(const GemmParams& p, hipStream_t s) {
  constexpr int LR = TileCfg<T>::LDS_ROW;
  constexpr size_t smem = 2 * (size_t)(BM + BN) * LR * sizeof(T);
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  auto kern = gemm_kernel<T, A_F32, EPI, OutT, BM, BN>;
  if constexpr (smem > 48 * 1024) {
    static std::atomic<uint64_t> lds_set{0};
    if (int rc = v2a_enable_lds(reinterpret_cast<const void*>(kern), smem, lds_set, "v2a_gemm")) return rc;
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


// ---- v2: bf16 x bf16, LDS-DMA staging, 3-deep ring -------------------------------------------
// Both tiles arrive by global_load_lds_dwordx4 (1 KiB = 8 rows x 128 B per wave-instruction, LDS
// image linear per wave, XOR swizzle applied to the per-lane SOURCE chunk: rule 21 of the CDNA guide),
// two K tiles stay in flight across the single raw s_barrier of each iteration (counted vmcnt),
// and no VGPRs or ds_writes are spent on staging.
// Wait until at most min(J, younger) of the youngest K tiles' DMAs (LPW instructions per wave each) are still outstanding:
// the counted s_waitcnt needs an immediate, so the tail of the K loop walks down a chain of them.
template <int J, int LPW>
__device__ __forceinline__ void ring_wait(int younger) {
  if constexpr (J == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    static_assert(J * LPW <= 63, "vmcnt is a 6-bit field");
    if (younger >= J) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(J * LPW) : "memory");
    else ring_wait<J - 1, LPW>(younger);
  }
}

// S3 (the "bf16x3" mode natively): both operands arrive as hi | lo bf16 planes (V2A_BF16_SPLIT rows of A, W rows [W_hi | W_lo]),
// a ring stage holds the four plane tiles of a K step and every fragment pair feeds three MFMAs,
//   acc += A_lo W_hi + A_hi W_lo + A_hi W_hi   (fp32-grade product, relative error ~2^-16),
// so a K step moves 2x the bytes of a bf16 step for 3x its flops -- where the three-segment form on the plain kernel
// ([A_hi | A_hi | A_lo] x [W_hi | W_lo | W_hi]) moved 3x the bytes and needed one launch per logical K segment.
// (Round 4 measured a software-pipelined form of the K loop -- the fragment reads of tile k + 1 issued behind the barrier of step k into a
// second register set, landing under the MFMAs of tile k: 1-5 % SLOWER at equal occupancy, 16-27 % where the extra registers cost a
// resident workgroup, profiles/r04_ring_pipe_probe.txt.  The exposed part of a K step is not the LDS read latency; the form is not kept.)
// BK (round 5): K extent of a ring stage, 64 or 32.  With 32 the LDS rows are 64 B -- a 1 KB DMA instruction then covers 16 rows, the
// 16-B chunk index of a row is XOR-ed with 3 for rows 8..15 of each 16-row group (conflict-free for the lane groups of ds_read_b128:
// every 16 lanes of a group then cover all 16 slots of a 256-B bank row) -- and a stage holds half the bytes: the split-operand 128x256
// tile (hi + lo planes) fits three stages in 144 KB.
template <int EPI, typename OutT, int BM, int BN, int WGM, int WGN, int NST = 3, bool S3 = false, int BK = 64>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_bf16_dma_kernel(const GemmParams p) {
  static_assert(BK == 64 || BK == 32, "K extent of a stage");
  constexpr int NW = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int ROWB = BK * 2;                            // bytes of an LDS row
  constexpr int RPG = 1024 / ROWB;                        // rows per DMA instruction (1 KB per wave-instruction)
  constexpr int CPR = BK / 8;                             // 16-B chunks per row
  constexpr int KK = BK / 32;                             // MFMA K steps per stage
  constexpr int PLANE_BYTES = (BM + BN) * ROWB;           // one A tile + one W tile of a K step
  constexpr int STAGE_BYTES = PLANE_BYTES * (S3 ? 2 : 1); // S3: [A_hi | W_hi | A_lo | W_lo]
  constexpr int GA = BM / RPG, GW = BN / RPG;   // DMA groups (RPG rows each) of the A and W tiles
  constexpr int LPW = (GA + GW) / NW;           // DMA instructions per wave per K tile
  static_assert((GA + GW) % NW == 0, "DMA groups must divide evenly over the waves");
  // chunk swizzle of a row inside its DMA group (applied to the DMA source address and to the fragment reads)
  auto swz = [](int row) { return BK == 64 ? (row & 7) : (((row >> 3) & 1) * 3); };
  static_assert(EPI != V2A_EPI_GEGLU || (TN % 2 == 0), "GEGLU needs value/gate tile pairs");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int lr = lane & 15, lq = lane >> 4;

  int tm, tn;
  tile_of_block(p, blockIdx.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane source geometry: lane -> (row in 8-row group = lane >> 3, physical chunk = lane & 7);
  // DMA group g (0 .. GA+GW-1) is handled by wave g % NW; groups < GA are A rows, the rest W rows.
  // Each slot keeps ONE 32-bit per-lane byte offset for the whole K loop; the K advance and the
  // operand base are wave-uniform (SGPR), so an issue costs no vector address arithmetic.
  const int srow = lane / CPR;
  const int schunk = ((lane % CPR) ^ swz(srow)) << 3;  // logical chunk (elements) fetched into physical slot lane % CPR
  int grow[LPW];
  uint32_t goff[LPW];
  auto set_offsets = [&](int64_t lda) {
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      const int g = wave + i * NW;
      const int64_t rbase = g < GA ? (p.a_rowoff ? (int64_t)p.a_rowoff[grow[i]] : (int64_t)grow[i] * lda) : (int64_t)grow[i] * p.ldw;
      goff[i] = (uint32_t)((rbase + schunk) * 2);
    }
  };
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int g = wave + i * NW;
    if (g < GA) {
      const int r = m0 + g * RPG + srow;
      grow[i] = r < p.M ? r : p.M - 1;
    } else {
      const int r = n0 + (g - GA) * RPG + srow;
      grow[i] = r < p.N ? r : p.N - 1;
    }
  }
  set_offsets(p.lda[0]);
  // K-tile offset table read through the constant address space: a scalar load (lgkmcnt).  As a plain global load inside
  // the K loop it became a VECTOR load (the LDS-DMA stores "clobber" memory for the compiler) followed by s_waitcnt vmcnt(0)
  // -- on the dense path too, at the join of the two branches -- which drained the staged tiles in flight every K tile.
  typedef const __attribute__((address_space(4))) int32_t* const_i32_ptr;
  const const_i32_ptr koff_tab = (const_i32_ptr)p.a_koff;

  // Operand stream state, all wave-uniform: K tiles are issued strictly in order, so the A / W tile bases are running pointers
  // stepped by 128 B per K tile and the (at most two) segment switches are counted down.  The first version recomputed the
  // segment of every K tile from the kernel arguments -- p.a[sgi] / p.lda[sgi] with a run-time sgi are two dependent scalar
  // loads from the argument segment, plus ~60 scalar and vector-select instructions -- between the barrier and the first DMA of
  // EVERY K step: ~500 cycles per step for 128 cycles of MFMA on a 64x64 tile (ISA listing and K-loop probe in DESIGN.md section 4).
  const char* a_run = reinterpret_cast<const char*>(p.a[0]);
  const char* w_run = reinterpret_cast<const char*>(p.w);
  const char* const a_first = a_run;
  const int nseg = p.nseg;
  int seg = 0;
  int seg_left = (nseg > 1 ? p.kend[0] : p.K) / BK;       // K tiles left in the current segment
  // S3: byte offset of the lo plane inside an A row (= the segment's K extent) and inside a W row (= K)
  int64_t a_lo_bytes = S3 ? p.alo[0] * 2 : 0;
  const int64_t w_lo_bytes = S3 ? (int64_t)p.K * 2 : 0;
  int kt_next = 0;                                        // index of the next K tile to issue (offset-table path only)
  auto issue = [&](int stage) {
    if (seg_left == 0) {                                  // at most twice per kernel: the next A segment has its own base and row stride
      ++seg;
      a_run = reinterpret_cast<const char*>(seg == 1 ? p.a[1] : p.a[2]);
      seg_left = (seg == 1 ? (nseg > 2 ? p.kend[1] : p.K) - p.kend[0] : p.K - p.kend[1]) / BK;
      if constexpr (S3) a_lo_bytes = (seg == 1 ? p.alo[1] : p.alo[2]) * 2;
      set_offsets(seg == 1 ? p.lda[1] : p.lda[2]);
    }
    const char* ab = a_run;
    if (koff_tab) ab = a_first + (int64_t)koff_tab[kt_next] * 2;     // implicit-GEMM convolution: wave-uniform scalar load
    const char* wb = w_run;
    char* st = smem_raw + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      const int g = wave + i * NW;
      if (g < GA)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ab + goff[i]),
                                         (__attribute__((address_space(3))) void*)(st + g * 1024), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + goff[i]),
                                         (__attribute__((address_space(3))) void*)(st + g * 1024), 16, 0, 0);
    }
    if constexpr (S3) {          // the lo planes: same rows, the segment's / the weight's plane offset further
#pragma unroll
      for (int i = 0; i < LPW; ++i) {
        const int g = wave + i * NW;
        if (g < GA)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ab + a_lo_bytes + goff[i]),
                                           (__attribute__((address_space(3))) void*)(st + PLANE_BYTES + g * 1024), 16, 0, 0);
        else
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + w_lo_bytes + goff[i]),
                                           (__attribute__((address_space(3))) void*)(st + PLANE_BYTES + g * 1024), 16, 0, 0);
      }
    }
    a_run += ROWB;
    w_run += ROWB;
    --seg_left;
    ++kt_next;
  };

  // epilogue operands are prefetched unless registers are short: the 256x256 tile (128 accumulators per lane) and the
  // 8-wave 128x256 tile with BOTH residual and gate rows (64 + 64 registers on top of two fragment sets)
  // (round 4: the STORE epilogue's RoPE prefetch on that tile -- 64 more registers -- spilled 78 VGPRs to scratch, found by
  // tests/test_isa_guard.py: off there too)
  constexpr bool PF = BM * BN <= 128 * 256 && !((EPI == V2A_EPI_GATE_RESID || EPI == V2A_EPI_STORE) && BM * BN == 128 * 256);
  EpiPrefetch<EPI, TM, WN, PF> pf;
  if (p.vec_epi) pf.load(p, m0 + wm * WM, n0 + wn * WN, lane);
  // folded RMSNorm (consumer): the row's partial sums are requested ahead of the first operand DMA ...
  static_assert(BM <= 64 * NW, "one thread per tile row");
  float* rs_lds = reinterpret_cast<float*>(smem_raw + NST * STAGE_BYTES);
  const bool scaled = p.rssq != nullptr && p.vec_epi;
  RowScaleLoad rsl;
  if (scaled && tid < BM) rowscale_load(p, m0 + tid, rsl);
  const int nk = p.K / BK;
  // ring of NST stages, NST - 1 K tiles in flight while one is computed (3 by default; 2: the 256x256 tile, whose 64 KB
  // stages leave room for only two; 6: the small tiles of a one-clip launch, which has at most one workgroup per CU and is
  // bound by DMA latency x bytes in flight -- a K step of the 3-deep 64x64 ring took 1020 cycles for 128 cycles of MFMA)
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nk) issue(s0);
  // ... and turned into the row's scale behind them (visible to every wave after the K loop's barriers)
  if (scaled && tid < BM) rs_lds[tid] = rowscale_finish(p, rsl);
  // one K tile; STAGE is a compile-time ring position so every LDS address is base + immediate
  auto tile = [&](auto stage_c, int kt) {
    constexpr int STAGE = decltype(stage_c)::value;
    // tile kt has landed for this wave once only the younger tile's DMAs remain outstanding
#ifdef V2A_GEMM_PROBE   // K-loop attribution (scripts/probes/kloop_probe.py): bit 0 = no fragment reads / MFMAs, bit 1 = no DMA and no wait in the loop
    if (!(p.dbg & 2))
#endif
    ring_wait<NST - 2 < 0 ? 0 : NST - 2, LPW * (S3 ? 2 : 1)>(nk - 1 - kt);
    __builtin_amdgcn_s_barrier();   // ... and for every wave; also: everyone is done reading stage (kt-1) % NST
#ifdef V2A_GEMM_PROBE
    if (p.dbg & 1) {
      if (kt + NST - 1 < nk) issue((STAGE + NST - 1) % NST);
      return;
    }
#endif
    const bf16_t* As = reinterpret_cast<const bf16_t*>(smem_raw + STAGE * STAGE_BYTES);
    const bf16_t* Ws = As + BM * BK;
    // all 32-wide K steps of the tile are requested before the first MFMA: the later ones' LDS latency runs under the
    // first one's MFMA cluster instead of being exposed (the compiler otherwise emits read / wait / MFMA per step)
    bf16x8 af[KK][TM], bf[KK][TN];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WM + i * 16 + lr;
        af[kk][i] = *reinterpret_cast<const bf16x8*>(As + row * BK + (((kk * 4 + lq) ^ swz(row)) << 3));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WN + j * 16 + lr;
        bf[kk][j] = *reinterpret_cast<const bf16x8*>(Ws + row * BK + (((kk * 4 + lq) ^ swz(row)) << 3));
      }
    }
    if constexpr (S3) {
      // lo planes of the same rows; three products per fragment pair, small terms first
      const bf16_t* Al = As + PLANE_BYTES / 2;
      const bf16_t* Wl = Al + BM * BK;
      bf16x8 afl[KK][TM], bfl[KK][TN];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wm * WM + i * 16 + lr;
          afl[kk][i] = *reinterpret_cast<const bf16x8*>(Al + row * BK + (((kk * 4 + lq) ^ swz(row)) << 3));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wn * WN + j * 16 + lr;
          bfl[kk][j] = *reinterpret_cast<const bf16x8*>(Wl + row * BK + (((kk * 4 + lq) ^ swz(row)) << 3));
        }
      }
#ifdef V2A_GEMM_PROBE
      if (!(p.dbg & 2))
#endif
      if (kt + NST - 1 < nk) issue((STAGE + NST - 1) % NST);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
#ifdef V2A_GEMM_PROBE     // error attribution (tests/precision_attribution.py): bit 5 drops the two cross products = plain bf16 arithmetic
            if (!(p.dbg & 32)) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afl[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][i], bfl[kk][j], acc[i][j], 0, 0, 0);
            }
#else
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afl[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][i], bfl[kk][j], acc[i][j], 0, 0, 0);
#endif
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
          }
    } else {
    // the next tile's DMA is issued behind the first half's fragment reads: their LDS latency runs under the DMA issue
    // (~4 x LPW scalar instructions and LPW address translations) instead of after it
#ifdef V2A_GEMM_PROBE
    if (!(p.dbg & 2))
#endif
    if (kt + NST - 1 < nk) issue((STAGE + NST - 1) % NST);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
    // pin that order (the scheduler otherwise sinks the second half's reads back below the first MFMA cluster):
    // [first half's reads] [the DMA issue] then {a few MFMAs, one read of the second half} ... then the remaining MFMAs
    if constexpr (BK == 64) {
      constexpr int NL = TM + TN, NM = TM * TN, PER = NM / NL > 0 ? NM / NL : 1;
      __builtin_amdgcn_sched_group_barrier(0x100, NL, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, LPW, 0);
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NM - PER * NL, 0);
    }
    }
  };
  static_assert(NST >= 2 && NST <= 6, "ring depth");
  for (int kt = 0; kt < nk; kt += NST) {
    tile(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) tile(std::integral_constant<int, 1>{}, kt + 1);
    if constexpr (NST > 2) { if (kt + 2 < nk) tile(std::integral_constant<int, 2>{}, kt + 2); }
    if constexpr (NST > 3) { if (kt + 3 < nk) tile(std::integral_constant<int, 3>{}, kt + 3); }
    if constexpr (NST > 4) { if (kt + 4 < nk) tile(std::integral_constant<int, 4>{}, kt + 4); }
    if constexpr (NST > 5) { if (kt + 5 < nk) tile(std::integral_constant<int, 5>{}, kt + 5); }
  }
  if (p.vec_epi) {
    __builtin_amdgcn_s_barrier();   // every wave is done reading the K-loop stages; all DMAs were retired above
    static_assert(NW * 16 * (WN + 4) * 4 <= NST * STAGE_BYTES, "epilogue slabs must fit in the ring memory");
    float* tile = reinterpret_cast<float*>(smem_raw) + wave * (16 * (WN + 4));
    gemm_epilogue_lds<EPI, OutT, TM, TN, WM, WN, PF>(p, acc, tile, m0 + wm * WM, n0 + wn * WN, lane, pf, scaled ? rs_lds + wm * WM : nullptr);
  } else {
    gemm_epilogue<EPI, OutT, TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, lr, lq);
  }
}

template <int EPI, typename OutT, int BM, int BN, int WGM, int WGN, int NST, bool S3 = false, int BK = 64>
int launch_dma(const GemmParams& p_in, hipStream_t s) {
  GemmParams p = p_in;
  v2a_detail::fill_tile_map(p, BM, BN);
  V2A_REQUIRE((int64_t)p.tiles_m * p.tiles_n < (1 << 24), "v2a_gemm: %d x %d tiles exceed the tile map", p.tiles_m, p.tiles_n);
  constexpr size_t smem = NST * (size_t)(BM + BN) * (BK * 2) * (S3 ? 2 : 1) + BM * 4 + 16;    // the ring + one row scale per tile row (folded RMSNorm)
  static_assert(smem <= 160 * 1024, "ring does not fit the LDS");
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  // W loads keep the default cache policy: non-temporal (aux = 2) measured 9 % slower end to end here, the W
  // panel being re-read from L2 by the 13-25 M-band workgroups of its XCD (profiles/ notes in DESIGN.md)
  auto kern = gemm_bf16_dma_kernel<EPI, OutT, BM, BN, WGM, WGN, NST, S3, BK>;
  if constexpr (BK != 64) V2A_REQUIRE(!p.a_koff, "v2a_gemm(dma): K-tile offset tables are laid out for 64-wide K tiles");
  static std::atomic<uint64_t> lds_set{0};
  if (int rc = v2a_enable_lds(reinterpret_cast<const void*>(kern), smem, lds_set, "v2a_gemm(dma)")) return rc;
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WGM * WGN), smem, s, p);
  return v2a_check_launch("v2a_gemm(dma)");
}

template <int BM, int BN, int WGM, int WGN, int NST = 3>
int dispatch_dma(const v2a_gemm_args* a, const GemmParams& p, hipStream_t s) {
  const bool out_f32 = a->out_dtype == V2A_F32;
  switch (a->epilogue) {
    case V2A_EPI_STORE:
      return out_f32 ? launch_dma<V2A_EPI_STORE, float, BM, BN, WGM, WGN, NST>(p, s) : launch_dma<V2A_EPI_STORE, bf16_t, BM, BN, WGM, WGN, NST>(p, s);
    case V2A_EPI_GEGLU:   // fp32 output: the hidden activation of the bf16x3 mode, split into hi / lo planes afterwards
      if constexpr ((BN / WGN / 16) % 2 == 0)
        return out_f32 ? launch_dma<V2A_EPI_GEGLU, float, BM, BN, WGM, WGN, NST>(p, s) : launch_dma<V2A_EPI_GEGLU, bf16_t, BM, BN, WGM, WGN, NST>(p, s);
      break;
    case V2A_EPI_RESID:
      if (out_f32) return launch_dma<V2A_EPI_RESID, float, BM, BN, WGM, WGN, NST>(p, s);
      break;
    case V2A_EPI_GATE_RESID:
      if (out_f32) return launch_dma<V2A_EPI_GATE_RESID, float, BM, BN, WGM, WGN, NST>(p, s);
      break;
  }
  return v2a_fail(V2A_ERR_ARG, "v2a_gemm(dma): unsupported epilogue %d / out_dtype %d", a->epilogue, a->out_dtype);
}

// split-bf16 operands (bf16x3 mode): the epilogues that mode uses
template <int BM, int BN, int WGM, int WGN, int NST, int BK = 64>
int dispatch_s3(const v2a_gemm_args* a, const GemmParams& p, hipStream_t s) {
  const bool out_f32 = a->out_dtype == V2A_F32;
  switch (a->epilogue) {
    case V2A_EPI_STORE:
      if (out_f32) return launch_dma<V2A_EPI_STORE, float, BM, BN, WGM, WGN, NST, true, BK>(p, s);
      break;
    case V2A_EPI_GEGLU:
      if constexpr ((BN / WGN / 16) % 2 == 0) {
        if (a->out_dtype == V2A_BF16_SPLIT) return launch_dma<V2A_EPI_GEGLU, bf16_t, BM, BN, WGM, WGN, NST, true, BK>(p, s);
      }
      break;
    case V2A_EPI_RESID:
      if (out_f32) return launch_dma<V2A_EPI_RESID, float, BM, BN, WGM, WGN, NST, true, BK>(p, s);
      break;
    case V2A_EPI_GATE_RESID:
      if (out_f32) return launch_dma<V2A_EPI_GATE_RESID, float, BM, BN, WGM, WGN, NST, true, BK>(p, s);
      break;
  }
  return v2a_fail(V2A_ERR_ARG, "v2a_gemm(split bf16): unsupported epilogue %d / out_dtype %d for this tile shape", a->epilogue, a->out_dtype);
}

// split operands on the 8-phase kernel: three passes over the logical K -- A_hi x W_hi, A_hi x W_lo, A_lo x W_hi -- each over all logical
// segments (GemmParams::s3_kl; the kernel derives plane and segment of every K tile)
void split_as_three_passes(GemmParams& q) {
  q.s3_kl = q.K;
  q.K = 3 * q.K;
}

}  // namespace

// defaults: phase-interleaved 256x256 kernel for wide outputs once a launch has >= 400 tiles, i.e. from two clips per GPU on
// (A/B on MI355X: +1..2 % at 8 clips per GPU end to end).  At one clip its 224-280 workgroups of 128 KB LDS take every CU for
// 35-70 us: alone it is the fastest choice for the feed-forward GEMMs (774 vs 644 TF/s), beside the other two streams of the
// sampler it costs 1.3 % end to end, and with fewer tiles the 128x256 ring kernel fills the chip better anyway.
static constexpr v2a_detail::GemmTuning kDefaultTuning = {-1, 0, 1, 400, 1, 0};
v2a_detail::GemmTuning v2a_detail::g_gemm_tuning = kDefaultTuning;
int v2a_detail::g_dwconv_rows_per_wave = 4;
int v2a_detail::g_attn_one_group_from = 1536;
int v2a_detail::g_probe_dbg = 0;       // v2a_tuning.reserved[0]: read by probe builds only
int v2a_detail::g_dwconv_stream = 1;   // streaming depthwise conv for chip-filling launches (dwconv_rows_per_wave = -1 switches it off: A/B)

extern "C" int v2a_gemm_args_size(void) { return (int)sizeof(v2a_gemm_args); }

extern "C" int v2a_set_tuning(const v2a_tuning* t) {
  if (!t) {
    v2a_detail::g_gemm_tuning = kDefaultTuning;
    v2a_detail::g_dwconv_rows_per_wave = 4;
    v2a_detail::g_dwconv_stream = 1;
    v2a_detail::g_attn_one_group_from = 1536;
    v2a_detail::g_probe_dbg = 0;
    return V2A_OK;
  }
  // every field is checked before any is assigned: a rejected call leaves the previous tuning whole
  V2A_REQUIRE(t->dwconv_rows_per_wave == 0 || t->dwconv_rows_per_wave == 4 || t->dwconv_rows_per_wave == 6 || t->dwconv_rows_per_wave == -1,
              "v2a_set_tuning: dwconv_rows_per_wave %d", t->dwconv_rows_per_wave);
  V2A_REQUIRE(t->gemm_force_tile >= -1 && t->gemm_force_tile <= 8 && t->gemm_force_tile != 4, "v2a_set_tuning: gemm_force_tile %d", t->gemm_force_tile);
  V2A_REQUIRE(t->gemm_8phase >= 0 && t->gemm_8phase <= 2, "v2a_set_tuning: gemm_8phase %d", t->gemm_8phase);
  V2A_REQUIRE(t->gemm_8phase_min_tiles >= 0 && t->attn_one_group_from >= 0, "v2a_set_tuning: negative threshold");
  v2a_detail::g_dwconv_rows_per_wave = t->dwconv_rows_per_wave > 0 ? t->dwconv_rows_per_wave : 4;
  v2a_detail::g_dwconv_stream = t->dwconv_rows_per_wave == -1 ? 0 : 1;
  v2a_detail::g_attn_one_group_from = t->attn_one_group_from > 0 ? t->attn_one_group_from : 1536;
  v2a_detail::g_gemm_tuning = {t->gemm_force_tile, t->gemm_k_rotation ? 1 : 0, t->gemm_8phase,
                               t->gemm_8phase_min_tiles > 0 ? t->gemm_8phase_min_tiles : kDefaultTuning.min_tiles_8phase,
                               t->gemm_xcd_order_1x8 ? 0 : 1, t->reserved[0]};
  v2a_detail::g_probe_dbg = t->reserved[0];
  return V2A_OK;
}

// argument checks + the kernel parameter block
static int gemm_prepare(const v2a_gemm_args* a, GemmParams& p) {
  V2A_REQUIRE(a != nullptr, "v2a_gemm: null args");
  V2A_REQUIRE(a->nseg >= 1 && a->nseg <= 3, "v2a_gemm: nseg=%d", a->nseg);
  V2A_REQUIRE(a->M > 0 && a->N > 0, "v2a_gemm: M=%d N=%d", a->M, a->N);
  V2A_REQUIRE(a->compute_dtype == V2A_F32 || a->compute_dtype == V2A_BF16, "v2a_gemm: compute dtype %d", a->compute_dtype);
  const bool split_in = a->a_dtype == V2A_BF16_SPLIT;   // A rows and W rows as hi | lo bf16 planes: three MFMA products per fp32 product
  V2A_REQUIRE(a->a_dtype == a->compute_dtype || a->a_dtype == V2A_F32 || (split_in && a->compute_dtype == V2A_BF16),
              "v2a_gemm: A dtype %d with compute dtype %d", a->a_dtype, a->compute_dtype);
  const int bk = a->compute_dtype == V2A_BF16 ? 64 : 16;
  const int a_vec = a->a_dtype == V2A_F32 ? 4 : 8;  // elements per 16-byte load
  p = GemmParams{};
  int K = 0;
  for (int s = 0; s < a->nseg; ++s) {
    V2A_REQUIRE(a->a[s] != nullptr, "v2a_gemm: segment %d null", s);
    V2A_REQUIRE(a->ka[s] > 0 && a->ka[s] % bk == 0, "v2a_gemm: segment %d K=%d not a multiple of %d", s, a->ka[s], bk);
    V2A_REQUIRE(a->lda[s] % a_vec == 0 && ((uintptr_t)a->a[s] & 15) == 0, "v2a_gemm: segment %d not 16-byte aligned", s);
    if (split_in) V2A_REQUIRE(a->lda[s] >= 2 * (int64_t)a->ka[s], "v2a_gemm: split segment %d needs lda >= 2 * K (hi | lo planes)", s);
    p.a[s] = a->a[s];
    p.lda[s] = a->lda[s];
    K += a->ka[s];
    p.kend[s] = K;
  }
  p.nseg = a->nseg;
  V2A_REQUIRE(a->w != nullptr && ((uintptr_t)a->w & 15) == 0 && a->ldw % (a->compute_dtype == V2A_BF16 ? 8 : 4) == 0 && a->ldw >= (split_in ? 2 : 1) * (int64_t)K,
              "v2a_gemm: W pointer/ldw (%lld) misaligned or < K=%d (x2 for split operands)", (long long)a->ldw, K);
  V2A_REQUIRE(a->out != nullptr, "v2a_gemm: out null");
  p.w = a->w;
  p.ldw = a->ldw;
  p.bias = a->bias;
  p.M = a->M;
  p.N = a->N;
  p.K = K;
  p.out = a->out;
  p.ldo = a->ldo;
  p.out2 = reinterpret_cast<bf16_t*>(a->out_bf16);
  p.ldo2 = a->ld_out_bf16;
  if (a->out_bf16) V2A_REQUIRE(a->out_dtype == V2A_F32 && a->epilogue != V2A_EPI_GEGLU, "v2a_gemm: out_bf16 shadows an fp32 output only");
  p.out2_split = a->out_bf16 && a->out_bf16_split ? 1 : 0;
  p.out2_lo = a->out_bf16_lo_offset > 0 ? a->out_bf16_lo_offset : a->N;
  if (p.out2_split)
    V2A_REQUIRE(p.out2_lo >= a->N && p.out2_lo % 4 == 0 && a->ld_out_bf16 >= p.out2_lo + (int64_t)a->N,
                "v2a_gemm: a split shadow needs out_bf16_lo_offset (%lld) >= N, a multiple of 4, and ld_out_bf16 >= lo offset + N", (long long)p.out2_lo);
  else
    V2A_REQUIRE(a->out_bf16_lo_offset == 0, "v2a_gemm: out_bf16_lo_offset goes with out_bf16_split");
  for (int sg = 0; sg < a->nseg; ++sg) {
    p.alo[sg] = a->a_lo_offset[sg] > 0 ? a->a_lo_offset[sg] : a->ka[sg];
    if (split_in)
      V2A_REQUIRE(p.alo[sg] >= a->ka[sg] && p.alo[sg] % 8 == 0 && a->lda[sg] >= p.alo[sg] + (int64_t)a->ka[sg],
                  "v2a_gemm: split segment %d: a_lo_offset (%lld) must be >= K, a multiple of 8, and lda >= lo offset + K", sg, (long long)p.alo[sg]);
    else
      V2A_REQUIRE(a->a_lo_offset[sg] == 0, "v2a_gemm: a_lo_offset goes with split operands");
  }
  p.out_split = a->out_dtype == V2A_BF16_SPLIT ? 1 : 0;
  if (p.out_split) V2A_REQUIRE(a->epilogue == V2A_EPI_GEGLU && a->ldo >= a->N, "v2a_gemm: out_dtype V2A_BF16_SPLIT goes with the GEGLU epilogue and ldo >= N");
  p.resid = a->resid;
  p.ldr = a->ldr;
  p.gate = a->gate;
  p.step = a->step;
  p.gss = a->gate_step_stride;
  p.gbs = a->gate_batch_stride;
  p.rpb = a->rows_per_batch > 0 ? a->rows_per_batch : a->M;
  {
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const int osz = a->out_dtype == V2A_F32 ? 4 : 2;      // V2A_BF16 and V2A_BF16_SPLIT: bf16 elements
    const int ncols = a->epilogue == V2A_EPI_GEGLU ? a->N / 2 : a->N;
    bool ok = a->N % 4 == 0 && ncols % 4 == 0 && ((uintptr_t)a->out % (4 * osz)) == 0 && (a->ldo * osz) % (4 * osz) == 0;
    if (a->bias) ok = ok && al16(a->bias);
    if (a->resid) ok = ok && al16(a->resid) && a->ldr % 4 == 0;
    if (a->gate) ok = ok && al16(a->gate) && a->gate_step_stride % 4 == 0 && a->gate_batch_stride % 4 == 0;
    if (a->out_bf16) ok = ok && ((uintptr_t)a->out_bf16 & 7) == 0 && a->ld_out_bf16 % 4 == 0;
    p.vec_epi = ok ? 1 : 0;
  }
  p.relu = a->relu;
  // off by default: +0.7 % throughput, but the fp32 summation order of a row then depends on how many rows the call has, so a
  // clip's result would change (in the last bits) with the batch it is sampled in; v2a_set_tuning enables it
  const v2a_detail::GemmTuning tune = v2a_detail::g_gemm_tuning;
  p.krot = 0;
  p.dbg = tune.dbg;
  {
    // XCD grid over the tile space: minimise A_bytes * gn + W_bytes * gm (gemm_common.h, tile_of_block); v2a_set_tuning can
    // pin the old 1 x 8 order for A/B measurements
    const double ab = (double)a->M * K * (a->a_dtype == V2A_F32 ? 4 : 2), wb = (double)a->N * K * (a->compute_dtype == V2A_F32 ? 4 : 2);
    int best_gm = 1;
    double best = ab * 8 + wb;
    // (PMC, one clip: FETCH of the narrow GEMMs -30 %, L2 hit rate 0.71 -> 0.80; end to end +1.5 % at 8 clips per GPU and
    // -0.5 % .. +0.3 % at one clip, inside the run-to-run spread)
    if (tune.xcd_grid) {
      for (int gm = 2; gm <= 8; gm *= 2) {
        const double c = ab * (8 / gm) + wb * gm;
        if (c < best * 0.95) { best = c; best_gm = gm; }
      }
    }
    p.xcd_gm = best_gm;
    p.xcd_gn = 8 / best_gm;
  }
  p.a_rowoff = a->a_row_offset;
  p.a_koff = a->a_ktile_offset;
  p.o_rowoff = a->out_row_offset;
  if (a->a_row_offset || a->a_ktile_offset || a->out_row_offset) {
    V2A_REQUIRE((a->a_row_offset != nullptr) == (a->a_ktile_offset != nullptr), "v2a_gemm: a_row_offset and a_ktile_offset go together");
    V2A_REQUIRE(a->nseg == 1 && a->compute_dtype == V2A_BF16 && a->a_dtype == V2A_BF16 && p.vec_epi &&
                    (a->epilogue == V2A_EPI_STORE || a->epilogue == V2A_EPI_RESID) && !a->rope_table,
                "v2a_gemm: row/K-tile offset tables need one bf16 segment, bf16 compute, STORE/RESID and 16-byte aligned rows");
  }
  V2A_REQUIRE(!a->relu || a->epilogue != V2A_EPI_GEGLU, "v2a_gemm: relu with GEGLU");
  p.ngam = a->norm_gamma;
  p.ngss = a->norm_step_stride;
  p.ngbs = a->norm_batch_stride;
  p.nsw_row = a->norm_switch_row > 0 ? a->norm_switch_row : INT32_MAX;
  p.nsw_off = a->norm_switch_offset;
  p.ssq = a->norm_ssq;
  p.ssq_ld = a->ld_norm_ssq;
  p.rssq = a->row_ssq;
  p.rssq_ld = a->ld_row_ssq;
  p.rssq_parts = a->row_ssq_parts;
  p.rnorm = sqrtf((float)a->row_norm_dim);
  if (a->norm_gamma || a->norm_ssq || a->row_ssq) {
    V2A_REQUIRE(a->compute_dtype == V2A_BF16 && (a->a_dtype == V2A_BF16 || split_in) && p.vec_epi && a->epilogue != V2A_EPI_SIGMOID && !a->out_row_offset,
                "v2a_gemm: a folded RMSNorm needs bf16 x bf16 (or split bf16) operands, dense rows and 16-byte aligned epilogue operands");
    if (a->norm_gamma || a->norm_ssq)
      V2A_REQUIRE(a->out_bf16 && (a->epilogue == V2A_EPI_RESID || a->epilogue == V2A_EPI_GATE_RESID) && a->N % 32 == 0 &&
                      (!a->norm_gamma || (((uintptr_t)a->norm_gamma & 15) == 0 && a->norm_step_stride % 4 == 0 && a->norm_batch_stride % 4 == 0 &&
                                          a->norm_switch_offset % 4 == 0)) &&
                      (!a->norm_ssq || a->ld_norm_ssq >= a->N / 32),
                  "v2a_gemm: norm_gamma / norm_ssq go with RESID / GATE_RESID, an out_bf16 shadow and N %% 32 == 0 (N=%d)", a->N);
    if (a->row_ssq)
      V2A_REQUIRE(a->row_ssq_parts > 0 && a->row_ssq_parts <= 40 && a->row_norm_dim > 0 && a->ld_row_ssq >= (a->row_ssq_parts + 3) / 4 * 4 &&
                      a->ld_row_ssq % 4 == 0 && ((uintptr_t)a->row_ssq & 15) == 0,
                  "v2a_gemm: row_ssq needs row_ssq_parts <= 40 (%d), row_norm_dim (%d) and rows of whole float4 (zero padded)", a->row_ssq_parts,
                  a->row_norm_dim);
  }
  p.rope = a->rope_table;
  p.rope_cols = a->rope_cols;
  p.rope_pos_off = a->rope_pos_offset;
  if (a->rope_table) {
    V2A_REQUIRE(a->epilogue == V2A_EPI_STORE && a->compute_dtype == V2A_BF16 && (a->a_dtype == V2A_BF16 || split_in) && p.vec_epi &&
                    a->rope_cols % 64 == 0 && a->rope_cols <= a->N && ((uintptr_t)a->rope_table & 15) == 0,
                "v2a_gemm: fused RoPE needs the bf16 STORE epilogue with 16-byte aligned rows and rope_cols %% 64 == 0");
  }
  if (a->epilogue == V2A_EPI_RESID || a->epilogue == V2A_EPI_GATE_RESID)
    V2A_REQUIRE(a->resid != nullptr, "v2a_gemm: epilogue %d needs resid", a->epilogue);
  if (a->epilogue == V2A_EPI_GATE_RESID) V2A_REQUIRE(a->gate != nullptr, "v2a_gemm: GATE_RESID needs gate");
  if (a->epilogue == V2A_EPI_GEGLU) V2A_REQUIRE(a->N % 32 == 0, "v2a_gemm: GEGLU needs N %% 32 == 0 (N=%d)", a->N);
  return V2A_OK;
}

extern "C" int v2a_gemm(const v2a_gemm_args* a, v2a_stream_t stream) {
  GemmParams p;
  if (int rc = gemm_prepare(a, p)) return rc;
  const bool split_in = a->a_dtype == V2A_BF16_SPLIT;
  const int K = p.K;
  const v2a_detail::GemmTuning tune = v2a_detail::g_gemm_tuning;
  hipStream_t s = (hipStream_t)stream;
  if (a->compute_dtype == V2A_F32) {
    // exact-fp32 kernel: 64x64 tiles while 128x128 ones would leave CUs idle (one clip: 13 x 8 tiles for N = 1024); the K order of
    // an output element does not depend on the tile, so the result is the same bit for bit
    const int64_t t128 = (int64_t)((a->M + 127) / 128) * ((a->N + 127) / 128);
    if (t128 < 224 && a->M > 64) return dispatch_epi<float, false, 64, 64>(a, p, s);
    return dispatch_epi<float, false, 128, 128>(a, p, s);
  }
  if (split_in) {
    V2A_REQUIRE(p.vec_epi && !a->a_row_offset && !a->out_row_offset, "v2a_gemm: split operands need dense rows and 16-byte aligned epilogue operands");
    V2A_REQUIRE(a->tile_hint >= 0 && a->tile_hint <= 7, "v2a_gemm: tile_hint %d with split operands (0 = by shape, 1..7)", a->tile_hint);
    auto nt = [&](int bm, int bn) { return (int64_t)((a->M + bm - 1) / bm) * ((a->N + bn - 1) / bn); };
    // split-operand tile shapes (hi + lo planes double a stage): 1 = 64x64 (96 KB), 2 = 128x64 (144 KB), 3 = 128x128 with 8 waves and a
    // 2-deep ring (128 KB), 4 = 64x128 with 8 waves (144 KB), 5 = the 8-phase kernel; with 32-wide K stages (round 5): 6 = 128x256 with
    // 8 waves (wave tile 64x64; 3 stages of 48 KB), 7 = 128x128 with 8 waves (3 stages of 32 KB).  (64x128 with four waves on 32-wide
    // stages -- 72 KB, two workgroups per CU -- measured 15-25 % slower than shape 4 at every size: profiles/r05_split_probe.txt; not kept.)
    int cfg = a->tile_hint;
    // the phase-interleaved 256x256 kernel on three passes over the logical K (hi x hi, hi x lo, lo x hi; it re-reads A_hi and W_hi, but
    // its K loop hides the operand stream behind the MFMAs): wide outputs from 150 tiles (audio feed-forward at one clip: 76 us against
    // 102 us on the best split ring tile) and -- round 5 -- narrow ones (512 < N < 2048, any number of logical segments) from 150 tiles,
    // i.e. from ~6 clips per GPU: at 8 clips the 64x128 split ring ran them at 28 % of the MFMA peak (issued products) against 46 % here.
    // tile_hint 5 asks for it, 0 picks by shape.
    const bool wide8 = tune.use_8phase && (cfg == 5 || (cfg == 0 && a->N > 512 && nt(256, 256) >= 150));
    if (wide8) {
      GemmParams q = p;
      split_as_three_passes(q);
#ifdef V2A_GEMM_PROBE
      if (tune.dbg & 32) { q.s3_kl = 0; q.K = K; }      // error attribution: hi x hi only
#endif
      return v2a_detail::launch_gemm_8phase(q, a->epilogue, a->out_dtype == V2A_BF16_SPLIT ? V2A_BF16 : a->out_dtype, s);
    }
    // (a 128x128 tile whose two wave groups alternate along K -- one multiplies a stage while the other reads and stages the next -- was built
    // and measured in round 5: 20-25 % slower than shape 7, bound by the DMA issue of its loading waves; profiles/r05_pingpong_probe.txt, source
    // kept as scripts/probes/gemm_pingpong.hip.txt)
    if (cfg == 0) {
      // (stand-alone, scripts/split_probe.py, profiles/r05_split_probe.txt)
      if (a->epilogue == V2A_EPI_GEGLU || a->N >= 2048) cfg = nt(128, 128) >= 200 ? 3 : 4;
      // N <= 512 with many rows (the frames stream at 8 clips per GPU: 98 tiles of 256x256 cannot fill the chip on the 8-phase kernel):
      // 128x256 tiles on 32-wide K stages, 960 against 721 TF/s (64x128) at 12512x512x2048
      else if (a->N <= 512 && nt(128, 256) >= 128) cfg = 6;
      else cfg = nt(64, 128) >= 160 ? 4 : 1;
    }
    switch (cfg) {
      case 1: return dispatch_s3<64, 64, 2, 2, 3>(a, p, s);
      case 2: return dispatch_s3<128, 64, 2, 2, 3>(a, p, s);
      case 3: return dispatch_s3<128, 128, 2, 4, 2>(a, p, s);
      case 6: return dispatch_s3<128, 256, 2, 4, 3, 32>(a, p, s);
      case 7: return dispatch_s3<128, 128, 2, 4, 3, 32>(a, p, s);
      default: return dispatch_s3<64, 128, 2, 4, 3>(a, p, s);
    }
  }
  if (a->a_dtype == V2A_F32) return dispatch_epi<bf16_t, true, 128, 128>(a, p, s);
  if (a->epilogue == V2A_EPI_SIGMOID) return dispatch_epi<bf16_t, false, 128, 128>(a, p, s);
  // bf16 x bf16: LDS-DMA kernel; tile shape by how many workgroups the problem yields (256 CUs)
  auto ntiles = [&](int bm, int bn) { return (int64_t)((a->M + bm - 1) / bm) * ((a->N + bn - 1) / bn); };
  // wide outputs: 256x256 tile with the phase-interleaved K loop (gemm_8phase.hip) once the problem yields enough tiles
  V2A_REQUIRE(a->tile_hint >= 0 && a->tile_hint <= 16, "v2a_gemm: tile_hint %d", a->tile_hint);
  const bool dense = !a->a_row_offset && !a->out_row_offset && p.vec_epi;
  if (a->tile_hint > 0 && tune.force_tile < 0) {
    if (a->tile_hint == 7) {
      V2A_REQUIRE(dense, "v2a_gemm: tile_hint 7 (256x256 8-phase) needs dense rows and 16-byte aligned epilogue operands");
      return v2a_detail::launch_gemm_8phase(p, a->epilogue, a->out_dtype, s);
    }
    switch (a->tile_hint - 1) {
      case 5: return dispatch_dma<256, 256, 2, 4, 2>(a, p, s);
      case 0: return dispatch_dma<128, 256, 2, 4>(a, p, s);
      case 1: return dispatch_dma<128, 128, 2, 2>(a, p, s);
      case 2: return dispatch_dma<128, 64, 2, 2>(a, p, s);
      case 7: return dispatch_dma<64, 128, 2, 2, 6>(a, p, s);
      case 8: return dispatch_dma<64, 64, 2, 2, 6>(a, p, s);
      case 9: return dispatch_dma<64, 64, 2, 2, 4>(a, p, s);
      case 10: return dispatch_dma<64, 64, 2, 2, 5>(a, p, s);
      case 11: return dispatch_dma<128, 128, 2, 2, 4>(a, p, s);
      case 12: return dispatch_dma<128, 128, 2, 4, 3>(a, p, s);   // 8 waves (two per SIMD), wave tile 64x32
      case 13: return dispatch_dma<64, 64, 2, 4, 3>(a, p, s);     // 8 waves, wave tile 32x16 (no GEGLU)
      case 14: return dispatch_dma<128, 64, 4, 2, 3>(a, p, s);    // 8 waves, wave tile 32x32
      case 15: return dispatch_dma<64, 128, 2, 4, 3>(a, p, s);    // 8 waves, wave tile 32x32
      default: return dispatch_dma<64, 64, 2, 2>(a, p, s);
    }
  }
  // Tile shape by how many workgroups the problem yields on 256 CUs (stand-alone launches measured with scripts/gemm_probe.py;
  // M = 1564 / 3128 / 6256 / 12512 rows = 1 / 2 / 4 / 8 clips):
  //   * 256x256 phase-interleaved kernel: wide outputs from min_tiles_8phase tiles; narrow outputs (512 < N < 2048) once they
  //     fill >= 150 CUs with one tile each (8 clips: 1045 vs 793 TF/s at 12512x1024x4096, 1135 vs 967 at 12512x1280x5120);
  //   * 128x256 ring kernel: wide outputs below that, narrow ones from ~4 clips (778 vs 534 TF/s at 6256x1024x4096) and
  //     N <= 512 at 8 clips (653 vs 450 TF/s at 12512x512x2048);
  //   * 128x128 from ~2 clips, 64x64 below (one clip: only small tiles give every CU work).
  const bool can8 = tune.use_8phase && dense;
  // a forced 8-phase kernel applies to dense launches only: it knows nothing of the implicit-GEMM offset tables (Video2Roll
  // convolutions), which keep their by-shape choice
  if (tune.force_tile == 6 && dense) return v2a_detail::launch_gemm_8phase(p, a->epilogue, a->out_dtype, s);
  int cfg;
  if (tune.force_tile >= 0 && tune.force_tile != 6) cfg = tune.force_tile;
  else if (a->N <= 64) cfg = ntiles(128, 64) >= 512 ? 2 : 3;          // conv layers with few output channels
  else if (a->N <= 128) cfg = ntiles(128, 128) >= 512 ? 1 : (ntiles(128, 64) >= 512 ? 2 : 3);
  else if (a->N >= 2048) {
    if (can8 && ntiles(256, 256) >= tune.min_tiles_8phase) cfg = 6;
    else if (ntiles(256, 256) >= 512) cfg = 5;                         // 8-phase kernel switched off: 2-deep ring of 64 KB stages
    else if (ntiles(128, 256) >= 96) cfg = 0;
    else cfg = 3;
  } else if (a->N > 512) {
    if (can8 && ntiles(256, 256) >= 150) cfg = 6;
    else if (ntiles(128, 256) >= 180) cfg = 0;
    else if (ntiles(128, 128) >= 180) cfg = 1;
    else if (ntiles(64, 64) >= 2048 && ntiles(128, 64) >= 512) cfg = 2;
    else cfg = 3;
  } else {
    if (ntiles(128, 256) >= 150) cfg = 0;
    else if (ntiles(128, 128) >= 256) cfg = 1;
    else if (ntiles(64, 64) >= 2048 && ntiles(128, 64) >= 512) cfg = 2;
    else cfg = 3;
  }
  if (cfg == 6) return v2a_detail::launch_gemm_8phase(p, a->epilogue, a->out_dtype, s);
  switch (cfg) {
    case 5: return dispatch_dma<256, 256, 2, 4, 2>(a, p, s); // 8 waves, 128x64 wave tiles, 2-deep ring of 64 KB stages
    case 0: return dispatch_dma<128, 256, 2, 4>(a, p, s);   // 8 waves, 144 KB LDS, 1 workgroup/CU
    case 1: return dispatch_dma<128, 128, 2, 2>(a, p, s);   // 4 waves,  96 KB
    case 2: return dispatch_dma<128, 64, 2, 2>(a, p, s);    // 4 waves,  72 KB, 2 workgroups/CU
    case 7: return dispatch_dma<64, 128, 2, 2, 6>(a, p, s); // 4 waves, 144 KB: 6-deep ring, 1 workgroup/CU (<= 256 tiles)
    case 8: return dispatch_dma<64, 64, 2, 2, 6>(a, p, s);  // 4 waves,  96 KB: 6-deep ring, 1 workgroup/CU (<= 256 tiles)
    default: return dispatch_dma<64, 64, 2, 2>(a, p, s);    // 4 waves,  48 KB, 3 workgroups/CU
  }
}
