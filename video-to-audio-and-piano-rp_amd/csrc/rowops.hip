// HBM-bound row kernels of the V2A sampler for gfx950: RMSNorm/AdaptiveRMSNorm, depthwise
// conv + SiLU + residual, rotary embedding, CFG + Euler update, and the small fp32 setup
// kernels (proj_in / proj_frames scatter, register fill, time conditioning).
// All loads/stores are 16 B per lane on the channel-contiguous (B, N, d) layout.
#include "v2a_common.h"
#include <utility>

namespace v2a_detail { extern int g_dwconv_rows_per_wave; extern int g_dwconv_stream; }

thread_local char v2a_err_buf[512] = {0};

extern "C" int v2a_abi_version(void) { return 8; }
extern "C" const char* v2a_last_error(void) { return v2a_err_buf; }

namespace {

// ------------------------------------------------------------------------------------------
// RMSNorm: one wave per row; the row (d <= 2048 floats) stays in registers between the
// sum-of-squares pass and the scale pass, so x is read once: 4*d B in, sizeof(T)*d B out.
// ------------------------------------------------------------------------------------------
template <typename OutT, int VPL /* float4 per lane */, bool SPLIT = false /* bf16 hi | lo planes, lo at + d */>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, int64_t ldx, OutT* __restrict__ y,
                                                      int64_t ldy, int64_t rows, int d, const float* gamma,
                                                      const int32_t* step, int64_t gss, int64_t gbs, int rpb) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  const float* g = step_vec(gamma, step, gss, gbs, row / rpb);
  const int nvec = d >> 2;
  f32x4 v[VPL];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    if (c < nvec) {
      v[i] = *reinterpret_cast<const f32x4*>(xr + 4 * c);
      ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
  }
  ss = wave_sum(ss);
  // F.normalize: x / max(||x||, eps), eps = 1e-12; then * sqrt(d)
  const float inv = sqrtf((float)d) / fmaxf(sqrtf(ss), 1e-12f);
  OutT* yr = y + row * ldy;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    if (c < nvec) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(g + 4 * c);
      if constexpr (sizeof(OutT) == 4) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = v[i][j] * inv * gv[j];
        *reinterpret_cast<f32x4*>(yr + 4 * c) = o;
      } else {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(v[i][j] * inv * gv[j]);
        *reinterpret_cast<bf16x4*>(yr + 4 * c) = o;
        if constexpr (SPLIT) {
          bf16x4 lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) lo[j] = (bf16_t)(v[i][j] * inv * gv[j] - (float)o[j]);
          *reinterpret_cast<bf16x4*>(yr + d + 4 * c) = lo;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Depthwise conv (k taps along N) + bias + SiLU + mask + residual.
// Block = 4 waves = 4 consecutive position tiles of TN outputs for the same 256 channels (a lane owns 4 channels).
// The k x 256 tap weights are fetched ONCE per block into LDS, zero-padded to a multiple of TN taps.  Input row i of a tile
// feeds output t through tap i - t, so a wave walks its rows in groups of TN with a window of TN taps in registers: row
// 8g + r brings tap 8g + r into slot r and output t multiplies slot (r - t) mod TN -- taps before the kernel start are the
// zero-initialised window, taps past its end the zero padding, so the inner body has no conditions.  A real (not unrolled)
// loop over row groups keeps the live set at acc + window + two row groups: three waves per SIMD.  (The first version held
// all 31 taps in registers: 396 VGPRs, one wave per SIMD, and at 8 clips per GPU it ran VALU-bound at 1.4 TB/s with
// nothing to overlap a block's load phase with.)  Algorithmic traffic: 4*d B in + 4*d B out per position.
// ------------------------------------------------------------------------------------------
template <int KS, int TN, bool NORM>
__device__ __forceinline__ void dwconv_body(const float* __restrict__ x, float* __restrict__ out,
                                            const float* __restrict__ wt, const float* __restrict__ bias,
                                            int B, int N, int d, const int32_t* len, int P, int walkers, const v2a_dwconv_norm& nrm,
                                            int xcd, int cblk, int walker) {
  constexpr int HALF = KS / 2;
  constexpr int NG = ((TN + KS - 1 + TN - 1) / TN + 2) / 3 * 3;   // row groups of TN rows, a multiple of the 3 load buffers
  constexpr int TAPS = NG * TN;                          // tap rows in LDS: the k real taps, zero rows for the window slots past them
  __shared__ __attribute__((aligned(16))) float wl[TAPS * 256];
  // Work item = (sequence, position tile of 4*TN outputs); a block stages its 256-channel tap block once and walks items
  // (8 clips per GPU: 3 items per block, so the 40 KB tap stage and the block prologue are paid once per three tiles).
  // XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so flat id % 8 labels the L2 a block uses; each label owns a
  // contiguous range of items and its walkers take neighbouring items at the same time: the k-1 halo rows a tile shares with
  // its neighbours hit in that XCD's L2 instead of being fetched over the fabric once per tile (4.7x the algorithmic bytes
  // without this, measured in round 1).
  const int items = P * B, ipx = (items + 7) >> 3;       // items per XCD label
  if (walker >= walkers) return;                         // whole block (grid padding)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: row tests become scalar
  const int c4 = cblk * 64 + lane;                       // float4 channel group of this lane
  const bool cok = c4 * 4 < d;
  const int cc = cok ? c4 : d / 4 - 1;                   // idle lanes shadow a valid channel group, only the store is masked
  for (int i = threadIdx.x; i < TAPS * 64; i += 256) {   // taps x 64 float4 of this channel block, once per block
    const int tap = i >> 6;
    int ch4 = cblk * 64 + (i & 63);
    ch4 = ch4 * 4 < d ? ch4 : d / 4 - 1;
    f32x4 w = {0.f, 0.f, 0.f, 0.f};
    if (tap < KS) w = *reinterpret_cast<const f32x4*>(wt + (int64_t)tap * d + 4 * ch4);
    *reinterpret_cast<f32x4*>(wl + i * 4) = w;
  }
  __syncthreads();
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * cc);
  const uint32_t loff = (uint32_t)cc * 16u;
  const float* wlane = wl + lane * 4;

  for (int it = walker; it < ipx; it += walkers) {
    const int item = xcd * ipx + it;
    if (item >= items) break;
    const int b = item / P, ptile = item % P;
    const int n0 = (ptile * 4 + wave) * TN;              // wave-uniform
    const int L = len ? min(len[b], N) : N;
    f32x4 acc[TN], win[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      acc[t] = bv;
      win[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // row base is wave-uniform (SGPR), the lane contributes a constant 32-bit byte offset: no vector address math
    const char* xrow0 = reinterpret_cast<const char*>(x + (int64_t)b * N * d);
    const int64_t stride = (int64_t)d * 4;
    f32x4 rowsA[TN], rowsB[TN], rowsC[TN];
    // two row groups in flight ahead of the one being multiplied (three register buffers, loop unrolled by three)
    if (n0 >= HALF && n0 - HALF + NG * TN <= L) {
      // interior tile (all but the first and last ~4 wave tiles of a sequence): every row of the window is inside the valid
      // part of the sequence -- no clamps, no per-row tests, the load address is a scalar pointer stepped by the row stride
      const char* lp = xrow0 + (int64_t)(n0 - HALF) * stride;
      const float* wp = wlane;
      auto load_group = [&](f32x4 (&dst)[TN]) {
#pragma unroll
        for (int r = 0; r < TN; ++r) dst[r] = *reinterpret_cast<const f32x4*>(lp + r * stride + loff);
        lp += TN * stride;
      };
      auto group = [&](const f32x4 (&cur)[TN]) {
#pragma unroll
        for (int r = 0; r < TN; ++r) {
          win[r] = *reinterpret_cast<const f32x4*>(wp + r * 256);
          const f32x4 v = cur[r];
#pragma unroll
          for (int t = 0; t < TN; ++t) acc[t] = win[(r - t + TN) % TN] * v + acc[t];
        }
        wp += TN * 256;
      };
      load_group(rowsA);
      load_group(rowsB);
#pragma unroll 1
      for (int g = 0; g < NG; g += 3) {
        load_group(rowsC);
        group(rowsA);
        if (g + 3 < NG) load_group(rowsA);
        group(rowsB);
        if (g + 4 < NG) load_group(rowsB);
        group(rowsC);
      }
    } else {
      auto load_group = [&](int g, f32x4 (&dst)[TN]) {
#pragma unroll
        for (int r = 0; r < TN; ++r) {
          const int pc = min(max(n0 - HALF + g * TN + r, 0), N - 1);   // clamped into the sequence; skipped below if outside
          dst[r] = *reinterpret_cast<const f32x4*>(xrow0 + pc * stride + loff);
        }
      };
      auto group = [&](int g, const f32x4 (&cur)[TN]) {
#pragma unroll
        for (int r = 0; r < TN; ++r) {
          win[r] = *reinterpret_cast<const f32x4*>(wlane + (g * TN + r) * 256);
          const int pos = n0 - HALF + g * TN + r;          // wave-uniform
          // zero padding and masked rows contribute nothing: a scalar branch around the row's FMAs (a 0/1 factor on the row
          // costs 4 multiplies per row and made the vectoriser pack values ACROSS rows, waiting on every load right after
          // its issue); whole-vector expressions keep the arithmetic as v_pk_fma_f32
          if (pos >= 0 && pos < L) {
            const f32x4 v = cur[r];
#pragma unroll
            for (int t = 0; t < TN; ++t) acc[t] = win[(r - t + TN) % TN] * v + acc[t];
          }
        }
      };
      load_group(0, rowsA);
      load_group(1, rowsB);
#pragma unroll 1
      for (int g = 0; g < NG; g += 3) {
        load_group(g + 2, rowsC);
        group(g, rowsA);
        if (g + 3 < NG) load_group(g + 3, rowsA);
        group(g + 1, rowsB);
        if (g + 4 < NG) load_group(g + 4, rowsB);
        group(g + 2, rowsC);
      }
    }
    if (cok) {
      char* orow0 = reinterpret_cast<char*>(out + (int64_t)b * N * d);
      // the unmasked centre rows for the residual (x3:1082: conv(x, mask) + x) come back from L1/L2 here rather than
      // being held in 32 registers through the tap loop
      f32x4 center[TN];
#pragma unroll
      for (int t = 0; t < TN; ++t) center[t] = *reinterpret_cast<const f32x4*>(xrow0 + min(n0 + t, N - 1) * stride + loff);
      f32x4 gm = {1.f, 1.f, 1.f, 1.f};
      if constexpr (NORM) gm = *reinterpret_cast<const f32x4*>(step_vec(nrm.norm_gamma, nrm.step, nrm.norm_step_stride, nrm.norm_batch_stride, b) + 4 * cc);
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int n = n0 + t;
        if (n < N) {
          f32x4 o = center[t];
          if (n < L) {
            // x * sigmoid(x) with v_exp + v_rcp (2 ulp; an IEEE division is ten more instructions per value)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += acc[t][j] * __builtin_amdgcn_rcpf(1.0f + __expf(-acc[t][j]));
          }
          *reinterpret_cast<f32x4*>(orow0 + n * stride + loff) = o;
          if constexpr (NORM) {
            // the RMSNorm after the conv, folded: the bf16 operand of the next GEMM carries gamma, the row's sum of squares
            // is left per 32 channels (a lane octet) for that GEMM's epilogue
            const int64_t row = (int64_t)b * N + n;
            bf16x4 ob;
#pragma unroll
            for (int j = 0; j < 4; ++j) ob[j] = (bf16_t)(o[j] * gm[j]);
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(nrm.out_bf16) + row * nrm.ld_out_bf16 + 4 * cc) = ob;
            if (nrm.split) {      // bf16x3 mode: the lo plane of the gamma-scaled row, d columns further
              bf16x4 ol;
#pragma unroll
              for (int j = 0; j < 4; ++j) ol[j] = (bf16_t)(o[j] * gm[j] - (float)ob[j]);
              *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(nrm.out_bf16) + row * nrm.ld_out_bf16 + d + 4 * cc) = ol;
            }
            const float ss = octet_sum(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
            if ((lane & 7) == 0) nrm.norm_ssq[row * nrm.ld_norm_ssq + (cc >> 3)] = ss;
          }
        }
      }
    }
  }
}

template <int KS, int TN, bool NORM>
__global__ __launch_bounds__(256, TN == 4 ? 3 : 2) void dwconv_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                        const float* __restrict__ wt, const float* __restrict__ bias,
                                                        int B, int N, int d, const int32_t* len, int P, int walkers, v2a_dwconv_norm nrm) {
  const int CB = gridDim.x;
  const int flat = blockIdx.y * CB + blockIdx.x;
  const int xcd = flat & 7, slot = flat >> 3;
  const int cblk = slot % CB, walker = slot / CB;        // walkers per XCD label = `walkers`
  dwconv_body<KS, TN, NORM>(x, out, wt, bias, B, N, d, len, P, walkers, nrm, xcd, cblk, walker);
}

// ------------------------------------------------------------------------------------------
// Depthwise conv, streaming form for launches that fill the chip (round 3): every input row goes through LDS ONCE per block.
// The kernel above lets each wave read its own TN + k - 1 rows from global memory per TN outputs -- 8.5x the algorithmic reads
// at TN = 4, all of them L1 / L2 hits (64 B/clk/CU) that bound it at ~2.5 TB/s of the 8 TB/s (PMC: L2 requests 7.6x the
// algorithmic reads).  Here a block owns a segment of SEG positions x 256 channels of one sequence.  The LAST wave is a LOADER:
// it streams the rows by LDS-DMA (global_load_lds_dwordx4: a row piece of 256 channels = one 1 KB wave-instruction, 16 rows per
// chunk) into a 128-row ring (128 KB) with three chunks (48 KB) in flight; it issues nothing but those DMAs, so its counted
// `s_waitcnt vmcnt(32)` is exact -- a wave that also stores cannot wait for one of its older DMAs without waiting for the
// acknowledgement of its freshest stores or counting them (a first version without the loader, one chunk in flight and
// vmcnt(0) per step, ran at the speed of the kernel above: 16 KB in flight per CU against ~2.5 us of loaded HBM latency).
// The NCW = 8 compute waves each produce 16 / NCW = 2 outputs per chunk: they read the k + 1 rows of their outputs from LDS
// (256 B/clk/CU) with compile-time tap indices, in batches of 8 rows, and the residual rows from the same ring.  Chunks whose
// window lies inside the sequence take a straight-line path with no per-row test (a wave-uniform test per row makes the
// compiler branch around every LDS read and serialises them: measured 2x slower); the first and last steps of a sequence take
// the tested path.  The k taps of a lane's four channels are split between registers (the first KR = 24) and an LDS copy (the
// rest), which keeps a compute wave inside the 256 registers two waves on one SIMD leave each.  One barrier per 16 positions.
// ------------------------------------------------------------------------------------------
template <int KS, bool NORM, int NCW>
__global__ __launch_bounds__(64 * (NCW + 1)) void dwconv_stream_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                            const float* __restrict__ wt, const float* __restrict__ bias,
                                                            int B, int N, int d, const int32_t* len, int SEG, int nseg, int CB,
                                                            v2a_dwconv_norm nrm) {
  constexpr int TO = 16 / NCW;                                                 // outputs per compute wave per chunk (NCW compute waves + one loader)
  constexpr int HALF = KS / 2, CH = 16, RING_ROWS = 128, WIN = TO + KS - 1;   // chunk rows, ring rows, rows a wave reads per chunk
  constexpr int KR = 24;                                                       // taps kept in registers; taps KR .. KS-1 live in LDS
  static_assert(HALF < CH && WIN <= 3 * CH, "window must fit three chunks");
  extern __shared__ __attribute__((aligned(16))) char ring[];                  // RING_ROWS x 1 KB, then (KS - KR) x 1 KB of taps
  char* tapl = ring + RING_ROWS * 1024;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // item = (sequence, segment, channel block); channel blocks of one (sequence, segment) are neighbours
  int item = blockIdx.x;
  const int cblk = item % CB;
  item /= CB;
  const int seg = item % nseg, b = item / nseg;
  const int p0 = seg * SEG, p1 = min(p0 + SEG, N);
  const int nch = (p1 - p0 + CH - 1) / CH;           // output chunks of this segment
  const int w0 = p0 - CH;                            // sequence row of ring row 0: chunk j holds rows w0 + 16 j .. + 15
  const int c4 = cblk * 64 + lane;
  const bool cok = c4 * 4 < d;
  const int cc = cok ? c4 : d / 4 - 1;               // idle lanes shadow a valid channel group; only their stores are masked
  const int L = len ? min(len[b], N) : N;
  const float* xb = x + (int64_t)b * N * d;

  if (wave == NCW) {
    // ---- loader: chunk j -> ring rows (16 j .. 16 j + 15) mod 128
    const uint32_t loff = (uint32_t)cc * 16u;
    auto issue = [&](int j) {
#pragma unroll
      for (int r = 0; r < CH; ++r) {
        int row = w0 + CH * j + r;
        row = row < 0 ? 0 : (row >= N ? N - 1 : row);            // clamped: rows outside the sequence are never multiplied
        const char* src = reinterpret_cast<const char*>(xb + (int64_t)row * d) + loff;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ring + (((CH * j + r) & (RING_ROWS - 1)) << 10)), 16, 0, 0);
      }
    };
    const int last = nch + 1;                        // last chunk any window touches
    issue(0);
    issue(1);
    issue(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // barrier of step 0: chunks 0 .. 2 are in the ring
    for (int j = 3; j <= 5; ++j)
      if (j <= last) issue(j);
    for (int i = 1; i < nch; ++i) {
      // chunk i + 2 must have landed; chunks i + 3 and i + 4 (32 DMAs, issued later) may stay in flight.  Near the end fewer
      // chunks are outstanding and the counted wait would let the needed one through unfinished: wait for everything there.
      if (i + 4 <= last) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (i + 5 <= last) issue(i + 5);               // ring rows of chunk i - 3: dead since step i - 1
    }
    return;
  }

  // ---- compute waves
  for (int k = KR + wave; k < KS; k += NCW)            // taps KR .. KS-1 into LDS, one 1 KB row each (visible after the first barrier)
    *reinterpret_cast<f32x4*>(tapl + (k - KR) * 1024 + lane * 16) = *reinterpret_cast<const f32x4*>(wt + (int64_t)k * d + 4 * cc);
  f32x4 tap[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) tap[k] = *reinterpret_cast<const f32x4*>(wt + (int64_t)k * d + 4 * cc);
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * cc);
  const float* gmp = NORM ? step_vec(nrm.norm_gamma, nrm.step, nrm.norm_step_stride, nrm.norm_batch_stride, b) + 4 * cc : nullptr;
  float* ob = out + (int64_t)b * N * d;
  const char* lbase = ring + lane * 16;
  const char* tbase = tapl + lane * 16;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS taps are written before the first barrier lets anyone read them
  for (int i = 0; i < nch; ++i) {
    __builtin_amdgcn_s_barrier();                    // the loader has waited for chunk i + 2: rows w0 + 16 i .. + 47 are in the ring
    const int q0 = p0 + CH * i + TO * wave;          // first of this wave's TO outputs
    const int r0 = q0 - HALF;                        // sequence row of window row 0
    const int rr0 = r0 - w0;                         // its ring row (before the modulo): 16 i + 4 wave + 1
    f32x4 acc[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) acc[t] = bv;
    auto fma_row = [&](auto jc, const f32x4& v) {    // window row j feeds output t through tap j - t (compile-time indices)
      constexpr int j = decltype(jc)::value;
#pragma unroll
      for (int t = 0; t < TO; ++t) {
        constexpr int dummy = 0; (void)dummy;
        const int k = j - t;
        if (k >= 0 && k < KS) {
          const f32x4 w = k < KR ? tap[k < KR ? k : 0] : *reinterpret_cast<const f32x4*>(tbase + ((k < KR ? 0 : k - KR) << 10));
          acc[t] = w * v + acc[t];
        }
      }
    };
    if (r0 >= 0 && r0 + WIN <= L) {
      // interior (all but the first and last chunks of a sequence): every window row is valid -- one straight-line block, no
      // per-row tests (a wave-uniform test per row makes the compiler branch around each row and wait for its LDS read alone)
      // rows in batches of RB: RB LDS reads in flight, then their FMAs (left alone the compiler reads a row, waits for it and multiplies,
      // row by row: one exposed LDS latency per row)
      constexpr int RB = 8;
      auto batch = [&](auto b0c) {
        constexpr int B0 = decltype(b0c)::value;
        f32x4 v[RB];
        [&]<int... U>(std::integer_sequence<int, U...>) {
          ((B0 + U < WIN ? (void)(v[U] = *reinterpret_cast<const f32x4*>(lbase + (((rr0 + B0 + U) & (RING_ROWS - 1)) << 10))) : (void)0), ...);
        }(std::make_integer_sequence<int, RB>{});
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... U>(std::integer_sequence<int, U...>) {
          ((B0 + U < WIN ? fma_row(std::integral_constant<int, (B0 + U < WIN ? B0 + U : 0)>{}, v[U]) : (void)0), ...);
        }(std::make_integer_sequence<int, RB>{});
        __builtin_amdgcn_sched_barrier(0);
      };
      [&]<int... Bk>(std::integer_sequence<int, Bk...>) {
        (batch(std::integral_constant<int, Bk * RB>{}), ...);
      }(std::make_integer_sequence<int, (WIN + RB - 1) / RB>{});
    } else {
      [&]<int... J>(std::integer_sequence<int, J...>) {
        ((r0 + J >= 0 && r0 + J < L   // zero padding and masked rows contribute nothing
              ? fma_row(std::integral_constant<int, J>{}, *reinterpret_cast<const f32x4*>(lbase + (((rr0 + J) & (RING_ROWS - 1)) << 10)))
              : (void)0), ...);
      }(std::make_integer_sequence<int, WIN>{});
    }
    if (cok) {
      f32x4 gm = {1.f, 1.f, 1.f, 1.f};
      if constexpr (NORM) gm = *reinterpret_cast<const f32x4*>(gmp);
#pragma unroll
      for (int t = 0; t < TO; ++t) {
        const int n = q0 + t;
        if (n < p1) {
          // the residual row itself (x3:1082: conv(x, mask) + x), unmasked, from the ring: window row HALF + t
          f32x4 o = *reinterpret_cast<const f32x4*>(lbase + (((rr0 + HALF + t) & (RING_ROWS - 1)) << 10));
          if (n < L) {
            // x * sigmoid(x) with v_exp + v_rcp (2 ulp; an IEEE division is ten more instructions per value)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += acc[t][e] * __builtin_amdgcn_rcpf(1.0f + __expf(-acc[t][e]));
          }
          *reinterpret_cast<f32x4*>(ob + (int64_t)n * d + 4 * cc) = o;
          if constexpr (NORM) {
            const int64_t row = (int64_t)b * N + n;
            bf16x4 obf;
#pragma unroll
            for (int e = 0; e < 4; ++e) obf[e] = (bf16_t)(o[e] * gm[e]);
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(nrm.out_bf16) + row * nrm.ld_out_bf16 + 4 * cc) = obf;
            if (nrm.split) {
              bf16x4 ol;
#pragma unroll
              for (int e = 0; e < 4; ++e) ol[e] = (bf16_t)(o[e] * gm[e] - (float)obf[e]);
              *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(nrm.out_bf16) + row * nrm.ld_out_bf16 + d + 4 * cc) = ol;
            }
            const float ss = octet_sum(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
            if ((lane & 7) == 0) nrm.norm_ssq[row * nrm.ld_norm_ssq + (cc >> 3)] = ss;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// RoPE in place on nheads x 64 columns of each row.  One thread rotates 4 pairs.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T* __restrict__ qk, int64_t rows, int64_t row_stride, int nheads,
                                                   int rpb, int pos_offset, const float* __restrict__ cs, int layout) {
  // work item = (row, head, quad) with quad in [0, 8): pairs 4*quad .. 4*quad+3
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_row = (int64_t)nheads * 8;
  if (idx >= rows * per_row) return;
  const int64_t row = idx / per_row;
  const int rem = (int)(idx % per_row);
  const int head = rem >> 3, quad = rem & 7;
  const int pos = pos_offset + (int)(row % rpb);
  T* base = qk + row * row_stride + head * 64;
  const float* c = cs + ((int64_t)pos * 32 + quad * 4) * 2;
  float a[4], b[4];
  if (layout == 0) {  // interleaved: pair i = (2i, 2i+1) -> 8 consecutive elements
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = to_f32(base[quad * 8 + 2 * i]);
      b[i] = to_f32(base[quad * 8 + 2 * i + 1]);
    }
  } else {            // half split: pair i = (i, i + 32)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = to_f32(base[quad * 4 + i]);
      b[i] = to_f32(base[32 + quad * 4 + i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float co = c[2 * i], si = c[2 * i + 1];
    const float ra = a[i] * co - b[i] * si;   // t*cos + rotate_half(t)*sin, rotate_half = (-x2, x1)
    const float rb = b[i] * co + a[i] * si;
    a[i] = ra;
    b[i] = rb;
  }
  if (layout == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      base[quad * 8 + 2 * i] = from_f32<T>(a[i]);
      base[quad * 8 + 2 * i + 1] = from_f32<T>(b[i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      base[quad * 4 + i] = from_f32<T>(a[i]);
      base[32 + quad * 4 + i] = from_f32<T>(b[i]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Small fp32 linear with row scatter (proj_in + abs_pos_emb + register rows + both CFG halves + bf16
// shadow; proj_frames).  grid = (row blocks of 8, batch); a thread owns 4 consecutive output columns
// of 8 rows: per k one coalesced float4 of wt[k][:] and 8 LDS broadcasts feed 32 FMAs.
// ------------------------------------------------------------------------------------------
template <int ROWS>
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ a, int K, const float* __restrict__ wt,
                                                           const float* __restrict__ bias, const float* __restrict__ add,
                                                           int T, float* __restrict__ out, int64_t obs, int row_off, int d,
                                                           int dup, const float* __restrict__ regs, bf16_t* __restrict__ out2) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // A rows in LDS as (a, a) PAIRS, [ROWS][K][2]: the accumulation below is then plain two-wide vector math and compiles to v_pk_fma_f32
  // on whole register pairs.  With single values in LDS the compiler picked v_pk_fma_f32 ... op_sel:[0,1,0] (the HIGH half of a
  // loaded pair broadcast to both results) for every odd k, and that instruction form returned wrong sums in lanes 48..63
  // whenever ANOTHER PROCESS ran MFMA kernels on the same GPU (two ranks sharing one device; never inside one process): minimal
  // reproduction with the instruction forms side by side in scripts/probes/pkfma_probe.hip, log profiles/r03_pkfma_cross_process.txt
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2* as = reinterpret_cast<f32x2*>(smem_raw);
  const int b = blockIdx.y;
  const int fill = regs ? row_off : 0;             // leading register rows written by this launch
  const int rr0 = blockIdx.x * ROWS;               // row index within [0, fill + T)
  for (int i = threadIdx.x; i < ROWS * K; i += blockDim.x) {
    const int t = rr0 + i / K - fill;
    const float v = (t >= 0 && t < T) ? a[((int64_t)b * T + t) * K + (i % K)] : 0.f;
    as[i] = f32x2{v, v};
  }
  __syncthreads();
  for (int n = threadIdx.x * 4; n < d; n += blockDim.x * 4) {
    f32x2 acc2[ROWS][2];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc2[r][0] = acc2[r][1] = f32x2{0.f, 0.f};
#pragma unroll 8
    for (int k = 0; k < K; ++k) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(wt + (int64_t)k * d + n);
      const f32x2 w0 = {w[0], w[1]}, w1 = {w[2], w[3]};
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const f32x2 aa = as[r * K + k];
        acc2[r][0] += aa * w0;
        acc2[r][1] += aa * w1;
      }
    }
    f32x4 acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = f32x4{acc2[r][0][0], acc2[r][0][1], acc2[r][1][0], acc2[r][1][1]};
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int rr = rr0 + r;
      if (rr >= fill + T) break;
      f32x4 v;
      int orow;
      if (rr < fill) {
        v = *reinterpret_cast<const f32x4*>(regs + (int64_t)rr * d + n);
        orow = rr;
      } else {
        const int t = rr - fill;
        v = acc[r];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bv[e];
        if (add) {
          const f32x4 ad = *reinterpret_cast<const f32x4*>(add + (int64_t)t * d + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += ad[e];
        }
        orow = row_off + t;
      }
      bf16x4 o2;
#pragma unroll
      for (int e = 0; e < 4; ++e) o2[e] = (bf16_t)v[e];
      for (int h = 0; h < (dup > 0 ? 2 : 1); ++h) {
        const int64_t off = (int64_t)(b + h * dup) * obs + (int64_t)orow * d + n;
        *reinterpret_cast<f32x4*>(out + off) = v;
        if (out2) *reinterpret_cast<bf16x4*>(out2 + off) = o2;
      }
    }
  }
}

__global__ void fill_registers_kernel(float* __restrict__ out, int64_t obs, const float* __restrict__ regs, int B, int R,
                                      int d) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)R * d;
  if (idx >= per * B) return;
  const int64_t b = idx / per, r = idx % per;
  out[b * obs + r] = regs[r];
}

// time conditioning: grid = (grid points, column blocks of 256); wt rows are read coalesced, 8 k in flight
__global__ __launch_bounds__(256) void time_cond_kernel(const float* __restrict__ t, const float* __restrict__ fw,
                                                        const float* __restrict__ wt, const float* __restrict__ bias,
                                                        float* __restrict__ out, int d) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* e = reinterpret_cast<float*>(smem_raw);  // [d + 1]
  const int s = blockIdx.x;
  const float ts = t[s];
  const int half = d / 2;
  if (threadIdx.x == 0) e[0] = ts;
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float f = ts * fw[i] * 2.0f * 3.14159265358979323846f;  // x3:562 order: (t*w)*2*pi
    e[1 + i] = sinf(f);
    e[1 + half + i] = cosf(f);
  }
  __syncthreads();
  const int n = blockIdx.y * blockDim.x + threadIdx.x;
  if (n >= d) return;
  float acc = bias[n];
#pragma unroll 8
  for (int k = 0; k < d + 1; ++k) acc += e[k] * wt[(int64_t)k * d + n];
  out[(int64_t)s * d + n] = silu_f(acc);
}

// ------------------------------------------------------------------------------------------
// CFG + Euler.  Algorithmic traffic per element: read pc, pn, y + write y = 16 B.
// ------------------------------------------------------------------------------------------
// (Round 5 measured the step counter advanced by this launch itself -- every block counts in through a device atomic once it has used
// step[0], the last one increments -- instead of by the one-thread launch behind it: 750 same-address atomics cost the launch ~3 us
// (8.3 against 5.2 us at 8 clips, profiles/r05_kernel_stats_bf16x3_8clips_singlestream.csv), as much as the launch they replace.  Not kept.)
__global__ __launch_bounds__(256) void cfg_euler_kernel(float* __restrict__ y, const float* __restrict__ pred, int B,
                                                        int T, int C, int64_t pbs, int row_off, float s,
                                                        const float* __restrict__ dt, const int32_t* step,
                                                        const double* apg, float keep) {
  const int64_t idx4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_b = (int64_t)T * C;
  const int64_t total4 = (int64_t)B * per_b / 4;
  if (idx4 >= total4) return;
  const int64_t e = idx4 * 4;
  const int64_t b = e / per_b, r = e % per_b;
  const float h = dt[step ? step[0] : 0];
  const f32x4 pc = *reinterpret_cast<const f32x4*>(pred + b * pbs + (int64_t)row_off * C + r);
  const f32x4 pn = *reinterpret_cast<const f32x4*>(pred + (b + B) * pbs + (int64_t)row_off * C + r);
  f32x4 yv = *reinterpret_cast<const f32x4*>(y + e);
  float coef = 0.f;
  if (apg) {
    // parallel = (<upd, pred> / <pred, pred>) * pred  (unit = pred / max(|pred|, 1e-12))
    const double dot = apg[2 * b], nn = apg[2 * b + 1];
    const double nrm = sqrt(nn) > 1e-12 ? sqrt(nn) : 1e-12;
    coef = (float)(dot / (nrm * nrm));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float upd = pc[j] - pn[j];
    if (apg) {
      const float par = coef * pc[j];
      upd = (upd - par) + par * keep;
    }
    yv[j] += h * (pc[j] + upd * s);
  }
  *reinterpret_cast<f32x4*>(y + e) = yv;
}

__global__ __launch_bounds__(256) void apg_reduce_kernel(const float* __restrict__ pred, double* __restrict__ apg, int B,
                                                         int T, int C, int64_t pbs, int row_off, const int32_t* __restrict__ valid_rows) {
  // grid = (chunks, B); fp64 partial sums, one atomic pair per block.  valid_rows[0] (device side, so that ONE captured graph serves
  // every length of a shape bucket) bounds the rows summed: the reference sums over the batch's own (b, n, C) tensor, x3:162-173 --
  // rows a bucketed plan pads behind n are not part of it
  const int b = blockIdx.y;
  int rows = T;
  if (valid_rows) rows = min(T, max(0, valid_rows[0]));
  const int64_t per_b = (int64_t)rows * C;
  double dot = 0.0, nn = 0.0;
  const float* pc = pred + (int64_t)b * pbs + (int64_t)row_off * C;
  const float* pn = pred + (int64_t)(b + B) * pbs + (int64_t)row_off * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_b; i += (int64_t)gridDim.x * blockDim.x) {
    const double c = pc[i], n = pn[i];
    dot += (c - n) * c;
    nn += c * c;
  }
  __shared__ double sd[256], sn[256];
  sd[threadIdx.x] = dot;
  sn[threadIdx.x] = nn;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sd[threadIdx.x] += sd[threadIdx.x + o];
      sn[threadIdx.x] += sn[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(&apg[2 * b], sd[0]);
    atomicAdd(&apg[2 * b + 1], sn[0]);
  }
}

__global__ void step_advance_kernel(int32_t* step) { step[0] += 1; }

// rows x d fp32 -> bf16 hi | lo planes (V2A_BF16_SPLIT); one thread per float4
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, int64_t ldx, bf16_t* __restrict__ y, int64_t ldy,
                                                         int64_t rows, int d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per = d >> 2;
  if (i >= rows * per) return;
  const int64_t r = i / per;
  const int c = (int)(i % per) * 4;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
  bf16x4 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = (bf16_t)v[j];
    lo[j] = (bf16_t)(v[j] - (float)hi[j]);
  }
  *reinterpret_cast<bf16x4*>(y + r * ldy + c) = hi;
  *reinterpret_cast<bf16x4*>(y + r * ldy + d + c) = lo;
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
  bf16x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (bf16_t)v[j];
  *reinterpret_cast<bf16x4*>(y + 4 * i) = o;
}

}  // namespace

// ==========================================================================================
extern "C" int v2a_rmsnorm(const float* x, int64_t ldx, void* y, int64_t ldy, int32_t y_dtype, int64_t rows, int32_t d,
                           const float* gamma, const int32_t* step, int64_t gss, int64_t gbs, int32_t rpb,
                           v2a_stream_t stream) {
  V2A_REQUIRE(x && y && gamma, "v2a_rmsnorm: null pointer");
  V2A_REQUIRE(rows > 0 && d > 0 && d % 4 == 0 && d <= 2048, "v2a_rmsnorm: d=%d (need d %% 4 == 0, d <= 2048)", d);
  V2A_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && gss % 4 == 0 && gbs % 4 == 0, "v2a_rmsnorm: strides must be multiples of 4");
  V2A_REQUIRE(y_dtype == V2A_F32 || y_dtype == V2A_BF16 || y_dtype == V2A_BF16_SPLIT, "v2a_rmsnorm: y dtype %d", y_dtype);
  V2A_REQUIRE(y_dtype != V2A_BF16_SPLIT || ldy >= 2 * (int64_t)d, "v2a_rmsnorm: split output needs ldy >= 2*d");
  if (rpb <= 0) rpb = (int32_t)rows;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const int vpl = (d / 4 + 63) / 64;
#define V2A_RMS(OT, V) \
  hipLaunchKernelGGL((rmsnorm_kernel<OT, V>), grid, block, 0, s, x, ldx, (OT*)y, ldy, rows, d, gamma, step, gss, gbs, rpb)
  // exact float4-per-lane counts for the widths of the path (512 -> 2, 1024 -> 4, 1280 -> 5): with the 8-wide instantiation
  // d = 1280 carried three dead, bounds-checked vectors per lane and streamed 2.9 TB/s against 5.1 TB/s at d = 1024
  if (y_dtype == V2A_BF16_SPLIT) {
#define V2A_RMS_S(V) hipLaunchKernelGGL((rmsnorm_kernel<bf16_t, V, true>), grid, block, 0, s, x, ldx, (bf16_t*)y, ldy, rows, d, gamma, step, gss, gbs, rpb)
    if (vpl <= 1) V2A_RMS_S(1); else if (vpl <= 2) V2A_RMS_S(2); else if (vpl <= 3) V2A_RMS_S(3); else if (vpl <= 4) V2A_RMS_S(4);
    else if (vpl <= 5) V2A_RMS_S(5); else if (vpl <= 6) V2A_RMS_S(6); else V2A_RMS_S(8);
#undef V2A_RMS_S
  } else if (y_dtype == V2A_F32) {
    if (vpl <= 1) V2A_RMS(float, 1); else if (vpl <= 2) V2A_RMS(float, 2); else if (vpl <= 3) V2A_RMS(float, 3); else if (vpl <= 4) V2A_RMS(float, 4);
    else if (vpl <= 5) V2A_RMS(float, 5); else if (vpl <= 6) V2A_RMS(float, 6); else V2A_RMS(float, 8);
  } else {
    if (vpl <= 1) V2A_RMS(bf16_t, 1); else if (vpl <= 2) V2A_RMS(bf16_t, 2); else if (vpl <= 3) V2A_RMS(bf16_t, 3); else if (vpl <= 4) V2A_RMS(bf16_t, 4);
    else if (vpl <= 5) V2A_RMS(bf16_t, 5); else if (vpl <= 6) V2A_RMS(bf16_t, 6); else V2A_RMS(bf16_t, 8);
  }
#undef V2A_RMS
  return v2a_check_launch("v2a_rmsnorm");
}

static int dwconv_launch(const float* x, float* out, const float* wt, const float* bias, int32_t B, int32_t N, int32_t d, int32_t ksize,
                         const int32_t* len, const v2a_dwconv_norm* norm, v2a_stream_t stream) {
  V2A_REQUIRE(x && out && wt && bias, "v2a_dwconv: null pointer");
  V2A_REQUIRE(x != out, "v2a_dwconv: out must not alias x (halo reads)");
  V2A_REQUIRE(ksize == 31, "v2a_dwconv: kernel_size %d (only 31 is built, x3:726)", ksize);
  V2A_REQUIRE(d % 4 == 0 && B > 0 && N > 0, "v2a_dwconv: B=%d N=%d d=%d", B, N, d);
  if (norm)
    V2A_REQUIRE(norm->out_bf16 && norm->norm_gamma && norm->norm_ssq && d % 32 == 0 && norm->ld_out_bf16 % 4 == 0 && norm->ld_out_bf16 >= d &&
                    ((uintptr_t)norm->out_bf16 & 7) == 0 && ((uintptr_t)norm->norm_gamma & 15) == 0 && norm->norm_step_stride % 4 == 0 &&
                    norm->norm_batch_stride % 4 == 0 && norm->ld_norm_ssq >= d / 32 && (!norm->split || norm->ld_out_bf16 >= 2 * (int64_t)d),
                "v2a_dwconv: the folded norm needs out_bf16, norm_gamma, norm_ssq, d %% 32 == 0 (d=%d) and aligned rows", d);
  const int cb = (d / 4 + 63) / 64;                      // 256-channel blocks
  hipStream_t s = (hipStream_t)stream;
  {
    // streaming form (rows through LDS once per block) when the launch can fill the chip with segments of >= 96 positions:
    // one workgroup per CU (its registers and the 64 KB ring leave room for no second one)
    int nseg = (256 + B * cb - 1) / (B * cb);
    nseg = nseg < 1 ? 1 : nseg;
    int SEG = ((N + nseg - 1) / nseg + 15) / 16 * 16;
    if (SEG < 96) SEG = 96;
    nseg = (N + SEG - 1) / SEG;
    const int64_t blocks = (int64_t)B * cb * nseg;
    // ... and only when the one-workgroup-per-CU grid fills its last round (8 clips, d = 1024: 256 blocks, 30.3 us against 39.2 us with
    // the folded norm; d = 1280 gives 320 blocks = 1.25 rounds, or 240 blocks of 17 chunks -- 50.7 us against 44.4 us -- and stays on the
    // per-wave kernel; d = 512 gives 224 blocks of 7 chunks: 18.4 / 21.3 us against 19.8 / 22.0 us, round 5 -- within the noise, not taken)
    const int64_t rounds = (blocks + 255) / 256;
    if (v2a_detail::g_dwconv_stream && blocks >= 192 && blocks * 100 >= rounds * 256 * 97 && SEG <= 256) {
      const v2a_dwconv_norm none{};
      constexpr size_t smem = (128 + 31 - 24) * 1024;      // the row ring + the taps kept in LDS
      if (norm) {
        static std::atomic<uint64_t> lds_set{0};
        if (int rc = v2a_enable_lds(reinterpret_cast<const void*>(dwconv_stream_kernel<31, true, 8>), smem, lds_set, "v2a_dwconv(stream)")) return rc;
        hipLaunchKernelGGL((dwconv_stream_kernel<31, true, 8>), dim3((unsigned)blocks), dim3(576), smem, s, x, out, wt, bias, B, N, d, len, SEG, nseg, cb, *norm);
      } else {
        static std::atomic<uint64_t> lds_set{0};
        if (int rc = v2a_enable_lds(reinterpret_cast<const void*>(dwconv_stream_kernel<31, false, 8>), smem, lds_set, "v2a_dwconv(stream)")) return rc;
        hipLaunchKernelGGL((dwconv_stream_kernel<31, false, 8>), dim3((unsigned)blocks), dim3(576), smem, s, x, out, wt, bias, B, N, d, len, SEG, nseg, cb, none);
      }
      return v2a_check_launch("v2a_dwconv_silu_residual(stream)");
    }
  }
  const int TN = norm ? 4 : v2a_detail::g_dwconv_rows_per_wave;
  const int P = (N + 4 * TN - 1) / (4 * TN);             // position tiles of 4 waves x TN outputs
  const int items = P * B, ipx = (items + 7) / 8;        // work items, items per XCD label
  // walkers per XCD label: one per item until the resident blocks per CU (registers: 3 at 4 rows per wave, else 2) are reached,
  // then blocks walk items in `rounds` equal rounds
  const int cap = max(1, (TN == 4 ? 3 : 2) * 256 / 8 / cb);
  const int rounds = (ipx + cap - 1) / cap;
  const int walkers = (ipx + rounds - 1) / rounds;
  V2A_REQUIRE(walkers >= 1 && (int64_t)walkers * 8 <= 65535 && (int64_t)walkers * rounds * 8 >= items,
              "v2a_dwconv: grid does not cover the work (items=%d walkers=%d rounds=%d)", items, walkers, rounds);
  dim3 grid(cb, walkers * 8), block(256);
  const v2a_dwconv_norm none{};
  // The instantiation that is about to run must be launchable as compiled: 256 threads inside its register budget
  // (__launch_bounds__(256, 3) = 168 VGPRs; a variant that needs more is built with a smaller maxThreadsPerBlock) and its static
  // tap stage inside the 64 KB a kernel gets without opting in.  Asked once per (instantiation, device); a variant that does not
  // fit is refused here with V2A_ERR_ARG instead of reaching the runtime.
  auto launchable = [&](const void* kern, std::atomic<uint64_t>& done, const char* what) -> int {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return V2A_OK;
    hipFuncAttributes fa{};
    const hipError_t e = hipFuncGetAttributes(&fa, kern);
    if (e != hipSuccess) return v2a_fail(V2A_ERR_LAUNCH, "v2a_dwconv(%s): hipFuncGetAttributes: %s", what, hipGetErrorString(e));
    if (fa.maxThreadsPerBlock < 256 || fa.sharedSizeBytes > 64 * 1024)
      return v2a_fail(V2A_ERR_ARG, "v2a_dwconv(%s): kernel variant not launchable with 256 threads (max threads %d, %d VGPRs, %zu bytes of LDS)", what,
                      fa.maxThreadsPerBlock, fa.numRegs, (size_t)fa.sharedSizeBytes);
    done.fetch_or(bit, std::memory_order_release);
    return V2A_OK;
  };
  if (norm) {
    static std::atomic<uint64_t> ok{0};
    if (int rc = launchable(reinterpret_cast<const void*>(dwconv_kernel<31, 4, true>), ok, "rows 4, folded norm")) return rc;
    hipLaunchKernelGGL((dwconv_kernel<31, 4, true>), grid, block, 0, s, x, out, wt, bias, B, N, d, len, P, walkers, *norm);
  } else if (TN == 6) {
    static std::atomic<uint64_t> ok{0};
    if (int rc = launchable(reinterpret_cast<const void*>(dwconv_kernel<31, 6, false>), ok, "rows 6")) return rc;
    hipLaunchKernelGGL((dwconv_kernel<31, 6, false>), grid, block, 0, s, x, out, wt, bias, B, N, d, len, P, walkers, none);
  } else {
    static std::atomic<uint64_t> ok{0};
    if (int rc = launchable(reinterpret_cast<const void*>(dwconv_kernel<31, 4, false>), ok, "rows 4")) return rc;
    hipLaunchKernelGGL((dwconv_kernel<31, 4, false>), grid, block, 0, s, x, out, wt, bias, B, N, d, len, P, walkers, none);
  }
  return v2a_check_launch("v2a_dwconv_silu_residual");
}

extern "C" int v2a_dwconv_silu_residual(const float* x, float* out, const float* wt, const float* bias, int32_t B,
                                        int32_t N, int32_t d, int32_t ksize, const int32_t* len, v2a_stream_t stream) {
  return dwconv_launch(x, out, wt, bias, B, N, d, ksize, len, nullptr, stream);
}

extern "C" int v2a_dwconv_silu_residual_norm(const float* x, float* out, const float* wt, const float* bias, int32_t B, int32_t N, int32_t d,
                                             int32_t ksize, const int32_t* len, const v2a_dwconv_norm* norm, v2a_stream_t stream) {
  V2A_REQUIRE(norm != nullptr, "v2a_dwconv_silu_residual_norm: null norm arguments");
  return dwconv_launch(x, out, wt, bias, B, N, d, ksize, len, norm, stream);
}

extern "C" int v2a_rope_inplace(void* qk, int32_t dtype, int64_t rows, int64_t row_stride, int32_t nheads,
                                int32_t rpb, int32_t pos_offset, const float* cs, int32_t layout, v2a_stream_t stream) {
  V2A_REQUIRE(qk && cs, "v2a_rope: null pointer");
  V2A_REQUIRE(rows > 0 && nheads > 0 && rpb > 0, "v2a_rope: rows=%lld nheads=%d rpb=%d", (long long)rows, nheads, rpb);
  V2A_REQUIRE(layout == 0 || layout == 1, "v2a_rope: layout %d", layout);
  const int64_t total = rows * nheads * 8;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == V2A_F32)
    hipLaunchKernelGGL((rope_kernel<float>), grid, block, 0, s, (float*)qk, rows, row_stride, nheads, rpb, pos_offset, cs, layout);
  else if (dtype == V2A_BF16)
    hipLaunchKernelGGL((rope_kernel<bf16_t>), grid, block, 0, s, (bf16_t*)qk, rows, row_stride, nheads, rpb, pos_offset, cs, layout);
  else
    return v2a_fail(V2A_ERR_ARG, "v2a_rope: dtype %d", dtype);
  return v2a_check_launch("v2a_rope_inplace");
}

extern "C" int v2a_linear_small(const float* a, int64_t M, int32_t K, const float* wt, const float* bias,
                                const float* add, int32_t T, float* out, int64_t obs, int32_t row_off, int32_t d,
                                int32_t dup, const float* regs, void* out_bf16, v2a_stream_t stream) {
  V2A_REQUIRE(a && wt && out, "v2a_linear_small: null pointer");
  V2A_REQUIRE(M > 0 && K > 0 && K <= 2048 && T > 0 && d > 0 && d % 4 == 0 && M % T == 0, "v2a_linear_small: M=%lld K=%d T=%d d=%d",
              (long long)M, K, T, d);
  V2A_REQUIRE(obs % 4 == 0 && (((uintptr_t)out | (uintptr_t)wt | (uintptr_t)bias | (uintptr_t)add | (uintptr_t)regs) & 15) == 0 &&
                  ((uintptr_t)out_bf16 & 7) == 0, "v2a_linear_small: 16-byte alignment");
  // rows per block: 8 when that still yields >= 2 blocks per CU, else 2 (one clip: 98 blocks of 8 rows took 46 us per
  // evaluation on 98 CUs; 391 blocks of 2 rows spread the same work over the chip)
  const int rows = (regs ? row_off : 0) + T;
  const int64_t nb8 = (int64_t)((rows + 7) / 8) * (M / T);
  if (nb8 >= 512) {
    dim3 grid((unsigned)((rows + 7) / 8), (unsigned)(M / T)), block(256);
    hipLaunchKernelGGL((linear_small_kernel<8>), grid, block, 2 * 8 * K * sizeof(float), (hipStream_t)stream, a, K, wt, bias, add,
                       T, out, obs, row_off, d, dup, regs, (bf16_t*)out_bf16);
  // (4-row blocks -- 196 blocks, one round at one clip -- measured 27 us against 16.5 us for the 391 blocks of 2 rows: the block time is
  // the latency chain of its K weight-row loads, and more blocks per CU overlap more of it)
  } else {
    dim3 grid((unsigned)((rows + 1) / 2), (unsigned)(M / T)), block(256);
    hipLaunchKernelGGL((linear_small_kernel<2>), grid, block, 2 * 2 * K * sizeof(float), (hipStream_t)stream, a, K, wt, bias, add,
                       T, out, obs, row_off, d, dup, regs, (bf16_t*)out_bf16);
  }
  return v2a_check_launch("v2a_linear_small");
}

extern "C" int v2a_fill_registers(float* out, int64_t obs, const float* regs, int32_t B, int32_t R, int32_t d,
                                  v2a_stream_t stream) {
  V2A_REQUIRE(out && regs && B > 0 && R > 0 && d > 0, "v2a_fill_registers: bad args");
  const int64_t total = (int64_t)B * R * d;
  hipLaunchKernelGGL(fill_registers_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out,
                     obs, regs, B, R, d);
  return v2a_check_launch("v2a_fill_registers");
}

extern "C" int v2a_time_cond(const float* t, int32_t S, const float* fw, const float* wt, const float* bias, float* out,
                             int32_t d, v2a_stream_t stream) {
  V2A_REQUIRE(t && fw && wt && bias && out, "v2a_time_cond: null pointer");
  V2A_REQUIRE(S > 0 && d > 0 && d % 2 == 0, "v2a_time_cond: S=%d d=%d", S, d);
  hipLaunchKernelGGL(time_cond_kernel, dim3(S, (d + 255) / 256), dim3(256), (d + 1) * sizeof(float), (hipStream_t)stream, t, fw,
                     wt, bias, out, d);
  return v2a_check_launch("v2a_time_cond");
}

extern "C" int v2a_apg_reduce(const float* pred, double* apg, int32_t B, int32_t T, int32_t C, int64_t pbs,
                              int32_t row_off, const int32_t* valid_rows, v2a_stream_t stream) {
  V2A_REQUIRE(pred && apg && B > 0 && T > 0 && C > 0, "v2a_apg_reduce: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(apg, 0, sizeof(double) * 2 * B, s);
  if (e != hipSuccess) return v2a_fail(V2A_ERR_LAUNCH, "v2a_apg_reduce memset: %s", hipGetErrorString(e));
  const int64_t per_b = (int64_t)T * C;
  int chunks = (int)((per_b + 256 * 8 - 1) / (256 * 8));
  if (chunks > 64) chunks = 64;
  hipLaunchKernelGGL(apg_reduce_kernel, dim3(chunks, B), dim3(256), 0, s, pred, apg, B, T, C, pbs, row_off, valid_rows);
  return v2a_check_launch("v2a_apg_reduce");
}

extern "C" int v2a_cfg_euler(float* y, const float* pred, int32_t B, int32_t T, int32_t C, int64_t pbs, int32_t row_off,
                             float cfg_strength, const float* dt, const int32_t* step, const double* apg, float keep,
                             v2a_stream_t stream) {
  V2A_REQUIRE(y && pred && dt, "v2a_cfg_euler: null pointer");
  V2A_REQUIRE(B > 0 && T > 0 && C > 0 && C % 4 == 0 && pbs % 4 == 0, "v2a_cfg_euler: B=%d T=%d C=%d", B, T, C);
  const int64_t total4 = (int64_t)B * T * C / 4;
  hipLaunchKernelGGL(cfg_euler_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, pred,
                     B, T, C, pbs, row_off, cfg_strength, dt, step, apg, keep);
  return v2a_check_launch("v2a_cfg_euler");
}

extern "C" int v2a_step_advance(int32_t* step, v2a_stream_t stream) {
  V2A_REQUIRE(step, "v2a_step_advance: null pointer");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
  return v2a_check_launch("v2a_step_advance");
}

extern "C" int v2a_split_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t d, v2a_stream_t stream) {
  V2A_REQUIRE(x && y && rows > 0 && d > 0 && d % 4 == 0, "v2a_split_bf16: bad args (d %% 4 must be 0)");
  V2A_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldy >= 2 * (int64_t)d && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0,
              "v2a_split_bf16: strides / alignment (ldy >= 2*d)");
  const int64_t n4 = rows * (d / 4);
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, (bf16_t*)y, ldy, rows, d);
  return v2a_check_launch("v2a_split_bf16");
}

extern "C" int v2a_cast_bf16(const float* x, void* y, int64_t n, v2a_stream_t stream) {
  V2A_REQUIRE(x && y && n > 0 && n % 4 == 0, "v2a_cast_bf16: bad args (n %% 4 must be 0)");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n / 4);
  return v2a_check_launch("v2a_cast_bf16");
}
