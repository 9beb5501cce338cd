// 256x256 bf16 MFMA GEMM for gfx950 with a phase-interleaved K loop:  acc[m][n] = sum_k A[m][k] * W[n][k].
//
// One workgroup = 8 waves (2 along M x 4 along N) = one 256x256 output tile, wave tile 128x64 = four 64x32 quadrants.
// A K tile (64 deep) lives in LDS as four 16 KB half tiles, two LDS buffers (128 KB, one workgroup per CU):
//   AH0 / AH1 : the first / second 64 rows of BOTH wave rows      (local row r -> tile row r + 64*(r >= 64) + 64*s)
//   BH0 / BH1 : the first / second 32 columns of ALL FOUR wave columns (local row r -> tile col 64*(r >> 5) + (r & 31) + 32*s)
// so that every wave reads one A sub-tile (8 x ds_read_b128) or one B sub-tile (4 x ds_read_b128) per phase.  Rows are
// 128 B with the 16-B chunk index XOR-swizzled by (row & 7), applied on the SOURCE address of the LDS-DMA
// (global_load_lds_dwordx4 writes 8 rows x 128 B per wave-instruction, lane-linear) and on the fragment reads.
//
// K loop: 4 phases per K tile, one quadrant (16 x v_mfma_f32_16x16x32_bf16) per phase:
//   phase      reads (this K tile t, buffer b)     LDS-DMA issued                         MFMA quadrant
//   P1         A0 (8), B0 (4)                      --                                     (0,0)
//   P2         B1 (4)                              AH1(t+1) -> b^1                        (0,1)
//   P3         A1 (8)  (over A0's registers)       AH0(t+2) -> b                          (1,1)
//   P4         --                                  BH0(t+2), BH1(t+2) -> b, vmcnt(6)      (1,0)   B0 stays in registers since P1
// (the read-heavy phase issues no DMA and the read-free phase issues two half tiles: 12 / 6 / 10 / 4 LDS-or-DMA instructions
// per phase instead of 14 / 6 / 10 / 2).  Each phase is  {reads, DMA issue} s_barrier {MFMAs} s_barrier.  Every half tile is
// restaged two phases or more after its last fragment read and read one phase or more after the counted vmcnt + barrier that
// retires it, which is what the STAGGERED form needs: the waves of wave row 1 run one barrier behind wave row 0 (they share
// SIMDs pairwise), so on every SIMD one wave multiplies while its partner reads LDS and issues DMA.  vmcnt is never 0 inside
// the loop: three half tiles stay in flight across every barrier.
//
// Dense operands only (no offset tables); up to 3 K-concatenated A segments; split (hi | lo plane) operands as three passes over the
// logical K (GemmParams::s3_kl); epilogues shared with gemm.hip.
#include "gemm_common.h"

namespace {

// probe builds only (scripts/probes/ph8_probe.py): -DV2A_8PH_SKIP=bits drops parts of the steady-state K loop to see which
// pipe bounds it -- 1 W fragment reads, 2 W half-tile DMAs, 4 A fragment reads, 8 A half-tile DMAs (results are then wrong)
#if defined(V2A_GEMM_PROBE) && defined(V2A_8PH_SKIP)
constexpr int kSkip = V2A_8PH_SKIP;
#else
constexpr int kSkip = 0;
#endif

// MODE: 1 = wave rows staggered, 2 = lock-step.  (A variant with TWO phases per K tile -- 32 MFMAs between barriers, the same registers --
// was 2-3 % faster per launch only while its DMAs were issued in the read blocks, where the wave row that runs one barrier behind may
// still have fragment reads of the restaged half tile in flight: safe by timing, not by a barrier.  Issued behind the barriers that make
// it safe by construction it was 10-15 % slower than four phases: profiles/r03_8phase_two_phase_mode.txt.  Four phases stay.)
template <int EPI, typename OutT, int MODE>
__global__ __launch_bounds__(512) void gemm_bf16_8ph_kernel(GemmParams p) {
  constexpr bool STAGGER = MODE != 2;
  constexpr int BM = 256, BN = 256, WM = 128, WN = 64, TM = 8, TN = 4;
  constexpr int HALF = 128 * 128;               // bytes of a half tile
  constexpr int BUF = 4 * HALF;                 // AH0 | AH1 | BH0 | BH1
  static_assert(EPI != V2A_EPI_GEGLU || (TN % 2 == 0), "GEGLU needs value/gate tile pairs");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;      // waves w and w + 4 (same wc) share a SIMD
  const int lr = lane & 15, lq = lane >> 4;

  int tm, tn;
  tile_of_block(p, blockIdx.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- LDS-DMA source geometry: wave w fills 8-row groups w and w + 8 of every half tile
  const int srow = lane >> 3;
  const uint32_t schunk_b = (uint32_t)(((lane & 7) ^ srow) << 4);   // byte offset of the logical chunk this lane fetches
  uint32_t woff[2][2];       // byte offset into W per (half s, group i)
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 8 * (wave + 8 * i) + srow;                 // local row of the half tile
      const int wrw = n0 + 64 * (r >> 5) + (r & 31) + 32 * s;
      woff[s][i] = (uint32_t)(((int64_t)(wrw < p.N ? wrw : p.N - 1) * p.ldw) * 2) + schunk_b;
    }
  const char* wbase = reinterpret_cast<const char*>(p.w);
  const int nk = p.K / 64;

  // segment table as scalar base + deltas: a segment switch is two s_cselect per quantity, not a kernel-argument load.  Deltas rather
  // than a select between the three values themselves: a select of two captured variables becomes a load through a selected address,
  // which keeps the whole closure on the stack (176 B of scratch per lane, measured).
  const int nseg = p.nseg;
  const int64_t a0 = (int64_t)p.a[0];
  const int64_t da1 = nseg > 1 ? (int64_t)p.a[1] - a0 : 0, da2 = nseg > 2 ? (int64_t)p.a[2] - (int64_t)p.a[1] : 0;
  const int32_t ldb0 = (int32_t)(p.lda[0] * 2);
  const int32_t dl1 = nseg > 1 ? (int32_t)(p.lda[1] * 2) - ldb0 : 0, dl2 = nseg > 2 ? (int32_t)((p.lda[2] - p.lda[1]) * 2) : 0;
  const int kend0 = nseg > 1 ? p.kend[0] : 0x7fffffff, kend1 = nseg > 2 ? p.kend[1] : 0x7fffffff;
  const int dk2 = nseg > 2 ? p.kend[1] - p.kend[0] : 0;
  // split operands (GemmParams::s3_kl): K tile kt belongs to pass kt / nkl and is K tile kt % nkl of the logical K; pass 2 reads the lo
  // plane of its A segment (GemmParams::alo further along the row), pass 1 the lo plane of W (s3_kl elements further along the row)
  const int kl = p.s3_kl;
  const int nkl = kl > 0 ? kl / 64 : 0x3fffffff;                 // plain operands: every K tile is in "pass 0"
  const int klog = kl > 0 ? kl : p.K;                            // the logical K
  const int sb0 = kl > 0 ? (int)(2 * p.alo[0]) : 0;              // bytes from a row's hi plane to its lo plane, by segment
  const int dsb1 = kl > 0 && nseg > 1 ? (int)(2 * p.alo[1]) - sb0 : 0;
  const int dsb2 = kl > 0 && nseg > 2 ? (int)(2 * p.alo[2]) - (sb0 + dsb1) : 0;
  const int64_t wlo = (int64_t)kl * 2;
  // Operand streams (round 5).  Each of the three DMA streams -- A half 0, A half 1, W (both halves) -- is issued strictly in K-tile order, so
  // each keeps a RUNNING wave-uniform base pointer stepped by 128 B per K tile and a count of K tiles left in its (pass, segment); per-lane
  // row offsets (row x the segment's row stride) are VGPRs set up per (pass, segment).  Deriving pass, segment, base and row stride of EVERY K
  // tile from kt cost ~45 scalar instructions and two 64-bit multiply-adds per staging call -- ~350 cycles of issue in the load part of a
  // phase whose partner multiplies for 256: the load parts, not the MFMAs, paced the loop (profiles/r05_8phase_even_bands_ab.txt).
  const char* aptr[2];
  int aleft[2] = {0, 0};
  uint32_t aoff[2][2];
  auto a_setup = [&](int s, int kt) {
    const bool p1 = kt >= nkl, p2 = kt >= 2 * nkl;         // p2 implies p1
    const int k0 = (kt - (p1 ? nkl : 0) - (p2 ? nkl : 0)) * 64;
    const bool s1 = k0 >= kend0, s2 = k0 >= kend1;       // s2 implies s1
    const int kbeg = (s1 ? kend0 : 0) + (s2 ? dk2 : 0);
    const int lo = p2 ? sb0 + (s1 ? dsb1 : 0) + (s2 ? dsb2 : 0) : 0;
    aptr[s] = reinterpret_cast<const char*>(a0 + (s1 ? da1 : 0) + (s2 ? da2 : 0)) + (int64_t)(k0 - kbeg) * 2 + lo;
    const uint32_t ldb = (uint32_t)(ldb0 + (s1 ? dl1 : 0) + (s2 ? dl2 : 0));
    const int kstop = min(klog, s2 ? 0x7fffffff : (s1 ? kend1 : kend0));   // end of this segment within the pass
    aleft[s] = (kstop - k0) >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 8 * (wave + 8 * i) + srow;                 // local row of the half tile
      const int ar = m0 + r + ((r >= 64) ? 64 : 0) + 64 * s;
      aoff[s][i] = (uint32_t)(ar < p.M ? ar : p.M - 1) * ldb + schunk_b;
    }
  };
  auto stage_a = [&](int s, int kt, int buf) {
    if (aleft[s] == 0) a_setup(s, kt);
    char* dst = smem_raw + buf * BUF + s * HALF;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(aptr[s] + aoff[s][i]),
                                       (__attribute__((address_space(3))) void*)(dst + (wave + 8 * i) * 1024), 16, 0, 0);
    aptr[s] += 128;
    --aleft[s];
  };
  const char* wptr = wbase;
  int wleft = 0;
  auto stage_b = [&](int kt, int buf) {                    // both halves of W's K tile kt
    if (wleft == 0) {
      const bool p1 = kt >= nkl, p2 = kt >= 2 * nkl;
      const int ktl = kt - (p1 ? nkl : 0) - (p2 ? nkl : 0);
      wptr = wbase + (int64_t)ktl * 128 + (p1 && !p2 ? wlo : 0);
      wleft = min(nkl, nk) - ktl;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      char* dst = smem_raw + buf * BUF + (2 + s) * HALF;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        // the 32-bit row offset is made opaque here: hoisted out of the K loop as a zero-extended 64-bit pair it cost a 64-bit vector add
        // per DMA (and four more registers); as a 32-bit offset beside the scalar base the DMA takes it as is
        uint32_t wo = woff[s][i];
        asm volatile("" : "+v"(wo));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wptr + wo),
                                         (__attribute__((address_space(3))) void*)(dst + (wave + 8 * i) * 1024), 16, 0, 0);
      }
    }
    wptr += 128;
    --wleft;
  };

  // ---- fragment reads: row = sub-tile base + 16*i + lr, chunk = (4*kk + lq) ^ (row & 7) = ... ^ (lr & 7)
  const int a_frag_off = (64 * wr + lr) * 128;
  const int b_frag_off = (32 * wc + lr) * 128;
  const int ch0 = ((0 + lq) ^ (lr & 7)) << 4, ch1 = ((4 + lq) ^ (lr & 7)) << 4;
  bf16x8 af[4][2], bf0[2][2], bf1[2][2];
  auto read_a = [&](int s, int buf) {
    const char* base = smem_raw + buf * BUF + s * HALF + a_frag_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i][0] = *reinterpret_cast<const bf16x8*>(base + i * 2048 + ch0);
      af[i][1] = *reinterpret_cast<const bf16x8*>(base + i * 2048 + ch1);
    }
  };
  auto read_b = [&](int s, int buf, bf16x8 (&bf)[2][2]) {
    const char* base = smem_raw + buf * BUF + (2 + s) * HALF + b_frag_off;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf[j][0] = *reinterpret_cast<const bf16x8*>(base + j * 2048 + ch0);
      bf[j][1] = *reinterpret_cast<const bf16x8*>(base + j * 2048 + ch1);
    }
  };
  // Row bands of the tile that lie wholly behind the last row of the problem are not multiplied (round 5): M = 1564 leaves the seventh
  // 256-row tile band of a one-clip launch with 28 valid rows -- three of its four 64-row bands (six of eight quadrant phases per wave pair) would
  // multiply padding.  The phases keep their barriers, fragment reads and DMAs (the counted vmcnt chain does not change); a skipped quadrant costs no
  // MFMA issue, so such a workgroup's K loop runs at the pace of its operand stream and frees its CU early.  Wave-uniform: wr comes from readfirstlane.
  // (The 32-column bands behind the last column -- N = 3088 leaves the 13th tile column 16 columns wide -- skipped the same way: +-0 at 1 and 8 clips,
  // profiles/r05_bf16x3_8clips_ab.txt item 6; not kept.)
  const int rows_valid = (p.dbg & 512) ? 256 : p.M - m0;       // v2a_tuning.reserved[0] bit 9: multiply every band (A/B)
  const bool need0 = wr * 128 < rows_valid, need1 = wr * 128 + 64 < rows_valid;
  auto quadrant = [&](auto sa_c, auto sb_c, const bf16x8 (&bf)[2][2]) {
    constexpr int SA = decltype(sa_c)::value, SB = decltype(sb_c)::value;
    if (!(SA == 0 ? need0 : need1)) return;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[4 * SA + i][2 * SB + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][kk], bf[j][kk], acc[4 * SA + i][2 * SB + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // folded RMSNorm (consumer): request the tile rows' partial sums ahead of the operand DMAs (gemm_common.h, rowscale_load)
  float* rs_lds = reinterpret_cast<float*>(smem_raw + 2 * BUF);
  const bool scaled = p.rssq != nullptr && p.vec_epi;
  RowScaleLoad rsl;
  if (scaled && tid < 256) rowscale_load(p, m0 + tid, rsl);
  // ---- prologue: K tile 0 whole, plus the three halves of K tile 1 the steady state would have issued already
  stage_a(0, 0, 0);
  stage_b(0, 0);
  stage_a(1, 0, 0);
  if (nk > 1) {
    stage_a(0, 1, 1);
    stage_b(1, 1);
    if (scaled && tid < 256) rs_lds[tid] = rowscale_finish(p, rsl);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    if (scaled && tid < 256) rs_lds[tid] = rowscale_finish(p, rsl);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if constexpr (STAGGER) {
    if (wr == 1) __builtin_amdgcn_s_barrier();     // wave row 1 runs one barrier behind wave row 0 from here on
  }

  auto ktile = [&](auto buf_c, int t) {
    constexpr int B = decltype(buf_c)::value;
    // P1
    if (!(kSkip & 4) || t == 0) read_a(0, B);
    if (!(kSkip & 1) || t == 0) read_b(0, B, bf0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    quadrant(I0{}, I0{}, bf0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // P2
    if (!(kSkip & 1) || t == 0) read_b(1, B, bf1);
    if (t + 1 < nk && !(kSkip & 8)) stage_a(1, t + 1, B ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    quadrant(I0{}, I1{}, bf1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // P3
    if (!(kSkip & 4) || t == 0) read_a(1, B);
    if (t + 2 < nk && !(kSkip & 8)) stage_a(0, t + 2, B);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    quadrant(I1{}, I1{}, bf1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // P4: K tile t+1 (last piece: AH1 from P2) must have landed for every wave before anyone reads it (next P1 and later)
    if (t + 2 < nk) {
      if (!(kSkip & 2)) stage_b(t + 2, B);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    quadrant(I1{}, I0{}, bf0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  for (int t = 0; t < nk; t += 2) {
    ktile(I0{}, t);
    if (t + 1 < nk) ktile(I1{}, t + 1);
  }
  if constexpr (STAGGER) {
    if (wr == 0) __builtin_amdgcn_s_barrier();     // re-align the two wave rows: everyone is out of the K loop after this
  }
  // every DMA was retired by the last P4 wait; the last barrier above orders all fragment reads before the slab writes
  if (p.vec_epi) {
    static_assert(8 * 16 * (WN + 4) * 4 <= 2 * BUF, "epilogue slabs must fit in the ring memory");
    float* tile = reinterpret_cast<float*>(smem_raw) + wave * (16 * (WN + 4));
    EpiPrefetch<EPI, TM, WN, false> pf;
    gemm_epilogue_lds<EPI, OutT, TM, TN, WM, WN, false>(p, acc, tile, m0 + wr * WM, n0 + wc * WN, lane, pf, scaled ? rs_lds + wr * WM : nullptr);
  } else {
    gemm_epilogue<EPI, OutT, TM, TN, WM, WN>(p, acc, m0, n0, wr, wc, lr, lq);
  }
}

template <int EPI, typename OutT>
int launch_8ph(const GemmParams& p_in, hipStream_t s) {
  GemmParams p = p_in;
  v2a_detail::fill_tile_map(p, 256, 256);
  constexpr size_t smem = 2 * 4 * 128 * 128 + 256 * 4;     // two buffers of four half tiles + one row scale per tile row (folded RMSNorm)
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  const int mode = v2a_detail::g_gemm_tuning.use_8phase;
  auto go = [&](auto kern) -> int {
    static std::atomic<uint64_t> lds_set{0};
    if (int rc = v2a_enable_lds(reinterpret_cast<const void*>(kern), smem, lds_set, "v2a_gemm(8-phase)")) return rc;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), smem, s, p);
    return 0;
  };
  int rc;
  if (mode == 2) rc = go(gemm_bf16_8ph_kernel<EPI, OutT, 2>);
  else rc = go(gemm_bf16_8ph_kernel<EPI, OutT, 1>);
  if (rc) return rc;
  return v2a_check_launch("v2a_gemm(8-phase)");
}

}  // namespace

int v2a_detail::launch_gemm_8phase(const GemmParams& p, int epilogue, int out_dtype, hipStream_t s) {
  const bool out_f32 = out_dtype == V2A_F32;
  switch (epilogue) {
    case V2A_EPI_STORE:
      return out_f32 ? launch_8ph<V2A_EPI_STORE, float>(p, s) : launch_8ph<V2A_EPI_STORE, bf16_t>(p, s);
    case V2A_EPI_GEGLU:
      return out_f32 ? launch_8ph<V2A_EPI_GEGLU, float>(p, s) : launch_8ph<V2A_EPI_GEGLU, bf16_t>(p, s);
    case V2A_EPI_RESID:
      if (out_f32) return launch_8ph<V2A_EPI_RESID, float>(p, s);
      break;
    case V2A_EPI_GATE_RESID:
      if (out_f32) return launch_8ph<V2A_EPI_GATE_RESID, float>(p, s);
      break;
  }
  return v2a_fail(V2A_ERR_ARG, "v2a_gemm(8-phase): unsupported epilogue %d / out_dtype %d", epilogue, out_dtype);
}
