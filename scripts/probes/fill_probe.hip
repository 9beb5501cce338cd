// Microbenchmark (tuning aid, not part of the library): how fast can a CU pull an L2-resident bf16 panel
//   (a) into LDS with global_load_lds_dwordx4 (the GEMM's staging path), (b) into VGPRs with global_load_dwordx4 ?
// Every workgroup streams `iters` tiles of ROWS x 128 B (one K tile of a 64-wide bf16 row block) from a panel of
// `rows` x `K` bf16 that all workgroups share (like the A operand of the M = 1564 GEMMs).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/fill_probe scripts/probes/fill_probe.hip ; run: /tmp/fill_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ROWS, int DEPTH>
__global__ __launch_bounds__(256) void fill_lds(const char* __restrict__ panel, int rows, int K, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int G = ROWS / 8;               // 8-row DMA groups per tile
  constexpr int LPW = G / 4;                // per wave
  const int nk = K / 64;
  const int band = (blockIdx.x * 7) % ((rows + ROWS - 1) / ROWS);
  uint32_t goff[LPW];
  for (int i = 0; i < LPW; ++i) {
    int r = band * ROWS + (wave + i * 4) * 8 + (lane >> 3);
    r = r < rows ? r : rows - 1;
    goff[i] = (uint32_t)(((int64_t)r * K + (((lane & 7) ^ (lane >> 3)) << 3)) * 2);
  }
  for (int it = 0; it < iters; ++it) {
    const char* base = panel + (int64_t)(it % nk) * 128;
    char* st = smem + (it % DEPTH) * ROWS * 128;
    for (int i = 0; i < LPW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + goff[i]),
                                       (__attribute__((address_space(3))) void*)(st + (wave + i * 4) * 1024), 16, 0, 0);
    if (it >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * LPW) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && sink) sink[blockIdx.x] = reinterpret_cast<float*>(smem)[blockIdx.x & 63];
}

// the GEMM's K-loop skeleton: counted wait, workgroup barrier, next issue, then `MF` dependent-free MFMAs per wave as the compute phase
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int ROWS, int DEPTH, int MF, bool SPREAD>
__global__ __launch_bounds__(256) void fill_bar(const char* __restrict__ panel, int rows, int K, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int G = ROWS / 8;
  constexpr int LPW = G / 4;
  const int nk = K / 64;
  const int band = (blockIdx.x * 7) % ((rows + ROWS - 1) / ROWS);
  uint32_t goff[LPW];
  for (int i = 0; i < LPW; ++i) {
    int r = band * ROWS + (wave + i * 4) * 8 + (lane >> 3);
    r = r < rows ? r : rows - 1;
    goff[i] = (uint32_t)(((int64_t)r * K + (((lane & 7) ^ (lane >> 3)) << 3)) * 2);
  }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  bf16x8_t a = {}, b = {};
  auto issue = [&](int it) {
    const char* base = panel + (int64_t)(it % nk) * 128;
    char* st = smem + (it % DEPTH) * ROWS * 128;
    for (int i = 0; i < LPW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + goff[i]),
                                       (__attribute__((address_space(3))) void*)(st + (wave + i * 4) * 1024), 16, 0, 0);
  };
  for (int d = 0; d < DEPTH - 1; ++d) issue(d);
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * LPW) : "memory");
    __builtin_amdgcn_s_barrier();
    if (!SPREAD || MF == 0) {
      issue(it + DEPTH - 1);
#pragma unroll
      for (int m = 0; m < MF; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
    } else {
      // the same DMA instructions, one every MF / LPW MFMAs instead of all up front
      const char* base = panel + (int64_t)((it + DEPTH - 1) % nk) * 128;
      char* st = smem + ((it + DEPTH - 1) % DEPTH) * ROWS * 128;
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        if (m % (MF / LPW) == 0 && m / (MF / LPW) < LPW) {
          const int i = m / (MF / LPW);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + goff[i]),
                                           (__attribute__((address_space(3))) void*)(st + (wave + i * 4) * 1024), 16, 0, 0);
        }
        acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 123.456f) sink[blockIdx.x] = reinterpret_cast<float*>(smem)[lane];
}

template <int ROWS, int DEPTH>
__global__ __launch_bounds__(256) void fill_vgpr(const char* __restrict__ panel, int rows, int K, int iters, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int G = ROWS / 8;
  constexpr int LPW = G / 4;
  const int nk = K / 64;
  const int band = (blockIdx.x * 7) % ((rows + ROWS - 1) / ROWS);
  uint32_t goff[LPW];
  for (int i = 0; i < LPW; ++i) {
    int r = band * ROWS + (wave + i * 4) * 8 + (lane >> 3);
    r = r < rows ? r : rows - 1;
    goff[i] = (uint32_t)(((int64_t)r * K + ((lane & 7) << 3)) * 2);
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 ring[DEPTH][LPW];
  for (int it = 0; it < iters + DEPTH - 1; ++it) {
    if (it >= DEPTH - 1) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
        if ((it - (DEPTH - 1)) % DEPTH == d)
#pragma unroll
          for (int i = 0; i < LPW; ++i) acc += ring[d][i];
    }
    if (it < iters) {
      const char* base = panel + (int64_t)(it % nk) * 128;
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
        if (it % DEPTH == d)
#pragma unroll
          for (int i = 0; i < LPW; ++i) ring[d][i] = *reinterpret_cast<const f32x4*>(base + goff[i]);
    }
  }
  if (sink && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[blockIdx.x] = acc[0];
}

template <typename F> float time_ms(F launch, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int rows = 1564, K = argc > 1 ? atoi(argv[1]) : 1024, iters = 512;
  printf("panel %d x %d bf16 = %.1f MB\n", rows, K, rows * (double)K * 2 / 1e6);
  char* panel; float* sink;
  hipMalloc(&panel, (size_t)rows * K * 2 + 4096);
  hipMemset(panel, 0, (size_t)rows * K * 2 + 4096);
  hipMalloc(&sink, 1 << 16);
  for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
    const int grid = 256 * wgs_per_cu;
#define RUN(KERN, ROWS, DEPTH, SMEM)                                                                                      \
    {                                                                                                                    \
      hipFuncSetAttribute(reinterpret_cast<const void*>(KERN<ROWS, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); \
      float ms = time_ms([&] { hipLaunchKernelGGL((KERN<ROWS, DEPTH>), dim3(grid), dim3(256), SMEM, 0, panel, rows, K, iters, sink); }, 5); \
      double bytes = (double)grid * iters * ROWS * 128;                                                                  \
      printf("%-10s rows/tile %3d depth %d  %d WG/CU: %7.1f us  %6.2f TB/s chip  %6.1f GB/s per CU\\n", #KERN, ROWS, DEPTH, wgs_per_cu, \
             ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);                                                          \
    }
    RUN(fill_lds, 128, 3, 3 * 128 * 128)
    RUN(fill_lds, 256, 3, 3 * 256 * 128)
#define RUNB(ROWS, DEPTH, MF, SMEM, SPREAD)                                                                                      \
    {                                                                                                                    \
      hipFuncSetAttribute(reinterpret_cast<const void*>(fill_bar<ROWS, DEPTH, MF, SPREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); \
      float ms = time_ms([&] { hipLaunchKernelGGL((fill_bar<ROWS, DEPTH, MF, SPREAD>), dim3(grid), dim3(256), SMEM, 0, panel, rows, K, iters, sink); }, 5); \
      double bytes = (double)grid * iters * ROWS * 128;                                                                  \
      printf("fill_bar%s rows/tile %3d depth %d mfma/wave %2d  %d WG/CU: %7.1f us  %6.2f TB/s chip  %6.1f GB/s per CU  (%.0f clk per tile at 2.4 GHz)\\n", SPREAD ? "(spread)" : "        ", ROWS, DEPTH, MF, wgs_per_cu, \
             ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256, ms * 1e-3 / iters / wgs_per_cu * 2.4e9);                   \
    }
    RUNB(128, 3, 32, 3 * 128 * 128, false)
    RUNB(128, 3, 32, 3 * 128 * 128, true)
    RUNB(256, 3, 32, 3 * 256 * 128, false)
    RUNB(256, 3, 32, 3 * 256 * 128, true)
    RUNB(256, 3, 64, 3 * 256 * 128, false)
    RUNB(256, 3, 64, 3 * 256 * 128, true)
    RUN(fill_vgpr, 128, 3, 0)
    RUN(fill_vgpr, 256, 3, 0)
    RUN(fill_vgpr, 256, 6, 0)
  }
  hipError_t e = hipDeviceSynchronize();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
