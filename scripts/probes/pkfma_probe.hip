// Why does v2a_linear_small give wrong results in lanes 48..63 while ANOTHER process runs MFMA kernels on the same GPU?  (debug aid)
// Three forms of the same inner loop (2 rows x 4 columns per lane, K = 32, d = 256, one active wave per block):
//   0  the product kernel's source form (compiler emits v_pk_fma_f32 + partial s_waitcnt vmcnt(N) / lgkmcnt(N))
//   1  the same with every FMA as an explicit v_fmac_f32 (no packed math)
//   2  the source form with a full s_waitcnt vmcnt(0) lgkmcnt(0) between the loads of a k group and its FMAs
//   3  explicit v_pk_fma_f32 on a materialised (av, av) register pair: no op_sel modifiers
//   4  explicit v_pk_fma_f32 with op_sel_hi:[1,0,1] (src1's LOW half broadcast to both results), as the compiler emits it
//   5  explicit v_pk_fma_f32 with op_sel:[0,1,0] (src1's HIGH half broadcast), the compiler's other form
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/pkfma scripts/probes/pkfma_probe.hip && /tmp/pkfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <unistd.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FORM>
__global__ __launch_bounds__(256) void lin(const float* __restrict__ a, int K, const float* __restrict__ wt, int T, float* __restrict__ out, int d) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* as = reinterpret_cast<float*>(smem_raw);
  const int rr0 = blockIdx.x * 2;
  for (int i = threadIdx.x; i < 2 * K; i += blockDim.x) {
    const int t = rr0 + i / K;
    as[i] = t < T ? a[(int64_t)t * K + (i % K)] : 0.f;
  }
  __syncthreads();
  for (int n = threadIdx.x * 4; n < d; n += blockDim.x * 4) {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if constexpr (FORM == 2) {
      for (int k0 = 0; k0 < K; k0 += 8) {
        f32x4 w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const f32x4*>(wt + (int64_t)(k0 + j) * d + n);
        float av[2][8];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int j = 0; j < 8; ++j) av[r][j] = as[r * K + k0 + j];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][e] += av[r][j] * w[j][e];
      }
    } else {
#pragma unroll 8
      for (int k = 0; k < K; ++k) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(wt + (int64_t)k * d + n);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const float av = as[r * K + k];
          if constexpr (FORM >= 3) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              f32x2 x = {acc[r][2 * h], acc[r][2 * h + 1]};
              const f32x2 ww = {w[2 * h], w[2 * h + 1]};
              if constexpr (FORM == 3) {
                const f32x2 aa = {av, av};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(ww), "v"(aa));
              } else if constexpr (FORM == 4) {
                const f32x2 aa = {av, 0.f};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(x) : "v"(ww), "v"(aa));
              } else {
                const f32x2 aa = {0.f, av};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(x) : "v"(ww), "v"(aa));
              }
              acc[r][2 * h] = x[0];
              acc[r][2 * h + 1] = x[1];
            }
          } else if constexpr (FORM == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = acc[r][e];
              asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(av), "v"(w[e]));
              acc[r][e] = x;
            }
          } else {
            acc[r][0] += av * w[0];
            acc[r][1] += av * w[1];
            acc[r][2] += av * w[2];
            acc[r][3] += av * w[3];
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (rr0 + r < T) *reinterpret_cast<f32x4*>(out + (int64_t)(rr0 + r) * d + n) = acc[r];
  }
}

int main(int argc, char** argv) {
  const int T = 600, K = 32, d = 256, reps = argc > 1 ? atoi(argv[1]) : 300;
  std::vector<float> ha(T * K), hw(K * d), ref(T * d), got(T * d);
  srand(1);
  for (auto& v : ha) v = (float)rand() / RAND_MAX * 2 - 1;
  for (auto& v : hw) v = (float)rand() / RAND_MAX * 2 - 1;
  for (int t = 0; t < T; ++t)
    for (int n = 0; n < d; ++n) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)ha[t * K + k] * hw[k * d + n];
      ref[t * d + n] = (float)s;
    }
  float *da, *dw, *dout;
  CK(hipMalloc(&da, ha.size() * 4));
  CK(hipMalloc(&dw, hw.size() * 4));
  CK(hipMalloc(&dout, ref.size() * 4));
  CK(hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  auto run = [&](int form) -> int {
    CK(hipMemset(dout, 0, ref.size() * 4));
    dim3 grid(T / 2), block(256);
    if (form == 0) hipLaunchKernelGGL(lin<0>, grid, block, 2 * K * 4, 0, da, K, dw, T, dout, d);
    else if (form == 1) hipLaunchKernelGGL(lin<1>, grid, block, 2 * K * 4, 0, da, K, dw, T, dout, d);
    else if (form == 2) hipLaunchKernelGGL(lin<2>, grid, block, 2 * K * 4, 0, da, K, dw, T, dout, d);
    else if (form == 3) hipLaunchKernelGGL(lin<3>, grid, block, 2 * K * 4, 0, da, K, dw, T, dout, d);
    else if (form == 4) hipLaunchKernelGGL(lin<4>, grid, block, 2 * K * 4, 0, da, K, dw, T, dout, d);
    else hipLaunchKernelGGL(lin<5>, grid, block, 2 * K * 4, 0, da, K, dw, T, dout, d);
    CK(hipMemcpy(got.data(), dout, ref.size() * 4, hipMemcpyDeviceToHost));
    return 0;
  };
  auto count_bad = [&](int form, int n, int* lane_lo, int* lane_hi) {
    int bad = 0;
    *lane_lo = 999; *lane_hi = -1;
    for (int it = 0; it < n; ++it) {
      if (run(form)) return -1;
      bool b = false;
      for (int i = 0; i < T * d; ++i)
        if (fabsf(got[i] - ref[i]) > 1e-3f) {
          b = true;
          const int lane = (i % d) / 4;
          if (lane < *lane_lo) *lane_lo = lane;
          if (lane > *lane_hi) *lane_hi = lane;
        }
      bad += b;
    }
    return bad;
  };
  int lo, hi;
  for (int f = 0; f < 6; ++f) printf("quiet device, form %d: %d of 50 launches wrong\n", f, count_bad(f, 50, &lo, &hi));
  fflush(stdout);
  if (argc > 2) {            // argv[2] = shell command that starts the loading process in the background
    if (system(argv[2]) != 0) printf("could not start the load\n");
    sleep(8);
  }
  for (int pass = 0; pass < 2; ++pass)
    for (int f = 0; f < 6; ++f) {
      const int bad = count_bad(f, reps, &lo, &hi);
      printf("beside the other process, form %d: %d of %d launches wrong; lanes %d..%d\n", f, bad, reps, bad ? lo : -1, bad ? hi : -1);
      fflush(stdout);
    }
  return 0;
}
