// Probe (tuning aid): do per-stream CU masks (hipExtStreamCreateWithCUMask) partition the MI355X's CUs, how do mask bits map to
// XCDs, and does a kernel captured from a masked stream keep its mask when the hipGraph is replayed?
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/cumask scripts/probes/cumask_probe.hip && /tmp/cumask
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void census(uint32_t* out, int spin) {
  if (threadIdx.x == 0) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 0xF;          // HW_REG_XCC_ID[3:0]
    const uint32_t hwid = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | 4);              // HW_REG_HW_ID (wave, simd, cu, sh, se ...)
    out[blockIdx.x * 2] = xcc;
    out[blockIdx.x * 2 + 1] = hwid;
  }
  // keep the block alive so that blocks spread over every CU the dispatcher may use
  const long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}

static void report(const char* what, const std::vector<uint32_t>& h, int nblk) {
  int per_xcc[16] = {0};
  std::vector<int> seen(16 * 4096, 0);
  int distinct = 0;
  for (int b = 0; b < nblk; ++b) {
    const uint32_t xcc = h[2 * b] & 15, hw = h[2 * b + 1];
    per_xcc[xcc]++;
    const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;    // gfx9 HW_ID: CU_ID[11:8], SH_ID[12], SE_ID[15:13]
    const int key = xcc * 4096 + se * 64 + sh * 16 + cu;
    if (!seen[key]) { seen[key] = 1; ++distinct; }
  }
  printf("%-44s blocks per XCC:", what);
  for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
  printf("   distinct (xcc, se, sh, cu): %d\n", distinct);
}

int main() {
  const int nblk = 2048, spin = 200000;
  uint32_t* d;
  CK(hipMalloc(&d, nblk * 8));
  std::vector<uint32_t> h(nblk * 2);
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  auto run = [&](hipStream_t s, const char* what) -> int {
    CK(hipMemsetAsync(d, 0xFF, nblk * 8, s));
    hipLaunchKernelGGL(census, dim3(nblk), dim3(64), 0, s, d, spin);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, nblk * 8, hipMemcpyDeviceToHost));
    report(what, h, nblk);
    return 0;
  };
  hipStream_t plain;
  CK(hipStreamCreate(&plain));
  if (run(plain, "unmasked stream")) return 1;
  struct M { const char* name; uint32_t w[8]; } masks[] = {
      {"mask words 0-3 all ones (bits 0..127)", {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0}},
      {"mask words 4-7 all ones (bits 128..255)", {0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}},
      {"mask word 0 only (bits 0..31)", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0}},
      {"every even bit", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
      {"bits 0..7 of every word", {0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu, 0xFFu}},
  };
  hipStream_t masked[5];
  for (int i = 0; i < 5; ++i) {
    hipError_t e = hipExtStreamCreateWithCUMask(&masked[i], 8, masks[i].w);
    if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask(%s): %s\n", masks[i].name, hipGetErrorString(e)); return 1; }
    if (run(masked[i], masks[i].name)) return 1;
  }
  // does a captured launch keep the mask?
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(masked[0], hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(census, dim3(nblk), dim3(64), 0, masked[0], d, spin);
  CK(hipStreamEndCapture(masked[0], &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipMemset(d, 0xFF, nblk * 8));
  CK(hipGraphLaunch(ge, masked[0]));
  CK(hipStreamSynchronize(masked[0]));
  CK(hipMemcpy(h.data(), d, nblk * 8, hipMemcpyDeviceToHost));
  report("graph captured on mask 0, replayed on mask 0", h, nblk);
  CK(hipMemset(d, 0xFF, nblk * 8));
  CK(hipGraphLaunch(ge, plain));
  CK(hipStreamSynchronize(plain));
  CK(hipMemcpy(h.data(), d, nblk * 8, hipMemcpyDeviceToHost));
  report("same graph replayed on the unmasked stream", h, nblk);
  // two masked streams at once: do they overlap in time on disjoint CUs?
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, plain));
    CK(hipStreamWaitEvent(masked[0], e0, 0));
    CK(hipStreamWaitEvent(masked[1], e0, 0));
    hipLaunchKernelGGL(census, dim3(1024), dim3(64), 0, masked[0], d, 2000000);
    if (rep == 1) hipLaunchKernelGGL(census, dim3(1024), dim3(64), 0, masked[1], d + 4096, 2000000);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e1, plain));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s: %.3f ms\n", rep == 0 ? "one masked stream (half the chip) alone" : "two masked streams (disjoint halves) together", ms);
  }
  printf("status: ok\n");
  return 0;
}
