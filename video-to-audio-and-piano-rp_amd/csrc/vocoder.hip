// Encodec (24 kHz SEANet) decoder -- the vocoder behind `EncodecWrapper.decode` (x3:434-437, predict.py:277-278), SURVEY 8f N1.
// Every convolution of the stack is a v2a_gemm call on a time-major [T][C] buffer (overlapping rows: lda = C, K = k*C;
// transposed convolutions: K = 2C, N = stride*Cout), so only two kernels live here:
//   * elu_pad:    out[pad + t] = ELU(x[t]) with `pad` reflected (causal Conv1d) or zero (ConvTranspose1d) rows in front;
//   * lstm_layer / lstm2: the recurrence of one nn.LSTM layer, or of both layers one step apart, as a persistent kernel --
//                 the recurrent weights stay in registers across all T steps, the H/8 workgroups exchange h_t through tagged
//                 64-bit words in global memory (no barrier, no fences).
#include "v2a_common.h"

namespace {

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : expm1f(x); }

// one thread per float4 of the padded output
__global__ __launch_bounds__(256) void elu_pad_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t T, int C4, int pad,
                                                      int reflect, int act) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (T + pad) * C4) return;
  const int64_t row = gid / C4;
  const int c4 = (int)(gid - row * C4);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  int64_t src = row - pad;
  if (src < 0) src = reflect ? -src : -1;            // reflect: padded row (pad - i) mirrors row i, the edge row is not repeated
  if (src >= 0) {
    v = *reinterpret_cast<const f32x4*>(x + (src * C4 + c4) * 4);
    if (act) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = elu_f(v[e]);
    }
  }
  *reinterpret_cast<f32x4*>(out + gid * 4) = v;
}

// ---- LSTM recurrence -------------------------------------------------------------------------------
// gx[t][4H] = W_ih x_t + b_ih + b_hh (a GEMM done by the caller); gates in torch order (i, f, g, o).
// Workgroup w owns hidden units [8w, 8w+8): 32 rows of W_hh (4 gates x 8 units), thread (row = tid >> 3, seg = tid & 7) keeps
// the H/8 weights of its row segment in registers for the whole sequence.  Per step: h_{t-1} (H floats, written by all
// workgroups) -> LDS, 64 FMAs per thread, 8-lane reduction, cell update by 8 threads, h_t published.
// Exchange without a barrier: every h value travels as one 64-bit word {h, tag = t + 1} written with a device-scope atomic
// store into a double-buffered table; a consumer polls the words it needs until their tag says "step t", so a step costs one
// store -> load round trip through the memory-side cache and no fences (the tag and the value arrive together).  Two
// buffers suffice: a workgroup can only run two steps ahead of a peer after that peer has read the older buffer.
constexpr int LSTM_H = 512;
constexpr int LSTM_SEG = LSTM_H / 8;       // 64 columns per thread
constexpr int LSTM_SPIN_CAP = 1 << 21;     // polls before a workgroup gives up (a missing peer must not hang the GPU)

__global__ __launch_bounds__(256) void lstm_layer_kernel(const float* __restrict__ gx, const float* __restrict__ whh,
                                                         float* __restrict__ hout, const float* __restrict__ resid,
                                                         float* __restrict__ y, int T, unsigned long long* ex /* [2][H] */, int* err) {
  __shared__ float hs[8 * (LSTM_SEG + 1)];   // h_{t-1}, segment-major with one pad float per segment (bank spread)
  __shared__ float gs[32];
  __shared__ int dead;
  const int tid = threadIdx.x;
  const int row = tid >> 3, seg = tid & 7;
  const int gate = row >> 3, unit = row & 7;
  const int u0 = blockIdx.x * 8;
  float w[LSTM_SEG];
  {
    const float* wr = whh + (int64_t)(gate * LSTM_H + u0 + unit) * LSTM_H + seg * LSTM_SEG;
#pragma unroll
    for (int j = 0; j < LSTM_SEG; j += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(wr + j);
      w[j] = v[0]; w[j + 1] = v[1]; w[j + 2] = v[2]; w[j + 3] = v[3];
    }
  }
  if (tid == 0) dead = 0;
  float c = 0.f;                             // cell state of unit tid (threads 0..7)
  for (int t = 0; t < T; ++t) {
    // operands that do not depend on h: issued first so their latency hides behind the wait for h_{t-1}
    float gxv = 0.f, rv = 0.f;
    if (seg == 0) gxv = gx[(int64_t)t * 4 * LSTM_H + gate * LSTM_H + u0 + unit];
    if (tid < 8 && resid) rv = resid[(int64_t)t * LSTM_H + u0 + tid];
    // h_{t-1} -> LDS (zeros at t = 0): thread j < H/4 polls its 4 words of buffer (t-1) & 1 until they carry tag t
    if (tid < LSTM_H / 4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (t > 0) {
        const unsigned long long* src = ex + (size_t)((t - 1) & 1) * LSTM_H + tid * 4;
        int spins = 0;
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned long long q = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[e] = __uint_as_float((unsigned)q);
            ok = ok && (int)(q >> 32) == t;
          }
          if (ok) break;
          if (++spins > LSTM_SPIN_CAP || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dead = 1;
            break;
          }
        }
      }
      const int col = tid * 4, sg = col / LSTM_SEG, j = col % LSTM_SEG;
      float* d = hs + sg * (LSTM_SEG + 1) + j;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    if (dead) return;                        // a peer never published: every workgroup leaves, the host reports the error flag
    float acc = 0.f;
    const float* hp = hs + seg * (LSTM_SEG + 1);
#pragma unroll
    for (int j = 0; j < LSTM_SEG; ++j) acc = fmaf(w[j], hp[j], acc);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (seg == 0) gs[row] = acc + gxv;
    __syncthreads();
    if (tid < 8) {
      const float gi = 1.f / (1.f + expf(-gs[tid])), gf = 1.f / (1.f + expf(-gs[8 + tid]));
      const float gg = tanhf(gs[16 + tid]), go = 1.f / (1.f + expf(-gs[24 + tid]));
      c = gf * c + gi * gg;
      const float h = go * tanhf(c);
      // publish {h_t, tag t + 1} first (the critical path of every peer), then the plain outputs
      const unsigned long long word = ((unsigned long long)(unsigned)(t + 1) << 32) | (unsigned long long)__float_as_uint(h);
      __hip_atomic_store(ex + (size_t)(t & 1) * LSTM_H + u0 + tid, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int64_t o = (int64_t)t * LSTM_H + u0 + tid;
      hout[o] = h;
      if (y) y[o] = h + rv;
    }
    // no barrier here: gs / hs are rewritten only after the next step's __syncthreads pair
  }
}

// ---- both LSTM layers in one pass ---------------------------------------------------------------------------------
// Pipeline step s computes layer 0 at time s (from h0[s-1]) and layer 1 at time s-1 (from x1 = h0[s-1] and h1[s-2]): both
// inputs were published during step s-1, so ONE store -> load round trip per step serves both layers and the sequence takes
// T + 1 steps instead of 2T.  A workgroup owns the same 8 hidden units in both layers; a thread keeps its 64-column row
// segments of W_hh0, W_ih1 and W_hh1 (192 floats) in registers.  Layer 1's input projection happens here, so its gates_x GEMM
// disappears.  Exchange tables as above, one double-buffered table per layer.
__global__ __launch_bounds__(256) void lstm2_kernel(const float* __restrict__ gx0, const float* __restrict__ whh0,
                                                    const float* __restrict__ wih1, const float* __restrict__ b1,
                                                    const float* __restrict__ whh1, const float* __restrict__ resid,
                                                    float* __restrict__ y, int T, unsigned long long* ex /* [2 layers][2][H] */, int* err) {
  __shared__ float hs0[8 * (LSTM_SEG + 1)];
  __shared__ float hs1[8 * (LSTM_SEG + 1)];
  __shared__ float gs[64];
  __shared__ int dead;
  const int tid = threadIdx.x;
  const int row = tid >> 3, seg = tid & 7;
  const int gate = row >> 3, unit = row & 7;
  const int u0 = blockIdx.x * 8;
  float w0[LSTM_SEG], wi[LSTM_SEG], w1[LSTM_SEG];
  {
    const int64_t off = (int64_t)(gate * LSTM_H + u0 + unit) * LSTM_H + seg * LSTM_SEG;
#pragma unroll
    for (int j = 0; j < LSTM_SEG; j += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(whh0 + off + j);
      const f32x4 b = *reinterpret_cast<const f32x4*>(wih1 + off + j);
      const f32x4 c = *reinterpret_cast<const f32x4*>(whh1 + off + j);
#pragma unroll
      for (int e = 0; e < 4; ++e) { w0[j + e] = a[e]; wi[j + e] = b[e]; w1[j + e] = c[e]; }
    }
  }
  const float bias1 = b1[gate * LSTM_H + u0 + unit];
  if (tid == 0) dead = 0;
  float c0 = 0.f, c1 = 0.f;                  // cell states of unit tid & 7: layer 0 in threads 0..7, layer 1 in threads 8..15
  unsigned long long* ex0 = ex;
  unsigned long long* ex1 = ex + 2 * LSTM_H;
  for (int s = 0; s <= T; ++s) {
    float gxv = 0.f, rv = 0.f;
    if (seg == 0 && s < T) gxv = gx0[(int64_t)s * 4 * LSTM_H + gate * LSTM_H + u0 + unit];
    if (tid >= 8 && tid < 16 && resid && s >= 1) rv = resid[(int64_t)(s - 1) * LSTM_H + u0 + (tid - 8)];
    // h0[s-1] (tag s) and h1[s-2] (tag s-1) -> LDS; threads 0..127 poll layer 0's table, 128..255 layer 1's
    {
      const int half = tid >> 7, t4 = tid & 127;
      const int want = half == 0 ? s : s - 1;                    // tag = time index + 1
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (want >= 1) {
        const unsigned long long* src = (half == 0 ? ex0 : ex1) + (size_t)((want - 1) & 1) * LSTM_H + t4 * 4;
        int spins = 0;
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned long long q = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[e] = __uint_as_float((unsigned)q);
            ok = ok && (int)(q >> 32) == want;
          }
          if (ok) break;
          if (++spins > LSTM_SPIN_CAP || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dead = 1;
            break;
          }
        }
      }
      const int col = t4 * 4, sg = col / LSTM_SEG, j = col % LSTM_SEG;
      float* d = (half == 0 ? hs0 : hs1) + sg * (LSTM_SEG + 1) + j;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    if (dead) return;
    const float* h0p = hs0 + seg * (LSTM_SEG + 1);
    const float* h1p = hs1 + seg * (LSTM_SEG + 1);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int j = 0; j < LSTM_SEG; ++j) {
      const float h0v = h0p[j];
      a0 = fmaf(w0[j], h0v, a0);
      a1 = fmaf(wi[j], h0v, a1);
      a1 = fmaf(w1[j], h1p[j], a1);
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      a0 += __shfl_xor(a0, o, 64);
      a1 += __shfl_xor(a1, o, 64);
    }
    if (seg == 0) {
      gs[row] = a0 + gxv;
      gs[32 + row] = a1 + bias1;
    }
    __syncthreads();
    if (tid < 8 && s < T) {                                      // layer 0, time s
      const float gi = 1.f / (1.f + expf(-gs[tid])), gf = 1.f / (1.f + expf(-gs[8 + tid]));
      const float gg = tanhf(gs[16 + tid]), go = 1.f / (1.f + expf(-gs[24 + tid]));
      c0 = gf * c0 + gi * gg;
      const float h = go * tanhf(c0);
      const unsigned long long word = ((unsigned long long)(unsigned)(s + 1) << 32) | (unsigned long long)__float_as_uint(h);
      __hip_atomic_store(ex0 + (size_t)(s & 1) * LSTM_H + u0 + tid, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (tid >= 8 && tid < 16 && s >= 1) {                 // layer 1, time s - 1
      const int u = tid - 8;
      const float gi = 1.f / (1.f + expf(-gs[32 + u])), gf = 1.f / (1.f + expf(-gs[40 + u]));
      const float gg = tanhf(gs[48 + u]), go = 1.f / (1.f + expf(-gs[56 + u]));
      c1 = gf * c1 + gi * gg;
      const float h = go * tanhf(c1);
      const unsigned long long word = ((unsigned long long)(unsigned)s << 32) | (unsigned long long)__float_as_uint(h);
      __hip_atomic_store(ex1 + (size_t)((s - 1) & 1) * LSTM_H + u0 + u, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      y[(int64_t)(s - 1) * LSTM_H + u0 + u] = h + rv;
    }
  }
}

}  // namespace

extern "C" int v2a_elu_pad(const float* x, float* out, int64_t T, int32_t C, int32_t pad, int32_t reflect, int32_t act,
                           v2a_stream_t stream) {
  V2A_REQUIRE(x && out && x != out, "v2a_elu_pad: null / aliased pointer");
  V2A_REQUIRE(T > 0 && C > 0 && C % 4 == 0 && pad >= 0 && (!reflect || pad < T), "v2a_elu_pad: T=%lld C=%d pad=%d", (long long)T, C, pad);
  V2A_REQUIRE((((uintptr_t)x | (uintptr_t)out) & 15) == 0, "v2a_elu_pad: 16-byte alignment");
  const int64_t total = (T + pad) * (C / 4);
  hipLaunchKernelGGL(elu_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, T, C / 4, pad,
                     reflect, act);
  return v2a_check_launch("v2a_elu_pad");
}

extern "C" int v2a_lstm_layer(const float* gates_x, const float* w_hh, float* h, const float* resid, float* y, int32_t T, int32_t H,
                              int32_t* workspace, v2a_stream_t stream) {
  V2A_REQUIRE(gates_x && w_hh && h && workspace, "v2a_lstm_layer: null pointer");
  V2A_REQUIRE(H == LSTM_H, "v2a_lstm_layer: hidden size %d (built for %d: encodec_24khz, 16 * num_filters)", H, LSTM_H);
  V2A_REQUIRE(T > 0 && (resid == nullptr || y != nullptr), "v2a_lstm_layer: T=%d / resid without y", T);
  V2A_REQUIRE((((uintptr_t)gates_x | (uintptr_t)w_hh | (uintptr_t)h) & 15) == 0 && ((uintptr_t)workspace & 7) == 0,
              "v2a_lstm_layer: alignment (16 bytes for tensors, 8 for the workspace)");
  hipStream_t s = (hipStream_t)stream;
  // workspace: [0, 4H) int32 = the two {h, tag} exchange tables (tags must start at 0), [4H] = error flag
  hipError_t e = hipMemsetAsync(workspace, 0, (4 * (size_t)H + 2) * sizeof(int32_t), s);
  if (e != hipSuccess) return v2a_fail(V2A_ERR_LAUNCH, "v2a_lstm_layer: memset: %s", hipGetErrorString(e));
  // H/8 = 64 workgroups of 256 threads: co-resident on any free MI355X (256 CUs); a workgroup whose peers never publish
  // gives up after LSTM_SPIN_CAP polls and raises the error flag
  hipLaunchKernelGGL(lstm_layer_kernel, dim3(H / 8), dim3(256), 0, s, gates_x, w_hh, h, resid, y, T,
                     reinterpret_cast<unsigned long long*>(workspace), workspace + 4 * H);
  return v2a_check_launch("v2a_lstm_layer");
}

extern "C" int v2a_lstm2(const float* gates_x0, const float* w_hh0, const float* w_ih1, const float* bias1, const float* w_hh1,
                         const float* resid, float* y, int32_t T, int32_t H, int32_t* workspace, v2a_stream_t stream) {
  V2A_REQUIRE(gates_x0 && w_hh0 && w_ih1 && bias1 && w_hh1 && y && workspace, "v2a_lstm2: null pointer");
  V2A_REQUIRE(H == LSTM_H, "v2a_lstm2: hidden size %d (built for %d: encodec_24khz, 16 * num_filters)", H, LSTM_H);
  V2A_REQUIRE(T > 0, "v2a_lstm2: T=%d", T);
  V2A_REQUIRE((((uintptr_t)gates_x0 | (uintptr_t)w_hh0 | (uintptr_t)w_ih1 | (uintptr_t)w_hh1) & 15) == 0 && ((uintptr_t)workspace & 7) == 0,
              "v2a_lstm2: alignment (16 bytes for tensors, 8 for the workspace)");
  hipStream_t s = (hipStream_t)stream;
  // workspace: [0, 8H) int32 = two double-buffered {h, tag} exchange tables (tags must start at 0), [8H] = error flag
  hipError_t e = hipMemsetAsync(workspace, 0, (8 * (size_t)H + 2) * sizeof(int32_t), s);
  if (e != hipSuccess) return v2a_fail(V2A_ERR_LAUNCH, "v2a_lstm2: memset: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(lstm2_kernel, dim3(H / 8), dim3(256), 0, s, gates_x0, w_hh0, w_ih1, bias1, w_hh1, resid, y, T,
                     reinterpret_cast<unsigned long long*>(workspace), workspace + 8 * H);
  return v2a_check_launch("v2a_lstm2");
}
