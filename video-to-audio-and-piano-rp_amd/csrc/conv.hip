// Video2Roll frame encoder: the memory-bound kernels around the conv-as-GEMM path.
//   im2col (NHWC and 5-frame-window variants), max / average pooling, the fused FRB + softmax + fc head and the
//   x3 temporal expansion of the roll.  The convolutions themselves are v2a_gemm calls on the patch matrix
//   (eval-mode BatchNorm folded into weight / bias, ReLU and residual in the GEMM epilogue).
// Reference: src/audeo/Video2RollNet.py:127-251 (`v2r`), E2TTS.encode_frames x3:1525-1553.
#include "v2a_common.h"

namespace {

template <typename OutT> __device__ __forceinline__ void store4(OutT* dst, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* dst, f32x4 v) { *reinterpret_cast<f32x4*>(dst) = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* dst, f32x4 v) {
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
  *reinterpret_cast<bf16x4*>(dst) = o;
}

// ---- im2col, NHWC source --------------------------------------------------------------------
// One thread per 4 consecutive k of one patch row: k = (ky*kw + kx)*C + c with C % 4 == 0, so the 4 values are one
// aligned float4 of the source pixel and consecutive lanes read / write consecutive addresses.
// Traffic: 4*K B read (L2: every input pixel is re-read kh*kw/stride^2 times) + sizeof(OutT)*Kpad B written per row.
template <typename OutT>
__global__ __launch_bounds__(256) void im2col_nhwc_kernel(const float* __restrict__ x, OutT* __restrict__ col, int64_t rows,
                                                          int H, int W, int C, int kw, int K, int K4pad, int stride, int pad,
                                                          int Ho, int Wo, int64_t ldo) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= rows * K4pad) return;
  const int64_t m = gid / K4pad;
  const int k = (int)(gid - m * K4pad) * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (k < K) {
    const int tap = k / C, c = k - tap * C;
    const int ky = tap / kw, kx = tap - ky * kw;
    const int xo = (int)(m % Wo);
    const int64_t t = m / Wo;
    const int yo = (int)(t % Ho);
    const int64_t n = t / Ho;
    const int yi = yo * stride - pad + ky, xi = xo * stride - pad + kx;
    if (yi >= 0 && yi < H && xi >= 0 && xi < W) v = *reinterpret_cast<const f32x4*>(x + ((n * H + yi) * W + xi) * C + c);
  }
  store4<OutT>(col + m * ldo + k, v);
}

// ---- im2col of the first layer: the 5-frame window is gathered on the fly (x3:1531-1539) ---------------------------
// x: (clips, T, H, W) single-channel; window n = clip*T + i; channel c = frame clamp(i + c - 2, 0, T-1).
// k = (c*kh + ky)*kw + kx: consecutive k are consecutive pixels of one frame row.  One thread per 4 k.
template <typename OutT>
__global__ __launch_bounds__(256) void im2col_window_kernel(const float* __restrict__ x, OutT* __restrict__ col, int64_t rows,
                                                            int T, int first, int H, int W, int kh, int kw, int K, int K4pad,
                                                            int stride, int pad, int Ho, int Wo, int64_t ldo) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= rows * K4pad) return;
  const int64_t m = gid / K4pad;
  const int k0 = (int)(gid - m * K4pad) * 4;
  const int xo = (int)(m % Wo);
  const int64_t t = m / Wo;
  const int yo = (int)(t % Ho);
  const int64_t n = first + t / Ho;       // global window index
  const int64_t clip = n / T;
  const int i = (int)(n - clip * T);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = k0 + e;
    if (k < K) {
      const int c = k / (kh * kw), r = k - c * kh * kw;
      const int ky = r / kw, kx = r - ky * kw;
      int f = i + c - 2;
      f = f < 0 ? 0 : (f > T - 1 ? T - 1 : f);
      const int yi = yo * stride - pad + ky, xi = xo * stride - pad + kx;
      if (yi >= 0 && yi < H && xi >= 0 && xi < W) v[e] = x[((clip * T + f) * H + yi) * W + xi];
    }
  }
  store4<OutT>(col + m * ldo + k0, v);
}

// ---- first-layer operand for the implicit GEMM: column patches of the replicate-padded clip -----------------------------------
// out[j][xo][y][e] (bf16) for j < T + 4 frames, xo < Wo, y < Hp = H + 2*pad, e < 16:
//   frame clamp(j - 2, 0, T - 1), row y - pad, column stride*xo - pad + e   (zero outside the image and for e >= kw)
// For output pixel (window i, yo, xo) and input channel c the kh x kw taps are then the kh consecutive 16-element rows
// starting at out[i + c][xo][stride*yo]: K tiles of 64 elements are contiguous, so the LDS-DMA GEMM reads them through its
// row / K-tile offset tables and the 28 MB-per-window patch matrix of v2a_im2col is never written.
__global__ __launch_bounds__(256) void frames_pack_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int64_t total, int T,
                                                          int H, int W, int Wo, int Hp, int kw, int stride, int pad) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one thread per (j, xo, y)
  if (gid >= total) return;
  const int y = (int)(gid % Hp);
  int64_t t = gid / Hp;
  const int xo = (int)(t % Wo);
  const int j = (int)(t / Wo);
  int f = j - 2;
  f = f < 0 ? 0 : (f > T - 1 ? T - 1 : f);
  const int yi = y - pad;
  bf16x8 lo, hi;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int xi = xo * stride - pad + e;
    float v = 0.f;
    if (e < kw && yi >= 0 && yi < H && xi >= 0 && xi < W) v = x[((int64_t)f * H + yi) * W + xi];
    if (e < 8) lo[e] = (bf16_t)v; else hi[e - 8] = (bf16_t)v;
  }
  bf16x8* dst = reinterpret_cast<bf16x8*>(out + gid * 16);
  dst[0] = lo;
  dst[1] = hi;
}

// ---- pooling, NHWC fp32, one thread per (n, yo, xo, 4 channels) ---------------------------------------------------------
template <bool MAX>
__global__ __launch_bounds__(256) void pool2d_kernel(const float* __restrict__ x, float* __restrict__ out, bf16_t* __restrict__ out2,
                                                     int64_t total, int H, int W, int C4, int k, int stride, int pad, int Ho, int Wo,
                                                     int ib, int ob) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int c4 = (int)(gid % C4);
  int64_t t = gid / C4;
  const int xo = (int)(t % Wo);
  t /= Wo;
  const int yo = (int)(t % Ho);
  const int64_t n = t / Ho;
  const int Hp = H + 2 * ib, Wp = W + 2 * ib;          // the maps may carry a zero border (implicit-GEMM layout)
  f32x4 acc = MAX ? f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY} : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ky = 0; ky < k; ++ky) {
    const int yi = yo * stride - pad + ky;
    if (yi < 0 || yi >= H) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int xi = xo * stride - pad + kx;
      if (xi < 0 || xi >= W) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((n * Hp + yi + ib) * Wp + xi + ib) * C4 + c4) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = MAX ? fmaxf(acc[e], v[e]) : acc[e] + v[e];
    }
  }
  if (!MAX) {
    const float inv = 1.0f / (float)(k * k);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] *= inv;
  }
  const int64_t o = (((n * (Ho + 2 * ob) + yo + ob) * (Wo + 2 * ob) + xo + ob) * C4 + c4) * 4;
  *reinterpret_cast<f32x4*>(out + o) = acc;
  if (out2) store4<bf16_t>(out2 + o, acc);
}

// ---- fused head: one workgroup of 128 threads (thread = channel) per window ---------------------------------------------
struct HeadParams {
  v2a_roll_head_args a;
};

// y[j] = act(b[j] + sum_k wt[k][j] * z[k]) for j = threadIdx.x < n_out; z in LDS
__device__ __forceinline__ float matvec_t(const float* __restrict__ wt, const float* __restrict__ b, const float* z, int n_in,
                                          int n_out, int j) {
  float acc = b[j];
#pragma unroll 8
  for (int k = 0; k < n_in; ++k) acc = fmaf(wt[(int64_t)k * n_out + j], z[k], acc);
  return acc;
}

__global__ __launch_bounds__(128) void roll_head_kernel(HeadParams hp) {
  const v2a_roll_head_args& a = hp.a;
  __shared__ float z[256];
  __shared__ float h[128];
  const int c = threadIdx.x;
  const int64_t n = blockIdx.x;
  const int P = a.P;
  const float invP = 1.0f / (float)P;
  const float* x2 = a.x2 + n * P * 128;
  const float* x3 = a.x3 + n * P * 128;
  const float* x4 = a.x4 + n * P * 128;
  const float* x5 = a.x5 + n * P * 64;
  // global average pools (v2r:46): per-channel position sums, lanes on consecutive channels
  float m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f;
  for (int p = 0; p < P; ++p) {
    m2 += x2[p * 128 + c];
    m3 += x3[p * 128 + c];
    m4 += x4[p * 128 + c];
    if (c < 64) m5 += x5[p * 64 + c];
  }
  m2 *= invP; m3 *= invP; m4 *= invP; m5 *= invP;
  // FRB4(xl = x4_, xh = x5): 192 -> 128 -> 128 (v2r:224)
  z[c] = m4;
  if (c < 64) z[128 + c] = m5;
  __syncthreads();
  h[c] = fmaxf(matvec_t(a.frb4_w1t, a.frb4_b1, z, 192, 128, c), 0.f);
  __syncthreads();
  const float s4 = sigmoid_f(matvec_t(a.frb4_w2t, a.frb4_b2, h, 128, 128, c));
  __syncthreads();
  // FRB3(xl = x3_, xh = p4 = s4 * x4_): mean(p4) = s4 * mean(x4_) (v2r:226)
  z[c] = m3;
  z[128 + c] = s4 * m4;
  __syncthreads();
  h[c] = fmaxf(matvec_t(a.frb3_w1t, a.frb3_b1, z, 256, 128, c), 0.f);
  __syncthreads();
  const float s3 = sigmoid_f(matvec_t(a.frb3_w2t, a.frb3_b2, h, 128, 128, c));
  __syncthreads();
  // FRB2(xl = x2_, xh = p3) (v2r:228)
  z[c] = m2;
  z[128 + c] = s3 * m3;
  __syncthreads();
  h[c] = fmaxf(matvec_t(a.frb2_w1t, a.frb2_b1, z, 256, 128, c), 0.f);
  __syncthreads();
  const float s2 = sigmoid_f(matvec_t(a.frb2_w2t, a.frb2_b2, h, 128, 128, c));
  __syncthreads();
  // out1 = p2*p3, softmax over positions per channel, out2 = softmax * p4 (v2r:230-234); only its position mean is needed
  const float s23 = s2 * s3;
  float mx = -INFINITY;
  for (int p = 0; p < P; ++p) mx = fmaxf(mx, x2[p * 128 + c] * x3[p * 128 + c] * s23);
  float den = 0.f, num = 0.f;
  for (int p = 0; p < P; ++p) {
    const float e = __expf(x2[p * 128 + c] * x3[p * 128 + c] * s23 - mx);
    den += e;
    num = fmaf(e, x4[p * 128 + c] * s4, num);
  }
  z[c] = num / den * invP;           // mean_p out2[p][c]
  __syncthreads();
  // mean_p(conv2(out2) + p4) = conv2_w * mean_p(out2) + conv2_b + s4 * mean_p(x4_)   (v2r:236-240)
  const float g = matvec_t(a.conv2_wt, a.conv2_b, z, 128, 128, c) + s4 * m4;
  __syncthreads();
  h[c] = g;
  __syncthreads();
  if (c < a.notes) {
    float o = matvec_t(a.fc_wt, a.fc_b, h, 128, a.notes, c);   // v2r:244-246
    if (a.apply_sigmoid) o = sigmoid_f(o);                      // x3:1541
    a.out[n * a.notes + c] = o;
  }
}

__global__ __launch_bounds__(256) void roll_expand_kernel(const float* __restrict__ roll, float* __restrict__ out, int64_t total,
                                                          int t, int notes, int rep, int l) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int j = (int)(gid % notes);
  const int64_t r = gid / notes;
  const int row = (int)(r % l);
  const int64_t b = r / l;
  const int src = row / rep;
  out[gid] = src < t ? roll[(b * t + src) * notes + j] : 0.f;
}

}  // namespace

extern "C" int v2a_im2col(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t stride,
                          int32_t pad, int32_t Ho, int32_t Wo, void* col, int64_t ldo, int32_t out_dtype, int32_t window_t,
                          int32_t window_first, v2a_stream_t stream) {
  V2A_REQUIRE(x && col, "v2a_im2col: null pointer");
  V2A_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, "v2a_im2col: bad geometry");
  V2A_REQUIRE(Ho == (H + 2 * pad - kh) / stride + 1 && Wo == (W + 2 * pad - kw) / stride + 1,
              "v2a_im2col: Ho/Wo (%d, %d) do not match the convolution geometry", Ho, Wo);
  V2A_REQUIRE(out_dtype == V2A_F32 || out_dtype == V2A_BF16, "v2a_im2col: out dtype %d", out_dtype);
  const int K = kh * kw * C;
  V2A_REQUIRE(ldo >= K && ldo % 4 == 0 && ((uintptr_t)col & 15) == 0 && ((uintptr_t)x & 15) == 0, "v2a_im2col: ldo=%lld (K=%d) / alignment",
              (long long)ldo, K);
  const int64_t rows = (int64_t)B * Ho * Wo;
  const int K4pad = (int)(ldo / 4);
  const int64_t threads = rows * K4pad;
  V2A_REQUIRE((threads + 255) / 256 < 0x7fffffffLL, "v2a_im2col: too many rows for one launch (%lld)", (long long)rows);
  dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (window_t > 0) {
    V2A_REQUIRE(C == 5 && window_first >= 0, "v2a_im2col: window mode needs C == 5 and window_first >= 0 (C=%d, first=%d)", C, window_first);
    if (out_dtype == V2A_F32)
      hipLaunchKernelGGL((im2col_window_kernel<float>), grid, block, 0, s, x, (float*)col, rows, window_t, window_first, H, W, kh, kw, K,
                         K4pad, stride, pad, Ho, Wo, ldo);
    else
      hipLaunchKernelGGL((im2col_window_kernel<bf16_t>), grid, block, 0, s, x, (bf16_t*)col, rows, window_t, window_first, H, W, kh, kw, K,
                         K4pad, stride, pad, Ho, Wo, ldo);
  } else {
    V2A_REQUIRE(C % 4 == 0, "v2a_im2col: NHWC mode needs C %% 4 == 0 (C=%d)", C);
    if (out_dtype == V2A_F32)
      hipLaunchKernelGGL((im2col_nhwc_kernel<float>), grid, block, 0, s, x, (float*)col, rows, H, W, C, kw, K, K4pad, stride, pad,
                         Ho, Wo, ldo);
    else
      hipLaunchKernelGGL((im2col_nhwc_kernel<bf16_t>), grid, block, 0, s, x, (bf16_t*)col, rows, H, W, C, kw, K, K4pad, stride, pad,
                         Ho, Wo, ldo);
  }
  return v2a_check_launch("v2a_im2col");
}

extern "C" int v2a_frames_pack(const float* frames, void* out, int32_t T, int32_t H, int32_t W, int32_t kw, int32_t stride, int32_t pad,
                               int32_t Wo, v2a_stream_t stream) {
  V2A_REQUIRE(frames && out, "v2a_frames_pack: null pointer");
  V2A_REQUIRE(T > 0 && H > 0 && W > 0 && kw > 0 && kw <= 16 && stride > 0 && pad >= 0, "v2a_frames_pack: bad geometry (kw=%d)", kw);
  V2A_REQUIRE(Wo == (W + 2 * pad - kw) / stride + 1, "v2a_frames_pack: Wo=%d does not match the convolution geometry", Wo);
  V2A_REQUIRE(((uintptr_t)out & 15) == 0, "v2a_frames_pack: 16-byte alignment");
  const int Hp = H + 2 * pad;
  const int64_t total = (int64_t)(T + 4) * Wo * Hp;
  hipLaunchKernelGGL(frames_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, frames, (bf16_t*)out,
                     total, T, H, W, Wo, Hp, kw, stride, pad);
  return v2a_check_launch("v2a_frames_pack");
}

extern "C" int v2a_pool2d(const float* x, float* out, void* out_bf16, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
                          int32_t stride, int32_t pad, int32_t mode, int32_t Ho, int32_t Wo, int32_t in_border, int32_t out_border,
                          v2a_stream_t stream) {
  V2A_REQUIRE(x && out && x != out, "v2a_pool2d: null / aliased pointer");
  V2A_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && k > 0 && stride > 0 && pad >= 0 && pad < k, "v2a_pool2d: bad geometry (C=%d k=%d)", C, k);
  V2A_REQUIRE(Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1, "v2a_pool2d: Ho/Wo mismatch");
  V2A_REQUIRE(mode == 0 || (mode == 1 && pad == 0), "v2a_pool2d: mode %d (average pooling is built for pad 0)", mode);
  V2A_REQUIRE(in_border >= 0 && out_border >= 0 && ((uintptr_t)out_bf16 & 7) == 0, "v2a_pool2d: borders / bf16 alignment");
  const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (mode == 0)
    hipLaunchKernelGGL((pool2d_kernel<true>), grid, block, 0, (hipStream_t)stream, x, out, (bf16_t*)out_bf16, total, H, W, C / 4, k,
                       stride, pad, Ho, Wo, in_border, out_border);
  else
    hipLaunchKernelGGL((pool2d_kernel<false>), grid, block, 0, (hipStream_t)stream, x, out, (bf16_t*)out_bf16, total, H, W, C / 4, k,
                       stride, pad, Ho, Wo, in_border, out_border);
  return v2a_check_launch("v2a_pool2d");
}

extern "C" int v2a_roll_head(const v2a_roll_head_args* a, v2a_stream_t stream) {
  V2A_REQUIRE(a != nullptr, "v2a_roll_head: null args");
  V2A_REQUIRE(a->x2 && a->x3 && a->x4 && a->x5 && a->out, "v2a_roll_head: null activation pointer");
  V2A_REQUIRE(a->frb4_w1t && a->frb4_b1 && a->frb4_w2t && a->frb4_b2 && a->frb3_w1t && a->frb3_b1 && a->frb3_w2t && a->frb3_b2 &&
                  a->frb2_w1t && a->frb2_b1 && a->frb2_w2t && a->frb2_b2 && a->conv2_wt && a->conv2_b && a->fc_wt && a->fc_b,
              "v2a_roll_head: null weight pointer");
  V2A_REQUIRE(a->B > 0 && a->P > 0 && a->notes > 0 && a->notes <= 128, "v2a_roll_head: B=%d P=%d notes=%d", a->B, a->P, a->notes);
  HeadParams hp{*a};
  hipLaunchKernelGGL(roll_head_kernel, dim3((unsigned)a->B), dim3(128), 0, (hipStream_t)stream, hp);
  return v2a_check_launch("v2a_roll_head");
}

extern "C" int v2a_roll_expand(const float* roll, float* out, int32_t B, int32_t t, int32_t notes, int32_t rep, int32_t l,
                               v2a_stream_t stream) {
  V2A_REQUIRE(roll && out && B > 0 && t > 0 && notes > 0 && rep > 0 && l > 0, "v2a_roll_expand: bad args");
  const int64_t total = (int64_t)B * l * notes;
  hipLaunchKernelGGL(roll_expand_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, roll, out, total, t,
                     notes, rep, l);
  return v2a_check_launch("v2a_roll_expand");
}
