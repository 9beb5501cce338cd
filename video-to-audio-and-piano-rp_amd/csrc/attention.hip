// Soft-clamped, key-masked, head-gated attention core for gfx950 (dim_head = 64).
//
//   s_ij = clamp * tanh(scale * <q_i, k_j> / clamp);  p = softmax_j(s_ij | j < kv_len)
//   o_i  = sigmoid(gate_i) * sum_j p_ij v_j;          rows i >= q_len -> 0
//
// The tanh soft-clamp sits between QK^T and the softmax, which rules out stock flash
// kernels.  Two implementations:
//   * attn_rowlane_kernel<T>: exact-fp32 arithmetic, one query row per lane, K/V tiles
//     broadcast from LDS.  This is the parity-mode kernel (and the fallback for odd shapes).
//   * attn_mfma_kernel (bf16): flash-style, QK^T and PV on v_mfma_f32_16x16x32_bf16 with
//     the softmax row reductions done by wavefront shuffles.
#include "v2a_common.h"

namespace {

struct AttnParams {
  const void *q, *k, *v, *gate;
  void* out;
  int64_t qrs, krs, vrs, grs, ors;
  int64_t qbs, kbs, vbs, gbs, obs;
  int32_t B, H, Nq, Nk;
  const int32_t* kv_len;
  const int32_t* q_len;
  float scale, clamp;
};

// ------------------------------------------------------------------------------------------
// Row-per-lane kernel.  grid = (ceil(Nq/64), H, B), block = 64.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void attn_rowlane_kernel(AttnParams p) {
  constexpr int D = 64, TK = 32;
  __shared__ __attribute__((aligned(16))) float ks[TK][D];
  __shared__ __attribute__((aligned(16))) float vs[TK][D];
  const int lane = threadIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  const int i = blockIdx.x * 64 + lane;
  const bool row_ok = i < p.Nq;
  const int kvn = p.kv_len ? min(p.kv_len[b], p.Nk) : p.Nk;

  float q[D], o[D];
  {
    const T* qp = reinterpret_cast<const T*>(p.q) + b * p.qbs + (int64_t)(row_ok ? i : 0) * p.qrs + h * D;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      q[c] = to_f32(qp[c]) * p.scale;
      o[c] = 0.f;
    }
  }
  float m = -INFINITY, l = 0.f;
  const float inv_clamp = p.clamp > 0.f ? 1.0f / p.clamp : 0.f;

  for (int j0 = 0; j0 < kvn; j0 += TK) {
    __syncthreads();
    // cooperative tile load: 2 * TK * D elements, coalesced along D
    for (int e = lane; e < TK * D; e += 64) {
      const int jj = e / D, c = e % D;
      const int j = j0 + jj;
      float kvv = 0.f, vvv = 0.f;
      if (j < kvn) {
        kvv = to_f32(reinterpret_cast<const T*>(p.k)[b * p.kbs + (int64_t)j * p.krs + h * D + c]);
        vvv = to_f32(reinterpret_cast<const T*>(p.v)[b * p.vbs + (int64_t)j * p.vrs + h * D + c]);
      }
      ks[jj][c] = kvv;
      vs[jj][c] = vvv;
    }
    __syncthreads();
    const int nj = min(TK, kvn - j0);
    float s[TK];
    float tmax = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < TK; ++jj) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < D; c += 4) {
        const f32x4 kk = *reinterpret_cast<const f32x4*>(&ks[jj][c]);  // same address in every lane: LDS broadcast
        acc += q[c] * kk[0] + q[c + 1] * kk[1] + q[c + 2] * kk[2] + q[c + 3] * kk[3];
      }
      if (p.clamp > 0.f) acc = tanhf(acc * inv_clamp) * p.clamp;
      s[jj] = jj < nj ? acc : -INFINITY;
      tmax = fmaxf(tmax, s[jj]);
    }
    const float mn = fmaxf(m, tmax);
    const float alpha = __expf(m - mn);  // m = -inf on the first tile -> 0
    l *= alpha;
#pragma unroll
    for (int c = 0; c < D; ++c) o[c] *= alpha;
#pragma unroll
    for (int jj = 0; jj < TK; ++jj) {
      const float pj = jj < nj ? __expf(s[jj] - mn) : 0.f;
      l += pj;
#pragma unroll
      for (int c = 0; c < D; c += 4) {
        const f32x4 vv = *reinterpret_cast<const f32x4*>(&vs[jj][c]);
        o[c] += pj * vv[0];
        o[c + 1] += pj * vv[1];
        o[c + 2] += pj * vv[2];
        o[c + 3] += pj * vv[3];
      }
    }
    m = mn;
  }
  if (!row_ok) return;
  const int qn = p.q_len ? min(p.q_len[b], p.Nq) : p.Nq;
  float g = 1.f;
  if (p.gate) g = sigmoid_f(to_f32(reinterpret_cast<const T*>(p.gate)[b * p.gbs + (int64_t)i * p.grs + h]));
  const float f = (i < qn && l > 0.f) ? g / l : 0.f;
  T* op = reinterpret_cast<T*>(p.out) + b * p.obs + (int64_t)i * p.ors + h * D;
#pragma unroll
  for (int c = 0; c < D; ++c) op[c] = from_f32<T>(o[c] * f);
}

}  // namespace

extern "C" int v2a_attention(const v2a_attn_args* a, v2a_stream_t stream) {
  V2A_REQUIRE(a != nullptr, "v2a_attention: null args");
  V2A_REQUIRE(a->q && a->k && a->v && a->out, "v2a_attention: null tensor");
  V2A_REQUIRE(a->B > 0 && a->H > 0 && a->Nq > 0 && a->Nk > 0, "v2a_attention: B=%d H=%d Nq=%d Nk=%d", a->B, a->H, a->Nq, a->Nk);
  V2A_REQUIRE(a->dtype == V2A_F32 || a->dtype == V2A_BF16, "v2a_attention: dtype %d", a->dtype);
  AttnParams p{};
  p.q = a->q; p.k = a->k; p.v = a->v; p.gate = a->gate; p.out = a->out;
  p.qrs = a->q_row_stride; p.krs = a->k_row_stride; p.vrs = a->v_row_stride; p.grs = a->gate_row_stride; p.ors = a->out_row_stride;
  p.qbs = a->q_batch_stride; p.kbs = a->k_batch_stride; p.vbs = a->v_batch_stride; p.gbs = a->gate_batch_stride; p.obs = a->out_batch_stride;
  p.B = a->B; p.H = a->H; p.Nq = a->Nq; p.Nk = a->Nk;
  p.kv_len = a->kv_len; p.q_len = a->q_len;
  p.scale = a->scale; p.clamp = a->softclamp;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((a->Nq + 63) / 64, a->H, a->B), block(64);
  if (a->dtype == V2A_F32)
    hipLaunchKernelGGL((attn_rowlane_kernel<float>), grid, block, 0, s, p);
  else
    hipLaunchKernelGGL((attn_rowlane_kernel<bf16_t>), grid, block, 0, s, p);
  return v2a_check_launch("v2a_attention");
}
