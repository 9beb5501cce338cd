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
#include <type_traits>

namespace v2a_detail { extern int g_attn_one_group_from; extern int g_probe_dbg; }

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
  int32_t out_split;     // split kernel only: out is a bf16 buffer that receives hi | lo planes (lo plane H * 64 columns after hi)
  int32_t dbg;           // probe builds only
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

// ------------------------------------------------------------------------------------------
// bf16 MFMA flash kernel.  grid = (ceil(Nq/64), H, B), block = 256 = 4 waves x 16 queries.
//
// Swapped product S^T = K Q^T on v_mfma_f32_16x16x32_bf16: the accumulator has the QUERY on the
// lane (col = lane & 15) and 4 keys per register group (row = 4*(lane>>4) + j), so
//   * the softmax row max needs two wave shuffles (xor 16, xor 32), the row sum one at the end;
//   * the fp32 P values, packed to bf16 in place, ARE the B operand of the next MFMA
//     O^T = V^T P^T (k-slot (g, jj) <-> key 16*(2ks + jj/4) + 4g + jj%4; the V^T fragment is read
//     in the same permuted key order), so P never touches LDS;
//   * O^T has the query on the lane again, so the rescale factor alpha is a per-lane scalar.
// K and V tiles sit in LDS row-major [key][64] with the 16-B chunk XOR-swizzled by (key & 7): conflict-free
// ds_read_b128 A fragments of K, and V^T fragments by the hardware-transposing ds_read_b64_tr_b16 (4 keys x 16
// head dimensions per 16-lane group).  2-deep LDS ring, register prefetch.
// ------------------------------------------------------------------------------------------

// NG = 1: 4 waves.  NG = 2: 8 waves, wave group g takes KV tiles g, g+2, ... of the same 64 queries (own LDS
// ring, shared barrier) and the two partial (m, l, O) states are merged through LDS at the end: twice the
// waves per SIMD to overlap the softmax VALU work of one wave with the MFMAs / loads of another.
// CLAMP: 0 = plain logits, 1 = soft clamp with the online running maximum, 2 = soft clamp WITHOUT a running maximum: clamped
// logits are bounded by +-clamp, so with clamp * log2(e) <= 100 the weights 2^v lie in [2^-100, 2^100] and their sums over any
// realistic key count stay far inside fp32 (and bf16 keeps fp32's exponent range for the P operand): the maximum, the
// subtraction, the rescale of O and l per tile and two wave shuffles per tile disappear from a VALU-bound loop.
template <int NG, int CLAMP>
__global__ __launch_bounds__(256 * NG) void attn_mfma_kernel(const AttnParams p) {
  const int h_ = blockIdx.y;
  constexpr int TK = 64;
  constexpr int K_ELEMS = TK * 64, V_ELEMS = TK * 64;     // both tiles row-major [key][64], 16-B chunks XOR-swizzled by (key & 7)
  constexpr int RING = 2 * (K_ELEMS + V_ELEMS);
  __shared__ __attribute__((aligned(16))) bf16_t lds_all[NG * RING];
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  bf16_t* lds = lds_all + grp * RING;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int h = h_, b = blockIdx.z;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int query = q0 + lr;
  const int kvn = p.kv_len ? min(p.kv_len[b], p.Nk) : p.Nk;
  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + b * p.qbs + h * 64;
  const bf16_t* Kg = reinterpret_cast<const bf16_t*>(p.k) + b * p.kbs + h * 64;
  const bf16_t* Vg = reinterpret_cast<const bf16_t*>(p.v) + b * p.vbs + h * 64;

  bf16x8 qf[2];
  {
    const int qr = query < p.Nq ? query : p.Nq - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(Q + (int64_t)qr * p.qrs + 32 * kk + 8 * g);
  }
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = CLAMP == 2 ? 0.f : -INFINITY, l = 0.f;
  constexpr float LOG2E = 1.4426950408889634f;
  // zc: raw QK^T -> argument of the base-2 exponential (2x*log2e with x = scale*s/clamp), or -> log2 units w/o clamp
  const float zc = p.clamp > 0.f ? 2.0f * LOG2E * p.scale / p.clamp : p.scale * LOG2E;
  const float c2 = p.clamp * LOG2E;

  // staging registers: rows (tid>>3)+32i, 16-B chunk tid&7 of the K tile and of the V tile (same geometry, same LDS image).
  // One register set: right after the barrier that ends iteration jt-1, tile jt+1 (requested a whole iteration earlier) is
  // written to the LDS buffer that iteration just stopped reading and tile jt+2 is requested into the same registers.
  bf16x8 kregA[2], vregA[2];
  const int kchunk = tid & 7, krow = tid >> 3;
  auto load_tile = [&](int j0, bf16x8 (&kreg)[2], bf16x8 (&vreg)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = j0 + krow + 32 * i;
      key = key < p.Nk ? key : p.Nk - 1;
      kreg[i] = *reinterpret_cast<const bf16x8*>(Kg + (int64_t)key * p.krs + kchunk * 8);
      vreg[i] = *reinterpret_cast<const bf16x8*>(Vg + (int64_t)key * p.vrs + kchunk * 8);
    }
  };
  auto store_tile = [&](bf16_t* base, const bf16x8 (&kreg)[2], const bf16x8 (&vreg)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = krow + 32 * i;
      const int off = row * 64 + ((kchunk ^ (row & 7)) << 3);
      *reinterpret_cast<bf16x8*>(base + off) = kreg[i];
      *reinterpret_cast<bf16x8*>(base + K_ELEMS + off) = vreg[i];
    }
  };
  // V^T fragments by the transposed LDS read (ds_read_b64_tr_b16): per 16-lane group a block of 4 keys x 16 head dimensions,
  // lane 4q+p of the group supplies the address of (key q, dimensions 4p..4p+3), lane i receives dimension i of the 4 keys --
  // the [d][4 consecutive keys] operand of O^T = V^T P^T straight from the row-major image (the round-1 kernel transposed V
  // on its way into LDS with eight 4-byte writes per thread and tile).  EXEC is all ones wherever this is read.
  const int vq = lr >> 2, vp = lr & 3;
  auto vt_read = [&](const bf16_t* vs, int key0, int dt) {
    const int key = key0 + 4 * g + vq;
    const int chunk = 2 * dt + (vp >> 1);
    const bf16_t* ad = vs + key * 64 + ((chunk ^ (key & 7)) << 3) + 4 * (vp & 1);
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)ad);
  };

  const int ntiles_all = (kvn + TK - 1) / TK;
  const int nit = (ntiles_all + NG - 1) / NG;             // iterations (barriers) of the whole block
  const int ntiles = (ntiles_all - grp + NG - 1) / NG;    // tiles of this wave group: global tile = jt * NG + grp
  if (ntiles > 0) {
    load_tile(grp * TK, kregA, vregA);
    store_tile(lds, kregA, vregA);
  }
  if (ntiles > 1) load_tile((NG + grp) * TK, kregA, vregA);
  __syncthreads();
  for (int jt = 0; jt < nit; ++jt) {
    if (jt >= ntiles) {          // this group ran out of tiles: keep the block's barrier count
      __syncthreads();
      continue;
    }
    const bf16_t* ks = lds + (jt & 1) * (K_ELEMS + V_ELEMS);
    const bf16_t* vt = ks + K_ELEMS;
    if (jt + 1 < ntiles) store_tile(lds + ((jt + 1) & 1) * (K_ELEMS + V_ELEMS), kregA, vregA);
    if (jt + 2 < ntiles) load_tile(((jt + 2) * NG + grp) * TK, kregA, vregA);
    // ---- S^T = K Q^T : 4 key tiles x 2 k-steps
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = 16 * t + lr;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + row * 64 + (((kk * 4 + g) ^ (row & 7)) << 3));
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[t], 0, 0, 0);
      }
    }
    // ---- soft clamp, key mask, online softmax (query on the lane), all in base-2 units:
    //   clamp*tanh(x)*log2e = C - 2C / (2^(2x log2e) + 1),  C = clamp*log2e   -> v_exp, v_rcp, 1 fma
    //   p = 2^(t - m)                                                          -> 1 sub, v_exp
    const int j0 = (jt * NG + grp) * TK;
    bf16x8 pf[2];
    if constexpr (CLAMP == 2) {
      // p = 2^(C - 2C / (2^(2x log2e) + 1)) directly; masked keys (last tile only) get weight 0
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float e = __builtin_amdgcn_exp2f(s[t][j] * zc);
          s[t][j] = __builtin_amdgcn_exp2f(fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f * c2, c2));
        }
      if (j0 + TK > kvn) {             // only the last tile can be partial (wave-uniform branch)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j0 + 16 * t + 4 * g + j >= kvn) s[t][j] = 0.f;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          l += s[t][j];
          pf[t >> 1][(t & 1) * 4 + j] = (bf16_t)s[t][j];
        }
    } else {
    float tmax = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v;
        if constexpr (CLAMP == 1) {
          const float e = __builtin_amdgcn_exp2f(s[t][j] * zc);
          v = fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f * c2, c2);
        } else {
          v = s[t][j] * zc;
        }
        s[t][j] = v;
      }
    if (j0 + TK > kvn) {             // only the last tile can be partial (wave-uniform branch)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j0 + 16 * t + 4 * g + j >= kvn) s[t][j] = -INFINITY;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) tmax = fmaxf(tmax, s[t][j]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mn = fmaxf(m, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
    l *= alpha;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[dt][j] *= alpha;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pv = __builtin_amdgcn_exp2f(s[t][j] - mn);
        l += pv;
        pf[t >> 1][(t & 1) * 4 + j] = (bf16_t)pv;
      }
    }
    // ---- O^T += V^T P^T : 4 d tiles x 2 k-steps of 32 keys (permuted key order, see header)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const bf16x4 lo = vt_read(vt, 32 * ks2, dt);
        const bf16x4 hi = vt_read(vt, 32 * ks2 + 16, dt);
        bf16x8 vf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vf[j] = lo[j];
          vf[4 + j] = hi[j];
        }
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ks2], o[dt], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- epilogue: row sum across the 4 key groups, merge of the wave groups, gate, query mask, 8-byte stores
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if constexpr (NG == 2) {
    // group 1 hands (m, l, O) to group 0, lane for lane (same query / d mapping), through its own dead LDS ring
    float* xch = reinterpret_cast<float*>(lds_all + RING);          // 18 floats x 256 lanes = 18 KB < RING bytes
    __syncthreads();
    if (grp == 1) {
      float* dst = xch + tid;
      dst[0] = m;
      dst[256] = l;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[(2 + dt * 4 + j) * 256] = o[dt][j];
    }
    __syncthreads();
    if (grp == 1) return;
    const float* src = xch + tid;
    const float m2 = src[0], l2 = src[256];
    const float mn = fmaxf(m, m2);
    const float a1 = __builtin_amdgcn_exp2f(m - mn), a2 = (m2 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[dt][j] = o[dt][j] * a1 + src[(2 + dt * 4 + j) * 256] * a2;
  }
  if (query >= p.Nq) return;
  const int qn = p.q_len ? min(p.q_len[b], p.Nq) : p.Nq;
  float gt = 1.f;
  if (p.gate) gt = sigmoid_f((float)reinterpret_cast<const bf16_t*>(p.gate)[b * p.gbs + (int64_t)query * p.grs + h]);
  const float f = (query < qn && l > 0.f) ? gt / l : 0.f;
  bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + b * p.obs + (int64_t)query * p.ors + h * 64 + 4 * g;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    bf16x4 ov;
#pragma unroll
    for (int j = 0; j < 4; ++j) ov[j] = (bf16_t)(o[dt][j] * f);
    *reinterpret_cast<bf16x4*>(op + 16 * dt) = ov;
  }
}

// ------------------------------------------------------------------------------------------
// Split-bf16 MFMA flash kernel for fp32 q, k, v, gate, out (the bf16x3 compute mode): the same schedule as attn_mfma_kernel
// with every operand as hi | lo bf16 planes (hi = bf16(v), lo = bf16(v - hi)) and every product as three MFMAs
//   K Q^T  ~ Kh Qh^T + Kl Qh^T + Kh Ql^T        V^T P^T ~ Vh^T Ph^T + Vl^T Ph^T + Vh^T Pl^T
// (the lo x lo term is 2^-16 of the product and is dropped, as in the bf16x3 GEMMs).  K / V tiles are split on their way from
// the fp32 rows into LDS (four planes per stage), q once per workgroup, P in registers after the fp32 softmax.  Replaces the
// one-query-per-lane fp32 VALU kernel in that mode: 518 -> ~70 us per self-attention launch at one clip.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bf16_t ha = (bf16_t)a[e], hb = (bf16_t)b[e];
    hi[e] = ha;
    hi[4 + e] = hb;
    lo[e] = (bf16_t)(a[e] - (float)ha);
    lo[4 + e] = (bf16_t)(b[e] - (float)hb);
  }
}

// QG (round 5): 2 = TWO groups of four waves share the K / V tiles of one ring -- 128 queries per workgroup, every thread converts and
// stages half as many K / V elements per tile and the tiles are fetched once per 128 queries instead of once per 64 (with NG = 1 only).
template <int NG, int CLAMP, int QG = 1>
__global__ __launch_bounds__(256 * NG * QG) void attn_mfma_split_kernel(const AttnParams p) {
  static_assert(QG == 1 || (QG == 2 && NG == 1), "query groups come with one key group");
  const int h_ = blockIdx.y;
  constexpr int TK = 64;
  constexpr int K_ELEMS = TK * 64, V_ELEMS = TK * 64;     // all four planes row-major [key][64], swizzled like the bf16 kernel's
  constexpr int STAGE = 2 * (K_ELEMS + V_ELEMS);          // Kh | Kl | Vh | Vl
  constexpr int RING = 2 * STAGE;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds_dyn[];
  const int hi_grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  const int grp = QG == 2 ? 0 : hi_grp;                   // key group (NG = 2) ...
  const int qgrp = QG == 2 ? hi_grp : 0;                  // ... or query group (QG = 2)
  bf16_t* lds = lds_dyn + grp * RING;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int h = h_, b = blockIdx.z;
  const int q0 = blockIdx.x * (64 * QG) + qgrp * 64 + wave * 16;
  const int query = q0 + lr;
  const int kvn = p.kv_len ? min(p.kv_len[b], p.Nk) : p.Nk;
  const float* Q = reinterpret_cast<const float*>(p.q) + b * p.qbs + h * 64;
  const float* Kg = reinterpret_cast<const float*>(p.k) + b * p.kbs + h * 64;
  const float* Vg = reinterpret_cast<const float*>(p.v) + b * p.vbs + h * 64;

  bf16x8 qh[2], ql[2];
  {
    const int qr = query < p.Nq ? query : p.Nq - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const float* qp = Q + (int64_t)qr * p.qrs + 32 * kk + 8 * g;
      split8(*reinterpret_cast<const f32x4*>(qp), *reinterpret_cast<const f32x4*>(qp + 4), qh[kk], ql[kk]);
    }
  }
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = CLAMP == 2 ? 0.f : -INFINITY, l = 0.f;
  constexpr float LOG2E = 1.4426950408889634f;
  const float zc = p.clamp > 0.f ? 2.0f * LOG2E * p.scale / p.clamp : p.scale * LOG2E;
  const float c2 = p.clamp * LOG2E;

  // staging registers (fp32 as loaded; split when written to LDS): K rows (tid>>3)+32i chunk tid&7; V key pair kp = lane&31,
  // d-chunk = 2*wave + (lane>>5).  Two sets, as in the bf16 kernel.
  struct Raw {
    f32x4 a, b;
  };
  constexpr int NR = 2 / QG;    // row passes per thread and tile: the 64 key rows over 256 (QG = 1) or 512 (QG = 2) staging threads
  Raw kregA[NR], vregA[NR];     // one set: written to LDS right after the barrier, re-requested at once (see attn_mfma_kernel)
  const int stid = QG == 2 ? (int)threadIdx.x : tid;
  const int kchunk = stid & 7, krow = stid >> 3;
  auto load_tile = [&](int j0, Raw (&kreg)[NR], Raw (&vreg)[NR]) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      int key = j0 + krow + 32 * i;
      key = key < p.Nk ? key : p.Nk - 1;
      const float* kpz = Kg + (int64_t)key * p.krs + kchunk * 8;
      kreg[i].a = *reinterpret_cast<const f32x4*>(kpz);
      kreg[i].b = *reinterpret_cast<const f32x4*>(kpz + 4);
      const float* vpz = Vg + (int64_t)key * p.vrs + kchunk * 8;
      vreg[i].a = *reinterpret_cast<const f32x4*>(vpz);
      vreg[i].b = *reinterpret_cast<const f32x4*>(vpz + 4);
    }
  };
  auto store_tile = [&](bf16_t* base, const Raw (&kreg)[NR], const Raw (&vreg)[NR]) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int row = krow + 32 * i;
      const int off = row * 64 + ((kchunk ^ (row & 7)) << 3);
      bf16x8 hi, lo;
      split8(kreg[i].a, kreg[i].b, hi, lo);
      *reinterpret_cast<bf16x8*>(base + off) = hi;
      *reinterpret_cast<bf16x8*>(base + K_ELEMS + off) = lo;
      split8(vreg[i].a, vreg[i].b, hi, lo);
      *reinterpret_cast<bf16x8*>(base + 2 * K_ELEMS + off) = hi;
      *reinterpret_cast<bf16x8*>(base + 2 * K_ELEMS + V_ELEMS + off) = lo;
    }
  };
  // V^T fragments by ds_read_b64_tr_b16 from the row-major planes (see attn_mfma_kernel)
  const int vq = lr >> 2, vp = lr & 3;
  auto vt_read = [&](const bf16_t* vs, int key0, int dt) {
    const int key = key0 + 4 * g + vq;
    const int chunk = 2 * dt + (vp >> 1);
    const bf16_t* ad = vs + key * 64 + ((chunk ^ (key & 7)) << 3) + 4 * (vp & 1);
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)ad);
  };

  const int ntiles_all = (kvn + TK - 1) / TK;
  const int nit = (ntiles_all + NG - 1) / NG;
  const int ntiles = (ntiles_all - grp + NG - 1) / NG;
  if (ntiles > 0) {
    load_tile(grp * TK, kregA, vregA);
    store_tile(lds, kregA, vregA);
  }
  if (ntiles > 1) load_tile((NG + grp) * TK, kregA, vregA);
  __syncthreads();
  for (int jt = 0; jt < nit; ++jt) {
    if (jt >= ntiles) {
      __syncthreads();
      continue;
    }
    const bf16_t* ksh = lds + (jt & 1) * STAGE;
    const bf16_t* ksl = ksh + K_ELEMS;
    const bf16_t* vth = ksh + 2 * K_ELEMS;
    const bf16_t* vtl = vth + V_ELEMS;
    if (jt + 1 < ntiles) store_tile(lds + ((jt + 1) & 1) * STAGE, kregA, vregA);
    if (jt + 2 < ntiles) load_tile(((jt + 2) * NG + grp) * TK, kregA, vregA);
    // ---- S^T = K Q^T in three passes
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = 16 * t + lr;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int off = row * 64 + (((kk * 4 + g) ^ (row & 7)) << 3);
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(ksh + off);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(ksl + off);
#ifdef V2A_GEMM_PROBE     // error attribution: hi planes only = the bf16 kernel's products
        if (!(p.dbg & 32)) {
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[kk], s[t], 0, 0, 0);
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[kk], s[t], 0, 0, 0);
        }
#else
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[kk], s[t], 0, 0, 0);
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[kk], s[t], 0, 0, 0);
#endif
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[kk], s[t], 0, 0, 0);
      }
    }
    const int j0 = (jt * NG + grp) * TK;
    if constexpr (CLAMP == 2) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float e = __builtin_amdgcn_exp2f(s[t][j] * zc);
          s[t][j] = __builtin_amdgcn_exp2f(fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f * c2, c2));
        }
      if (j0 + TK > kvn) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j0 + 16 * t + 4 * g + j >= kvn) s[t][j] = 0.f;
      }
    } else {
      float tmax = -INFINITY;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v;
          if constexpr (CLAMP == 1) {
            const float e = __builtin_amdgcn_exp2f(s[t][j] * zc);
            v = fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f * c2, c2);
          } else {
            v = s[t][j] * zc;
          }
          s[t][j] = v;
        }
      if (j0 + TK > kvn) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j0 + 16 * t + 4 * g + j >= kvn) s[t][j] = -INFINITY;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) tmax = fmaxf(tmax, s[t][j]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float mn = fmaxf(m, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      m = mn;
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[dt][j] *= alpha;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[t][j] = __builtin_amdgcn_exp2f(s[t][j] - mn);
    }
    // fp32 weights -> row sum, hi | lo operand planes
    bf16x8 pfh[2], pfl[2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pv = s[t][j];
        l += pv;
        const bf16_t hv = (bf16_t)pv;
        pfh[t >> 1][(t & 1) * 4 + j] = hv;
        pfl[t >> 1][(t & 1) * 4 + j] = (bf16_t)(pv - (float)hv);
      }
    // ---- O^T += V^T P^T in three passes
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        bf16x8 vfh, vfl;
        {
          const bf16x4 lo = vt_read(vth, 32 * ks2, dt), hi = vt_read(vth, 32 * ks2 + 16, dt);
          const bf16x4 lo2 = vt_read(vtl, 32 * ks2, dt), hi2 = vt_read(vtl, 32 * ks2 + 16, dt);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vfh[j] = lo[j];
            vfh[4 + j] = hi[j];
            vfl[j] = lo2[j];
            vfl[4 + j] = hi2[j];
          }
        }
#ifdef V2A_GEMM_PROBE
        if (!(p.dbg & 32)) {
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfl, pfh[ks2], o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfh, pfl[ks2], o[dt], 0, 0, 0);
        }
#else
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfl, pfh[ks2], o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfh, pfl[ks2], o[dt], 0, 0, 0);
#endif
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfh, pfh[ks2], o[dt], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if constexpr (NG == 2) {
    float* xch = reinterpret_cast<float*>(lds_dyn + RING);          // 18 floats x 256 lanes = 18 KB, group 1's dead ring
    __syncthreads();
    if (grp == 1) {
      float* dst = xch + tid;
      dst[0] = m;
      dst[256] = l;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[(2 + dt * 4 + j) * 256] = o[dt][j];
    }
    __syncthreads();
    if (grp == 1) return;
    const float* src = xch + tid;
    const float m2 = src[0], l2 = src[256];
    const float mn = fmaxf(m, m2);
    const float a1 = __builtin_amdgcn_exp2f(m - mn), a2 = (m2 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[dt][j] = o[dt][j] * a1 + src[(2 + dt * 4 + j) * 256] * a2;
  }
  if (query >= p.Nq) return;
  const int qn = p.q_len ? min(p.q_len[b], p.Nq) : p.Nq;
  float gt = 1.f;
  if (p.gate) gt = sigmoid_f(reinterpret_cast<const float*>(p.gate)[b * p.gbs + (int64_t)query * p.grs + h]);
  const float f = (query < qn && l > 0.f) ? gt / l : 0.f;
  if (p.out_split) {
    // the operand of the out-projection's split GEMM directly: hi | lo planes of the gated fp32 result, no v2a_split_bf16 pass
    bf16_t* ob = reinterpret_cast<bf16_t*>(p.out) + b * p.obs + (int64_t)query * p.ors + h * 64 + 4 * g;
    const int lo_off = p.H * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const f32x4 v = o[dt] * f;
      bf16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (bf16_t)v[e];
        lo[e] = (bf16_t)(v[e] - (float)hi[e]);
      }
      *reinterpret_cast<bf16x4*>(ob + 16 * dt) = hi;
      *reinterpret_cast<bf16x4*>(ob + lo_off + 16 * dt) = lo;
    }
    return;
  }
  float* op = reinterpret_cast<float*>(p.out) + b * p.obs + (int64_t)query * p.ors + h * 64 + 4 * g;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(op + 16 * dt) = o[dt] * f;
}

// ------------------------------------------------------------------------------------------
// fp32 MFMA flash kernel (the fp32 compute mode): the same schedule on v_mfma_f32_16x16x4_f32 -- fp32 operands, fp32 products,
// fp32 accumulation, as in the exact-fp32 GEMM.  The MFMA's k slot (c, g) carries head dimension 16 g + c for K Q^T, so a lane
// reads its 16 K values of a key as four ds_read_b128, and key 16 t + 4 g + j for V^T P^T, so the S^T accumulator register j of
// a lane IS the P^T operand and the V^T fragment is one ds_read_b128 (j = 0..3).  K tiles in LDS as [key][68] floats, V tiles
// transposed to [d][68].  Replaces the one-query-per-lane VALU kernel for aligned shapes: 518 -> ~80 us per launch.
// ------------------------------------------------------------------------------------------
template <int NG, int CLAMP>
__global__ __launch_bounds__(256 * NG) void attn_mfma_f32_kernel(AttnParams p) {
  constexpr int TK = 64, LD = 68;
  constexpr int T_ELEMS = 64 * LD;                      // floats of one K or V^T tile
  constexpr int STAGE = 2 * T_ELEMS;                    // K | V^T
  constexpr int RING = 2 * STAGE;
  extern __shared__ __attribute__((aligned(16))) float lds_f32[];
  const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  float* lds = lds_f32 + grp * RING;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int query = q0 + lr;
  const int kvn = p.kv_len ? min(p.kv_len[b], p.Nk) : p.Nk;
  const float* Q = reinterpret_cast<const float*>(p.q) + b * p.qbs + h * 64;
  const float* Kg = reinterpret_cast<const float*>(p.k) + b * p.kbs + h * 64;
  const float* Vg = reinterpret_cast<const float*>(p.v) + b * p.vbs + h * 64;

  f32x4 qf[4];                                           // q[query][16 g + 4 c4 + e]
  {
    const int qr = query < p.Nq ? query : p.Nq - 1;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) qf[c4] = *reinterpret_cast<const f32x4*>(Q + (int64_t)qr * p.qrs + 16 * g + 4 * c4);
  }
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = CLAMP == 2 ? 0.f : -INFINITY, l = 0.f;
  constexpr float LOG2E = 1.4426950408889634f;
  const float zc = p.clamp > 0.f ? 2.0f * LOG2E * p.scale / p.clamp : p.scale * LOG2E;
  const float c2 = p.clamp * LOG2E;

  struct Raw {
    f32x4 a, b;
  };
  Raw kregA[2], vregA[2];       // one set: written to LDS right after the barrier, re-requested at once (see attn_mfma_kernel)
  const int kchunk = tid & 7, krow = tid >> 3;
  const int kp = lane & 31, dch = wave * 2 + (lane >> 5);
  auto load_tile = [&](int j0, Raw (&kreg)[2], Raw (&vreg)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = j0 + krow + 32 * i;
      key = key < p.Nk ? key : p.Nk - 1;
      const float* kpz = Kg + (int64_t)key * p.krs + kchunk * 8;
      kreg[i].a = *reinterpret_cast<const f32x4*>(kpz);
      kreg[i].b = *reinterpret_cast<const f32x4*>(kpz + 4);
      int vk = j0 + 2 * kp + i;
      vk = vk < p.Nk ? vk : p.Nk - 1;
      const float* vpz = Vg + (int64_t)vk * p.vrs + dch * 8;
      vreg[i].a = *reinterpret_cast<const f32x4*>(vpz);
      vreg[i].b = *reinterpret_cast<const f32x4*>(vpz + 4);
    }
  };
  auto store_tile = [&](float* base, const Raw (&kreg)[2], const Raw (&vreg)[2]) {
    float* ks = base;
    float* vt = base + T_ELEMS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* dst = ks + (krow + 32 * i) * LD + kchunk * 8;
      *reinterpret_cast<f32x4*>(dst) = kreg[i].a;
      *reinterpret_cast<f32x4*>(dst + 4) = kreg[i].b;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      f32x2 pr;
      pr[0] = e < 4 ? vreg[0].a[e & 3] : vreg[0].b[e & 3];
      pr[1] = e < 4 ? vreg[1].a[e & 3] : vreg[1].b[e & 3];
      *reinterpret_cast<f32x2*>(vt + (dch * 8 + e) * LD + 2 * kp) = pr;
    }
  };

  const int ntiles_all = (kvn + TK - 1) / TK;
  const int nit = (ntiles_all + NG - 1) / NG;
  const int ntiles = (ntiles_all - grp + NG - 1) / NG;
  if (ntiles > 0) {
    load_tile(grp * TK, kregA, vregA);
    store_tile(lds, kregA, vregA);
  }
  if (ntiles > 1) load_tile((NG + grp) * TK, kregA, vregA);
  __syncthreads();
  for (int jt = 0; jt < nit; ++jt) {
    if (jt >= ntiles) {
      __syncthreads();
      continue;
    }
    const float* ks = lds + (jt & 1) * STAGE;
    const float* vt = ks + T_ELEMS;
    if (jt + 1 < ntiles) store_tile(lds + ((jt + 1) & 1) * STAGE, kregA, vregA);
    if (jt + 2 < ntiles) load_tile(((jt + 2) * NG + grp) * TK, kregA, vregA);
    // ---- S^T = K Q^T : 4 key tiles x 16 k-steps of 4 head dimensions
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* krowp = ks + (16 * t + lr) * LD + 16 * g;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(krowp + 4 * c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[c4][e], s[t], 0, 0, 0);
      }
    }
    const int j0 = (jt * NG + grp) * TK;
    if constexpr (CLAMP == 2) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float e = __builtin_amdgcn_exp2f(s[t][j] * zc);
          s[t][j] = __builtin_amdgcn_exp2f(fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f * c2, c2));
        }
      if (j0 + TK > kvn) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j0 + 16 * t + 4 * g + j >= kvn) s[t][j] = 0.f;
      }
    } else {
      float tmax = -INFINITY;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v;
          if constexpr (CLAMP == 1) {
            const float e = __builtin_amdgcn_exp2f(s[t][j] * zc);
            v = fmaf(__builtin_amdgcn_rcpf(e + 1.0f), -2.0f * c2, c2);
          } else {
            v = s[t][j] * zc;
          }
          s[t][j] = v;
        }
      if (j0 + TK > kvn) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j0 + 16 * t + 4 * g + j >= kvn) s[t][j] = -INFINITY;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) tmax = fmaxf(tmax, s[t][j]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float mn = fmaxf(m, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      m = mn;
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[dt][j] *= alpha;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[t][j] = __builtin_amdgcn_exp2f(s[t][j] - mn);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) l += s[t][j];
    // ---- O^T += V^T P^T : 4 d tiles x 16 k-steps of 4 keys; the accumulator register j of S^T is the P^T operand
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const float* vrow = vt + (16 * dt + lr) * LD + 4 * g;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 vf = *reinterpret_cast<const f32x4*>(vrow + 16 * t);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[j], s[t][j], o[dt], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if constexpr (NG == 2) {
    float* xch = lds_f32 + RING;                                    // 18 floats x 256 lanes = 18 KB, group 1's dead ring
    __syncthreads();
    if (grp == 1) {
      float* dst = xch + tid;
      dst[0] = m;
      dst[256] = l;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[(2 + dt * 4 + j) * 256] = o[dt][j];
    }
    __syncthreads();
    if (grp == 1) return;
    const float* src = xch + tid;
    const float m2 = src[0], l2 = src[256];
    const float mn = fmaxf(m, m2);
    const float a1 = __builtin_amdgcn_exp2f(m - mn), a2 = (m2 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[dt][j] = o[dt][j] * a1 + src[(2 + dt * 4 + j) * 256] * a2;
  }
  if (query >= p.Nq) return;
  const int qn = p.q_len ? min(p.q_len[b], p.Nq) : p.Nq;
  float gt = 1.f;
  if (p.gate) gt = sigmoid_f(reinterpret_cast<const float*>(p.gate)[b * p.gbs + (int64_t)query * p.grs + h]);
  const float f = (query < qn && l > 0.f) ? gt / l : 0.f;
  float* op = reinterpret_cast<float*>(p.out) + b * p.obs + (int64_t)query * p.ors + h * 64 + 4 * g;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(op + 16 * dt) = o[dt] * f;
}

template <int NG, int CLAMP>
int launch_attn_f32(const AttnParams& p, dim3 grid, hipStream_t s) {
  constexpr size_t smem = (size_t)NG * 2 * 2 * (64 * 68) * sizeof(float);
  auto kern = attn_mfma_f32_kernel<NG, CLAMP>;
  static std::atomic<uint64_t> lds_set{0};
  if (int rc = v2a_enable_lds(reinterpret_cast<const void*>(kern), smem, lds_set, "v2a_attention")) return rc;
  hipLaunchKernelGGL(kern, grid, dim3(256 * NG), smem, s, p);
  return V2A_OK;
}

template <int NG, int CLAMP, int QG = 1>
int launch_attn_split(const AttnParams& p, dim3 grid, hipStream_t s) {
  constexpr size_t smem = (size_t)NG * 2 * 2 * (64 * 64 + 64 * 64) * sizeof(bf16_t);
  auto kern = attn_mfma_split_kernel<NG, CLAMP, QG>;
  static std::atomic<uint64_t> lds_set{0};
  if (int rc = v2a_enable_lds(reinterpret_cast<const void*>(kern), smem, lds_set, "v2a_attention")) return rc;
  grid.x = (grid.x + QG - 1) / QG;                 // 64 * QG queries per workgroup
  hipLaunchKernelGGL(kern, grid, dim3(256 * NG * QG), smem, s, p);
  return V2A_OK;
}

}  // namespace

// 0 = no soft clamp, 1 = soft clamp with the running maximum, 2 = soft clamp with BOUNDED weights: logits lie in +-clamp, so
// p = 2^(logit * log2 e) lies in 2^(+-clamp * log2 e) and no maximum has to be tracked -- as long as the fp32 sums l = sum p
// and O = sum p v stay finite: Nk * max|v| * 2^(clamp * log2 e) < 2^128.  Mode 2 is taken while clamp * log2 e + log2 Nk <= 90
// (the shipped clamp 50 with 782 keys: 72.1 + 9.6), which leaves |v| up to 2^38; beyond that the running-maximum kernel runs.
static int attn_clamp_mode(float softclamp, int Nk) {
  if (!(softclamp > 0.f)) return 0;
  return softclamp * 1.4426950408889634f + log2f((float)(Nk > 1 ? Nk : 1)) <= 90.f ? 2 : 1;
}

extern "C" int v2a_attention(const v2a_attn_args* a, v2a_stream_t stream) {
  V2A_REQUIRE(a != nullptr, "v2a_attention: null args");
  V2A_REQUIRE(a->q && a->k && a->v && a->out, "v2a_attention: null tensor");
  V2A_REQUIRE(a->B > 0 && a->H > 0 && a->Nq > 0 && a->Nk > 0, "v2a_attention: B=%d H=%d Nq=%d Nk=%d", a->B, a->H, a->Nq, a->Nk);
  V2A_REQUIRE(a->dtype == V2A_F32 || a->dtype == V2A_BF16 || a->dtype == V2A_BF16_SPLIT, "v2a_attention: dtype %d", a->dtype);
  AttnParams p{};
  p.q = a->q; p.k = a->k; p.v = a->v; p.gate = a->gate; p.out = a->out;
  p.qrs = a->q_row_stride; p.krs = a->k_row_stride; p.vrs = a->v_row_stride; p.grs = a->gate_row_stride; p.ors = a->out_row_stride;
  p.qbs = a->q_batch_stride; p.kbs = a->k_batch_stride; p.vbs = a->v_batch_stride; p.gbs = a->gate_batch_stride; p.obs = a->out_batch_stride;
  p.B = a->B; p.H = a->H; p.Nq = a->Nq; p.Nk = a->Nk;
  p.kv_len = a->kv_len; p.q_len = a->q_len;
  p.scale = a->scale; p.clamp = a->softclamp;
  p.out_split = a->out_split ? 1 : 0;
  p.dbg = v2a_detail::g_probe_dbg;
  V2A_REQUIRE(!a->out_split || a->dtype == V2A_BF16_SPLIT, "v2a_attention: out_split goes with dtype V2A_BF16_SPLIT");
  hipStream_t s = (hipStream_t)stream;
  int rc = V2A_OK;
  dim3 grid((a->Nq + 63) / 64, a->H, a->B), block(64);
  if (a->dtype == V2A_BF16_SPLIT) {
    // fp32 tensors, split-bf16 MFMA arithmetic (bf16x3 mode); 16-byte aligned head slices and output rows, else the VALU kernel
    const bool aligned = (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->out) & 15) == 0 && a->q_row_stride % 4 == 0 &&
                         a->k_row_stride % 4 == 0 && a->v_row_stride % 4 == 0 && a->out_row_stride % 4 == 0 && a->q_batch_stride % 4 == 0 &&
                         a->k_batch_stride % 4 == 0 && a->v_batch_stride % 4 == 0 && a->out_batch_stride % 4 == 0;
    const dim3 g64((a->Nq + 63) / 64, a->H, a->B);
    const int cl = attn_clamp_mode(a->softclamp, a->Nk);
    V2A_REQUIRE(aligned || !a->out_split, "v2a_attention: out_split needs 16-byte aligned head slices");
    if (a->out_split) V2A_REQUIRE(((uintptr_t)a->out & 7) == 0 && a->out_row_stride >= 2 * (int64_t)a->H * 64, "v2a_attention: split output rows hold 2 * H * 64 bf16");
    // two wave groups split the key tiles of a workgroup's 64 queries only while the launch is short of workgroups (< 200, or the
    // v2a_tuning.attn_one_group_from override): one group per workgroup -- 64 KB of LDS, two workgroups per CU, no merge -- measured +1.3 % end to
    // end at 8 clips per GPU (3328 workgroups) and +0.7 % at one clip (416 / 208), profiles/r05_bf16x3_8clips_ab.txt
    const int one_from = v2a_detail::g_attn_one_group_from != 1536 ? v2a_detail::g_attn_one_group_from : 200;
    const bool two_groups = a->Nk > 128 && (int64_t)g64.x * g64.y * g64.z < one_from;
    if (!aligned) {
      hipLaunchKernelGGL((attn_rowlane_kernel<float>), grid, block, 0, s, p);
    } else if (two_groups) {
      if (cl == 2) rc = launch_attn_split<2, 2>(p, g64, s);
      else if (cl == 1) rc = launch_attn_split<2, 1>(p, g64, s);
      else rc = launch_attn_split<2, 0>(p, g64, s);
    } else if (a->Nq > 64 && a->Nk > 128 && !(v2a_detail::g_probe_dbg & 256)) {
      // one key group, two query groups per workgroup (128 queries share every K / V tile); v2a_tuning.reserved[0] bit 8: the 64-query form (A/B)
      if (cl == 2) rc = launch_attn_split<1, 2, 2>(p, g64, s);
      else if (cl == 1) rc = launch_attn_split<1, 1, 2>(p, g64, s);
      else rc = launch_attn_split<1, 0, 2>(p, g64, s);
    } else {
      if (cl == 2) rc = launch_attn_split<1, 2>(p, g64, s);
      else if (cl == 1) rc = launch_attn_split<1, 1>(p, g64, s);
      else rc = launch_attn_split<1, 0>(p, g64, s);
    }
  } else if (a->dtype == V2A_F32) {
    const bool aligned = (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->out) & 15) == 0 && a->q_row_stride % 4 == 0 &&
                         a->k_row_stride % 4 == 0 && a->v_row_stride % 4 == 0 && a->out_row_stride % 4 == 0 && a->q_batch_stride % 4 == 0 &&
                         a->k_batch_stride % 4 == 0 && a->v_batch_stride % 4 == 0 && a->out_batch_stride % 4 == 0;
    const dim3 g64((a->Nq + 63) / 64, a->H, a->B);
    const int cl = attn_clamp_mode(a->softclamp, a->Nk);
    if (!aligned) {
      hipLaunchKernelGGL((attn_rowlane_kernel<float>), grid, block, 0, s, p);
    } else if (a->Nk > 128) {
      if (cl == 2) rc = launch_attn_f32<2, 2>(p, g64, s);
      else if (cl == 1) rc = launch_attn_f32<2, 1>(p, g64, s);
      else rc = launch_attn_f32<2, 0>(p, g64, s);
    } else {
      if (cl == 2) rc = launch_attn_f32<1, 2>(p, g64, s);
      else if (cl == 1) rc = launch_attn_f32<1, 1>(p, g64, s);
      else rc = launch_attn_f32<1, 0>(p, g64, s);
    }
  } else {
    // MFMA path needs 16-byte aligned head slices for its vector loads and 8-byte aligned output rows
    const bool aligned = (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v) & 15) == 0 && ((uintptr_t)a->out & 7) == 0 &&
                         a->q_row_stride % 8 == 0 && a->k_row_stride % 8 == 0 && a->v_row_stride % 8 == 0 &&
                         a->q_batch_stride % 8 == 0 && a->k_batch_stride % 8 == 0 && a->v_batch_stride % 8 == 0 &&
                         a->out_row_stride % 4 == 0 && a->out_batch_stride % 4 == 0;
    const dim3 g64((a->Nq + 63) / 64, a->H, a->B);
    // soft clamp with bounded weights (no running maximum) while 2^(clamp * log2 e) stays far inside fp32: clamp <= 69
    const int cl = attn_clamp_mode(a->softclamp, a->Nk);
    // two wave groups split the key tiles of a workgroup's 64 queries (merged at the end) while the launch is short of
    // workgroups; from ~6 workgroups per CU on, one group per workgroup: no merge, 4-wave barriers
    const bool split_kv = a->Nk > 128 && (int64_t)g64.x * g64.y * g64.z < v2a_detail::g_attn_one_group_from;
    if (aligned && split_kv) {
      if (cl == 2) hipLaunchKernelGGL((attn_mfma_kernel<2, 2>), g64, dim3(512), 0, s, p);
      else if (cl == 1) hipLaunchKernelGGL((attn_mfma_kernel<2, 1>), g64, dim3(512), 0, s, p);
      else hipLaunchKernelGGL((attn_mfma_kernel<2, 0>), g64, dim3(512), 0, s, p);
    } else if (aligned) {
      if (cl == 2) hipLaunchKernelGGL((attn_mfma_kernel<1, 2>), g64, dim3(256), 0, s, p);
      else if (cl == 1) hipLaunchKernelGGL((attn_mfma_kernel<1, 1>), g64, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_mfma_kernel<1, 0>), g64, dim3(256), 0, s, p);
    }
    else
      hipLaunchKernelGGL((attn_rowlane_kernel<bf16_t>), grid, block, 0, s, p);
  }
  if (rc != V2A_OK) return rc;
  return v2a_check_launch("v2a_attention");
}
