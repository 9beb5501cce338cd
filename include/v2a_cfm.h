/*
 * v2a_cfm.h -- C ABI of the MI355X (gfx950) kernels behind the flow-matching V2A sampler.
 *
 * The reference (acappemin/Video-to-Audio-and-Piano-RP) has no FFI/operator layer: the hot
 * path is a Python class API (E2TTS.sample / transformer_with_pred_head) whose arithmetic
 * runs as stock PyTorch ops.  Each entry point below therefore names the reference
 * *module/function* it replaces (file:line, `x3` = src/e2_tts_pytorch/e2_tts_crossatt3.py;
 * `xt` = third-party x-transformers==1.37.4, requirements.txt:19, call sites given).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name says `host`;
 *   - no allocation and no synchronisation inside: the caller owns every buffer and passes
 *     the stream; all launches are graph-capturable.  The only process-wide state is the
 *     explicit tile-selection override of v2a_set_tuning() (benchmarking aid, defaults =
 *     automatic) and per-kernel "large LDS" attributes set once, thread-safely, at the first
 *     launch of each kernel; nothing is read from the environment;
 *   - return value 0 = ok, negative = error; v2a_last_error() gives the message of the
 *     last failing call on the calling thread;
 *   - "compute dtype" T is V2A_F32 (parity mode, exact-fp32 MFMA) or V2A_BF16 (bf16
 *     operands, fp32 accumulate).  Residual streams are always fp32;
 *   - a "step vector" is a float vector selected per launch by a device-side step counter
 *     and per row by its batch:  v = base + step[0]*step_stride + batch*batch_stride,
 *     batch = row / rows_per_batch.  step may be NULL (= 0).  This is how the per-Euler-step
 *     AdaLN / AdaptiveRMSNorm modulation tables are addressed from ONE captured hipGraph.
 */
#ifndef V2A_CFM_H
#define V2A_CFM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* v2a_stream_t; /* hipStream_t */

enum {
  V2A_F32 = 0,
  V2A_BF16 = 1,
  /* split-bf16 operand layout (the "bf16x3" parity mode): a row of d fp32 values v is stored as 2*d bf16 values
   * [hi_0 .. hi_{d-1} | lo_0 .. lo_{d-1}], hi = bf16(v), lo = bf16(v - hi).  A GEMM on the concatenation
   * [A_hi | A_hi | A_lo] x [W_hi | W_lo | W_hi]^T (three K segments of v2a_gemm) then sums hi*hi + hi*lo + lo*hi in fp32:
   * fp32-grade products (relative error ~2^-16) at three bf16 MFMAs instead of one fp32 MFMA at 1/16 of the rate.
   * Accepted as y_dtype of v2a_rmsnorm and by v2a_split_bf16; never a compute dtype. */
  V2A_BF16_SPLIT = 2
};

enum {
  V2A_OK = 0,
  V2A_ERR_ARG = -1,    /* shape/alignment/enum the kernels do not support */
  V2A_ERR_LAUNCH = -2  /* hipGetLastError() after launch */
};

int v2a_abi_version(void);        /* 8 */
const char* v2a_last_error(void);

/* ---------------------------------------------------------------------------------------
 * GEMM with fused epilogues:  acc[m][n] = sum_k A[m][k] * W[n][k]      (nn.Linear layout)
 * A is the virtual concatenation along K of up to 3 row-major segments (so
 * TextAudioCrossCondition's pack((audio,text,frames)) x3:693-700 and the U-Net skip
 * torch.cat((x, skip)) x3:1116-1117 are never materialised).
 * Replaces: every nn.Linear on the path -- xt Attention.to_q/to_k/to_v/to_out (sites
 * x3:808,813,881,914), xt FeedForward (x3:817,884,917), TextAudioCrossCondition
 * x3:698-700, skip_proj x3:1117, to_pred x3:2083, AdaLNZero.to_gamma x3:550 and
 * AdaptiveRMSNorm.to_gamma (table build).
 * ------------------------------------------------------------------------------------- */
enum {
  V2A_EPI_STORE = 0,      /* out = acc + bias                                    */
  V2A_EPI_SIGMOID = 1,    /* out = sigmoid(acc + bias)            (AdaLNZero table) */
  V2A_EPI_GEGLU = 2,      /* W rows packed [16 value | 16 gate] per 16 outputs:
                             out[m][j] = (acc_v + b_v) * gelu_erf(acc_g + b_g); out has N/2 cols */
  V2A_EPI_RESID = 3,      /* out = resid + acc + bias             (text/frames streams, cross-condition) */
  V2A_EPI_GATE_RESID = 4  /* out = resid + gate[n] * (acc + bias) (AdaLNZero x3:546-551 + residual x3:1128) */
};

typedef struct v2a_gemm_args {
  const void* a[3];      /* segment base pointers                                       */
  int64_t lda[3];        /* row stride of each segment, in elements                     */
  int32_t ka[3];         /* K extent of each segment; each a multiple of 64 (bf16) / 16 (f32) */
  int32_t nseg;          /* 1..3                                                        */
  int32_t a_dtype;       /* V2A_F32 or V2A_BF16; f32 A with bf16 compute is converted on load */
  const void* w;         /* [N][K] row-major, compute dtype, K = sum ka                 */
  int64_t ldw;
  const float* bias;     /* [N] or NULL                                                 */
  int32_t M, N;
  int32_t compute_dtype; /* V2A_F32 | V2A_BF16                                          */
  int32_t epilogue;      /* V2A_EPI_*                                                   */
  void* out;             /* [M][N] (GEGLU: [M][N/2])                                    */
  int64_t ldo;
  int32_t out_dtype;     /* V2A_F32 | V2A_BF16 (RESID/GATE_RESID/SIGMOID: f32 only; GEGLU: bf16, or f32 with bf16 compute) */
  void* out_bf16;        /* optional bf16 shadow copy of an f32 output (operand of a later GEMM), or NULL */
  int64_t ld_out_bf16;
  const float* resid;    /* [M][ldr] f32; may alias out                                 */
  int64_t ldr;
  const float* gate;     /* step vector of length N (GATE_RESID)                        */
  const int32_t* step;
  int64_t gate_step_stride, gate_batch_stride;
  int32_t rows_per_batch;
  /* optional fused rotary embedding (bf16 STORE epilogue only): columns [0, rope_cols) are 64-wide heads whose
   * interleaved pairs (2i, 2i+1) are rotated by cs_table[rope_pos_offset + m % rows_per_batch][i] = (cos, sin);
   * replaces a separate v2a_rope_inplace(layout 0) pass over the fused [q|k|v|gate] output */
  const float* rope_table;
  int32_t rope_cols, rope_pos_offset;
  int32_t relu;          /* non-zero: out = max(out, 0) after the epilogue (not GEGLU): the conv + folded-BatchNorm + ReLU
                          * and conv + BN + residual + ReLU blocks of the Video2Roll encoder, Video2RollNet.py:70-88 */
  /* implicit-GEMM convolution (bf16 compute, one bf16 A segment, STORE / RESID epilogue): with a_row_offset the A row m
   * starts at a[0] + a_row_offset[m] and K tile kt (64 elements) adds a_ktile_offset[kt] -- for an NHWC map stored with a
   * zero border, a_row_offset = top-left tap of output pixel m and a_ktile_offset[kt] = (ky*Wp + kx)*C + c0, so the patch
   * matrix of nn.Conv2d (Video2RollNet.py:9-12) is never materialised; lda is ignored.  With out_row_offset, row m of out,
   * resid and out_bf16 starts at base + out_row_offset[m] (the interior of the next layer's bordered map) instead of m * ld.
   * All offsets in elements, multiples of 8; NULL = dense rows. */
  const int32_t* a_row_offset;
  const int32_t* a_ktile_offset;
  const int32_t* out_row_offset;
  /* 0 = the library picks the tile shape for the fastest stand-alone launch.  k + 1 = use LDS-DMA tile configuration k of
   * v2a_tuning.gemm_force_tile for THIS call (bf16 x bf16 only).  The sampler passes 1 (128x256 tiles, one workgroup per CU) for
   * the text / frames streams: their GEMMs run beside the audio stream's, and few fat workgroups that own whole CUs disturb
   * the critical path less than many small ones spread over every CU (+3.5 % end to end, measured). */
  int32_t tile_hint;
  /* RMSNorm folded into its neighbours (bf16 x bf16, 16-byte aligned epilogue operands, N % 32 == 0):
   * PRODUCER (RESID / GATE_RESID with out_bf16): with norm_gamma the shadow is out_bf16[m][n] = bf16(out[m][n] * gamma[n]),
   * gamma = norm_gamma + step[0] * norm_step_stride + (m / rows_per_batch) * norm_batch_stride (+ norm_switch_offset for rows
   * m >= norm_switch_row: the rows a later GEMM of the block does not touch carry the NEXT norm's gamma), and with norm_ssq
   * the sums of squares of out[m][32 j .. 32 j + 31] go to norm_ssq[m * ld_norm_ssq + j].
   * CONSUMER (any epilogue): with row_ssq the accumulator row m is multiplied by
   * sqrt(row_norm_dim) / max(sqrt(sum_{j < row_ssq_parts} row_ssq[m * ld_row_ssq + j]), 1e-12) before the bias -- F.normalize
   * of xt RMSNorm / AdaptiveRMSNorm commutes with the product, so norm -> Linear costs no pass of its own.  row_ssq_parts <= 40;
   * rows of row_ssq are read as whole float4: ld_row_ssq a multiple of 4 and the columns past row_ssq_parts zero. */
  const float* norm_gamma;
  int64_t norm_step_stride, norm_batch_stride;
  int32_t norm_switch_row, norm_switch_offset;
  float* norm_ssq;
  int64_t ld_norm_ssq;
  const float* row_ssq;
  int64_t ld_row_ssq;
  int32_t row_ssq_parts, row_norm_dim;
  /* non-zero: the out_bf16 shadow is written in the V2A_BF16_SPLIT layout, row m = [hi_0 .. hi_{N-1} | lo_0 .. lo_{N-1}] of the
   * (gamma-scaled, when norm_gamma is given) fp32 result, ld_out_bf16 >= 2 * N: the operand of a later split-bf16 GEMM without a
   * v2a_split_bf16 pass.  Likewise out_dtype = V2A_BF16_SPLIT (GEGLU epilogue only): out row m = [hi | lo] planes of the N/2
   * hidden values, ldo >= N, exact erf GELU. */
  int32_t out_bf16_split;
  /* ABI 8, split operands / split shadow only.  a_lo_offset[s]: elements from the hi plane of a row of segment s to its lo plane; 0 = ka[s], the
   * layout [hi k | lo k].  out_bf16_lo_offset: the same for the shadow row; 0 = N.  They let ONE buffer of rows [x_hi | s_hi | x_lo | s_lo] serve
   * as a K = 2d segment (x and the U-Net skip concatenated, lo plane 2d further) AND, half by half, as a K = d segment or as the split shadow
   * two different GEMM epilogues write (lo plane 2d further, not d): the fused cross-condition + skip projection of the bf16x3 mode.
   * Multiples of 8 (a_lo_offset) / 4 (out_bf16_lo_offset); lda[s] >= a_lo_offset[s] + ka[s], ld_out_bf16 >= out_bf16_lo_offset + N. */
  int64_t a_lo_offset[3];
  int64_t out_bf16_lo_offset;
} v2a_gemm_args;

int v2a_gemm(const v2a_gemm_args* args, v2a_stream_t stream);
/* sizeof(v2a_gemm_args) as the library was built: a binding checks its mirror of the struct against this */
int v2a_gemm_args_size(void);

/* Tile-selection overrides of v2a_gemm for A/B measurements (bench.py, scripts/).  Library defaults: force -1, rotation 0,
 * 8-phase kernel on (staggered) from 400 tiles.
 * Process-wide; call it between launches, not concurrently with them.  NULL restores the defaults. */
typedef struct v2a_tuning {
  int32_t gemm_force_tile;        /* -1 = by shape; 0..5 = one LDS-DMA tile shape for every bf16 x bf16 GEMM (0 128x256, 1 128x128, 2 128x64, 3 64x64, 5 256x256), 6 = the 256x256 8-phase kernel, 7 / 8 = 64x128 / 64x64 with a 6-deep ring */
  int32_t gemm_k_rotation;        /* ignored since ABI 5 (was: M bands that share a W panel start their K walk at different K tiles; +0.7 %, and it made
                                   * the fp32 summation order depend on M): the K loop now walks running pointers */
  int32_t gemm_8phase;            /* 256x256 phase-interleaved kernel for wide outputs (N >= 2048): 0 off, 1 on (staggered wave rows), 2 on (lock-step) */
  int32_t gemm_8phase_min_tiles;  /* ... when the problem yields at least this many 256x256 tiles (0 = 400) */
  int32_t dwconv_rows_per_wave;   /* v2a_dwconv_silu_residual: output positions per wave pass of the small-launch kernel, 4 or 6 (0 = default 4);
                                   * -1 = never use the streaming kernel that chip-filling launches take (A/B) */
  int32_t gemm_xcd_order_1x8;     /* 1: every XCD walks whole column strips of the tile space (the round-1 order) instead of the
                                   * per-shape gm x gn rectangle grid that minimises operand re-fetch across the 8 L2s */
  int32_t attn_one_group_from;    /* v2a_attention (bf16): launches with at least this many workgroups run one wave group per workgroup
                                   * instead of two that split the key tiles (0 = default 1536) */
  int32_t reserved[1];            /* A/B bits: 128 = GEGLU epilogue with 8-byte (four-column) stores, 256 = split attention with 64 queries per workgroup, 512 = 8-phase kernel multiplies padding row bands too; others: probe builds only */
} v2a_tuning;
int v2a_set_tuning(const v2a_tuning* tuning);

/* ---------------------------------------------------------------------------------------
 * RMSNorm / AdaptiveRMSNorm:  y = x / max(|x|_2, 1e-12) * sqrt(d) * gamma
 * gamma is a step vector: RMSNorm passes g (strides 0); AdaptiveRMSNorm passes the
 * precomputed (to_gamma(c) + 1) table.  Output in the compute dtype (GEMM operand); y_dtype V2A_BF16_SPLIT writes the
 * hi / lo planes of the split layout (ldy >= 2*d).
 * Replaces: xt RMSNorm (x3:880,883,913,916,935), xt AdaptiveRMSNorm (x3:807,812,816).
 * ------------------------------------------------------------------------------------- */
int v2a_rmsnorm(const float* x, int64_t ldx, void* y, int64_t ldy, int32_t y_dtype,
                int64_t rows, int32_t d,
                const float* gamma, const int32_t* step, int64_t gamma_step_stride,
                int64_t gamma_batch_stride, int32_t rows_per_batch, v2a_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Position-generating depthwise conv, fused with mask, SiLU and the caller's residual:
 *   out[b,n,:] = x[b,n,:] + m[b,n] * silu(bias + sum_j wt[j,:] * (m*x)[b, n+j-k/2, :])
 * m[b,n] = n < len[b] (len NULL = all valid).  wt is the Conv1d weight transposed to
 * [k][d].  out may alias x only if out == x is NOT used (the kernel reads a halo): pass a
 * different buffer.
 * Replaces: DepthwiseConv x3:495-528 + the `+ x` at x3:1082,1097,1122.
 * ------------------------------------------------------------------------------------- */
int v2a_dwconv_silu_residual(const float* x, float* out, const float* wt, const float* bias,
                             int32_t B, int32_t N, int32_t d, int32_t ksize,
                             const int32_t* len, v2a_stream_t stream);
/* The same with the RMSNorm that follows it folded in (x3:1083,1098,1126): additionally
 *   out_bf16[b,n,c] = bf16(out[b,n,c] * gamma[c]),  gamma = norm_gamma + step[0] * step_stride + b * batch_stride
 *   norm_ssq[(b*N + n) * ld_ssq + j] = sum of out[b,n,32j..32j+31]^2
 * for the GEMM that consumes out_bf16 with row_ssq (v2a_gemm_args).  d % 32 == 0. */
typedef struct v2a_dwconv_norm {
  void* out_bf16;
  int64_t ld_out_bf16;
  const float* norm_gamma;
  const int32_t* step;
  int64_t norm_step_stride, norm_batch_stride;
  float* norm_ssq;
  int64_t ld_norm_ssq;
  int32_t split;          /* non-zero: out_bf16 rows in the V2A_BF16_SPLIT layout [hi d | lo d] (ld_out_bf16 >= 2 * d): bf16x3 mode */
  int32_t reserved;
} v2a_dwconv_norm;
int v2a_dwconv_silu_residual_norm(const float* x, float* out, const float* wt, const float* bias,
                                  int32_t B, int32_t N, int32_t d, int32_t ksize,
                                  const int32_t* len, const v2a_dwconv_norm* norm, v2a_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Rotary embedding applied in place to `nheads` consecutive 64-wide heads of every row
 * (the q and k column blocks of the fused QKV GEMM output).
 * position of row r = pos_offset + (r % rows_per_batch); table cs[pos][32][2] = (cos,sin)
 * of pos * 10000^(-2i/64).  layout 0 = interleaved pairs (2i,2i+1), 1 = half split (i,i+32).
 * Replaces: xt RotaryEmbedding.forward_from_seq_len + apply_rotary_pos_emb
 * (x3:779-781,983,988,994 and the rotary_pos_emb argument at x3:1084,1099,1126,1131).
 * ------------------------------------------------------------------------------------- */
int v2a_rope_inplace(void* qk, int32_t dtype, int64_t rows, int64_t row_stride, int32_t nheads,
                     int32_t rows_per_batch, int32_t pos_offset, const float* cs_table,
                     int32_t layout, v2a_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Soft-clamped, key-masked, head-gated attention core (dim_head = 64):
 *   s = clamp * tanh(scale * q.k / clamp);  p = softmax_j(s | j < kv_len[b]);
 *   o[b,i,h,:] = sigmoid(gate[b,i,h]) * sum_j p_j v[b,j,h,:];  rows i >= q_len[b] -> 0
 * q/k/v/gate/out are addressed as base + b*batch_stride + token*row_stride + h*64 (+c)
 * (gate: + h), so they can live inside the fused [q|k|v|gate] GEMM output.
 * Replaces: xt Attention forward minus its Linears (attend + to_v_head_gate epilogue +
 * out.masked_fill), sites x3:808,813,881,914,1084,1099,1126,1131.
 * ------------------------------------------------------------------------------------- */
typedef struct v2a_attn_args {
  const void *q, *k, *v, *gate;
  void* out;
  int64_t q_row_stride, k_row_stride, v_row_stride, gate_row_stride, out_row_stride;
  int64_t q_batch_stride, k_batch_stride, v_batch_stride, gate_batch_stride, out_batch_stride;
  int32_t B, H, Nq, Nk;
  const int32_t* kv_len; /* [B] or NULL (= Nk) */
  const int32_t* q_len;  /* [B] or NULL (= Nq) */
  float scale, softclamp;
  int32_t dtype;         /* V2A_F32 / V2A_BF16: dtype of q,k,v,gate,out and of the arithmetic; V2A_BF16_SPLIT: fp32 tensors,
                          * products as three bf16 MFMA passes over hi | lo operand planes (the bf16x3 mode) */
  int32_t out_split;     /* dtype V2A_BF16_SPLIT only, non-zero: out is a bf16 buffer in the V2A_BF16_SPLIT layout -- row =
                          * [hi of the H*64 outputs | lo of them], out_row_stride / out_batch_stride in bf16 elements -- i.e. the
                          * A operand of the out-projection's split GEMM, written directly */
} v2a_attn_args;

int v2a_attention(const v2a_attn_args* args, v2a_stream_t stream);
/* ---------------------------------------------------------------------------------------
 * Cross-attention of the audio stream in ONE launch (ABI 6): the q-projection GEMM of v2a_gemm -- STORE epilogue with the optional
 * row_ssq consumer scale, bias and fused RoPE -- whose 64-token x one-head output tile never leaves the workgroup: it is the Q
 * operand of QK^T over the clip's Nk <= 64 context keys, soft clamp, softmax, PV, per-head sigmoid gate, and only the attention
 * output rows are written.  Equal bit for bit to v2a_gemm(gemm) into a [q | gate] buffer followed by v2a_attention(attn) on it.
 *   gemm: nseg = 1, bf16 A and W, ka[0] a multiple of 512; W rows [0, H*64) = to_q, rows [H*64, H*64 + H) = the per-head gate
 *         rows (N >= H*65), bias likewise; M = attn->B * attn->Nq rows, rows_per_batch = attn->Nq; out / ldo / out_dtype ignored.
 *   attn: dtype V2A_BF16; k, v, out and their strides, B, H, Nq, Nk, kv_len, q_len, scale, softclamp as for v2a_attention;
 *         q, gate and their strides ignored.
 * Replaces: xt Attention.forward with a context -- to_q, rotary_emb on q, attend, to_v_gates -- at call site x3:1126 (audio
 * stream of a layer with text context), for launches small enough to be latency bound (the sampler: <= 2 clips per GPU).
 * ------------------------------------------------------------------------------------- */
int v2a_qproj_xattn(const v2a_gemm_args* gemm, const v2a_attn_args* attn, v2a_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Small fp32 linear with output row scatter (K arbitrary, VALU):
 *   out[(m / T)*out_batch_stride + (row_off + m % T)*d + n] =
 *        bias[n] + add[(m % T)*d + n] + sum_k a[m*K + k] * wt[k*d + n]
 * and, when `dup_batch_offset` > 0, the same value again at batch (m/T + dup_batch_offset)
 * (cond and null CFG halves share proj_in(x) + abs_pos_emb).
 * Replaces: proj_in x3:2027 + abs_pos_emb x3:957-960 + register pack x3:975-976;
 * proj_frames x3:2069.
 * ------------------------------------------------------------------------------------- */
int v2a_linear_small(const float* a, int64_t M, int32_t K, const float* wt, const float* bias,
                     const float* add, int32_t T, float* out, int64_t out_batch_stride,
                     int32_t row_off, int32_t d, int32_t dup_batch_offset,
                     const float* regs /* NULL, or [row_off][d]: also writes rows [0,row_off) = regs */,
                     void* out_bf16 /* NULL, or bf16 shadow with out's geometry */, v2a_stream_t stream);

/* out[b, r, :] = regs[r, :] for r < R, b < B  (register tokens, x3:975-997) */
int v2a_fill_registers(float* out, int64_t out_batch_stride, const float* regs, int32_t B,
                       int32_t R, int32_t d, v2a_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Time conditioning for every grid point at once:
 *   out[s,:] = silu(bias + Wt^T [t_s, sin(2 pi t_s w), cos(2 pi t_s w)])
 * wt = Linear(d+1, d).weight transposed to [d+1][d].
 * Replaces: RandomFourierEmbed x3:555-564 + time_cond_mlp x3:793-797,966-971.
 * ------------------------------------------------------------------------------------- */
int v2a_time_cond(const float* t, int32_t S, const float* fourier_w, const float* wt,
                  const float* bias, float* out, int32_t d, v2a_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * CFG combine + explicit Euler update, in place on y (B, T, C):
 *   f = pc + s*(pc - pn);  y += dt[step] * f
 * pc = pred[b, row_off + i, :], pn = pred[B + b, row_off + i, :]  (pred is (2B, Np, C)).
 * With apg != NULL (remove_parallel_component): apg[b] = {sum (pc-pn)*pc, sum pc*pc} in
 * fp64 from v2a_apg_reduce and   upd = orth + keep*par.
 * Replaces: cfg_transformer_with_pred_head x3:2106-2113, project x3:162-173,
 * torchdiffeq Euler step (x3:2255).
 * ------------------------------------------------------------------------------------- */
int v2a_apg_reduce(const float* pred, double* apg, int32_t B, int32_t T, int32_t C,
                   int64_t pred_batch_stride, int32_t row_off,
                   const int32_t* valid_rows /* ABI 7: device int or NULL (= T): only rows [0, valid_rows[0]) of every clip enter the sums --
                                                the frames of the call, when the plan's T is padded to a shape bucket */,
                   v2a_stream_t stream);
int v2a_cfg_euler(float* y, const float* pred, int32_t B, int32_t T, int32_t C,
                  int64_t pred_batch_stride, int32_t row_off, float cfg_strength,
                  const float* dt, const int32_t* step, const double* apg,
                  float keep_parallel_frac, v2a_stream_t stream);
/* y[r][0:d] = hi, y[r][d:2d] = lo of x[r][0:d] (V2A_BF16_SPLIT layout above), rows x d fp32 in, row strides in elements,
 * d % 4 == 0: the split operand copy of an fp32 buffer (residual streams, attention outputs, GEGLU hidden) for the
 * three-segment bf16 GEMMs of the bf16x3 mode */
int v2a_split_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t d, v2a_stream_t stream);
/* y[i] = bf16(x[i]), n % 4 == 0: bf16 operand copy of an fp32 stream that no GEMM epilogue produced
 * (the embed output x3:2027 feeding the first cross-condition GEMM) */
int v2a_cast_bf16(const float* x, void* y, int64_t n, v2a_stream_t stream);
/* step[0] += 1 (own launch: every block of the step has read step[0] before it runs) */
int v2a_step_advance(int32_t* step, v2a_stream_t stream);

/* =======================================================================================
 * N2: Video2Roll frame encoder -- `E2TTS.encode_frames` x3:1525-1553 running
 * `Video2RollNet.resnet18` (`v2r` = src/audeo/Video2RollNet.py:127-251).  Activations are NHWC fp32;
 * every convolution is v2a_im2col (patch matrix in the compute dtype) + v2a_gemm with the eval-mode BatchNorm
 * folded into the weights / bias and ReLU / residual in the epilogue.
 * ===================================================================================== */

/* Patch matrix of a 2-D convolution:  col[(n*Ho + yo)*Wo + xo][k],  zero outside the image and for k >= K.
 *   window_t == 0: x is NHWC fp32 (n < B), k = (ky*kw + kx)*C + c, C % 4 == 0
 *                  (replaces the unfold of nn.Conv2d at v2r:9-12,18-19,138,187-188)
 *   window_t  > 0: x is (B / window_t clips, window_t frames, H, W) single-channel fp32 and the C = 5 input channels of
 *                  window n = (clip, i) are frames clamp(i-2 .. i+2, 0, window_t-1) of that clip -- the 5-frame stack
 *                  built at x3:1531-1539 is never materialised; k = (c*kh + ky)*kw + kx (the conv1 weight's own order)
 * ldo = padded K (multiple of 64 for bf16, 16 for f32), out_dtype = V2A_F32 | V2A_BF16. */
int v2a_im2col(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t stride,
               int32_t pad, int32_t Ho, int32_t Wo, void* col, int64_t ldo, int32_t out_dtype, int32_t window_t,
               int32_t window_first, v2a_stream_t stream);

/* First-layer operand of the implicit GEMM (bf16): one clip's frames (T, H, W) fp32 -> column patches
 *   out[j][xo][y][e],  j < T + 4, xo < Wo, y < H + 2*pad, e < 16   (bf16; (T+4)*Wo*(H+2*pad)*16 elements)
 *   = frame clamp(j - 2, 0, T - 1) at row y - pad, column stride*xo - pad + e; zero outside the image and for e >= kw.
 * The 5-frame window i of x3:1531-1539 is frames j = i .. i+4 of this replicate-padded clip, and the kh x kw taps of output
 * pixel (yo, xo) in channel c are the kh consecutive 16-element rows from out[i + c][xo][stride*yo]: with
 *   a_row_offset[m]    = ((i*Wo + xo)*Hp + stride*yo) * 16
 *   a_ktile_offset[kt] = (kt / g) * Wo*Hp*16 + (kt % g) * 64,   g = ceil(kh / 4) K tiles per channel (weights zero-padded)
 * v2a_gemm computes conv1 (v2r:138, 11x11 / stride 2 / pad 4) without the patch matrix of v2a_im2col. */
int v2a_frames_pack(const float* frames, void* out, int32_t T, int32_t H, int32_t W, int32_t kw, int32_t stride, int32_t pad,
                    int32_t Wo, v2a_stream_t stream);

/* NHWC fp32 pooling: mode 0 = max (padding acts as -inf; nn.MaxPool2d(3, 2, 1) v2r:141), mode 1 = average over the full
 * k*k window (nn.AvgPool2d(2, 2) / (3, 1), pad 0, v2r:22-23).  C % 4 == 0.  The input / output maps may be stored with a
 * zero border of in_border / out_border pixels (the implicit-GEMM layout of v2a_gemm's offset tables): only interiors are
 * read / written.  out_bf16 (or NULL) receives a bf16 copy with out's geometry (operand of the next convolution). */
int v2a_pool2d(const float* x, float* out, void* out_bf16, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k,
               int32_t stride, int32_t pad, int32_t mode, int32_t Ho, int32_t Wo, int32_t in_border, int32_t out_border,
               v2a_stream_t stream);

/* Fused top of the network (v2r:224-249 + the sigmoid of x3:1541), one workgroup per window:
 *   FRB4/3/2 channel gates (global average pool -> fc1 -> ReLU -> fc2 -> sigmoid, v2r:44-57), out1 = p2*p3, softmax over
 *   the P positions per channel, out2 = softmax*p4, conv2 (1x1) + p4, global average pool, fc, optional sigmoid.
 * The 1x1 conv and the pool commute, so only per-channel position sums are formed.  All tensors fp32; x2_, x3_, x4_ are
 * (B, P, 128) and x5 is (B, P, 64) NHWC maps; weights are TRANSPOSED ([in][out]) so lanes read consecutive outputs. */
typedef struct v2a_roll_head_args {
  const float *x2, *x3, *x4, *x5;
  int32_t B, P;
  const float *frb4_w1t, *frb4_b1, *frb4_w2t, *frb4_b2;   /* [192][128], [128], [128][128], [128] */
  const float *frb3_w1t, *frb3_b1, *frb3_w2t, *frb3_b2;   /* [256][128] ...                        */
  const float *frb2_w1t, *frb2_b1, *frb2_w2t, *frb2_b2;   /* [256][128] ...                        */
  const float *conv2_wt, *conv2_b;                        /* [128][128], [128]                     */
  const float *fc_wt, *fc_b;                              /* [128][notes], [notes]                 */
  int32_t notes;                                          /* <= 128 (51: NOTES, x3:1523)           */
  int32_t apply_sigmoid;                                  /* 1: probabilities (encode_frames), 0: logits (ResNet.forward) */
  float* out;                                             /* (B, notes)                            */
} v2a_roll_head_args;
int v2a_roll_head(const v2a_roll_head_args* args, v2a_stream_t stream);

/* roll (B, t, notes) -> out (B, l, notes): every frame row repeated `rep` (3) times, cropped / zero-padded to l rows
 * (x3:1544-1553) */
int v2a_roll_expand(const float* roll, float* out, int32_t B, int32_t t, int32_t notes, int32_t rep, int32_t l,
                    v2a_stream_t stream);

/* =======================================================================================
 * N1: Encodec 24 kHz decoder, the vocoder behind `EncodecWrapper.decode` x3:434-437 (predict.py:277-278;
 * arithmetic: transformers EncodecDecoder, requirements.txt:20).  Activations are time-major [T][C] fp32, so every
 * causal Conv1d is a v2a_gemm over overlapping rows (lda = C, K = k*C) and every ConvTranspose1d(k = 2*stride) a v2a_gemm
 * with K = 2C, N = stride*Cout whose output IS the up-sampled [T*stride][Cout] signal.  Two kernels complete the stack:
 * ===================================================================================== */

/* out[(pad + t)][c] = act ? ELU(x[t][c]) : x[t][c] for t < T, preceded by `pad` rows: reflect != 0 -> row (pad - i) mirrors
 * row i (the causal reflect padding of EncodecConv1d: F.pad(x, (k-1, 0), "reflect")), else zeros (the x[q-1] operand of a
 * transposed convolution at q = 0).  out holds (T + pad) rows.  C % 4 == 0.  Replaces nn.ELU + the padding of every conv. */
int v2a_elu_pad(const float* x, float* out, int64_t T, int32_t C, int32_t pad, int32_t reflect, int32_t act,
                v2a_stream_t stream);

/* One nn.LSTM layer's recurrence over T steps (batch 1, hidden H = 512, zero initial state), gates in torch order (i,f,g,o):
 *   g_t = gates_x[t] + W_hh h_{t-1};  c_t = sig(f) c_{t-1} + sig(i) tanh(g);  h_t = sig(o) tanh(c_t)
 * gates_x[t] = W_ih x_t + b_ih + b_hh comes from a v2a_gemm.  h (T, H) receives every h_t; with y != NULL also
 * y[t] = h_t + resid[t] (the skip of EncodecLSTM).  Persistent kernel of H/8 workgroups with W_hh in registers; h_t travels
 * between workgroups as 64-bit {value, step tag} words (device-scope atomic store / polled load), one round trip per step.
 * workspace = 4*H + 2 int32, 8-byte aligned (zeroed by the call; workspace[4*H] != 0 afterwards means a workgroup timed
 * out waiting for its peers and the result is invalid). */
int v2a_lstm_layer(const float* gates_x, const float* w_hh, float* h, const float* resid, float* y, int32_t T, int32_t H,
                   int32_t* workspace, v2a_stream_t stream);

/* Both layers of EncodecLSTM (2-layer nn.LSTM + skip, batch 1, H = 512) in one persistent kernel: pipeline step s runs layer 0
 * at time s and layer 1 at time s - 1, both fed by values published in step s - 1, so the sequence costs T + 1 exchange round
 * trips instead of 2T and layer 1's input projection (w_ih1, bias1 = b_ih1 + b_hh1) is done in the kernel.
 *   y[t] = h1[t] + resid[t]   (resid may be NULL).   gates_x0[t] = W_ih0 x_t + b_ih0 + b_hh0 from a v2a_gemm.
 * workspace = 8*H + 2 int32, 8-byte aligned (zeroed by the call; workspace[8*H] != 0 afterwards = a workgroup timed out). */
int v2a_lstm2(const float* gates_x0, const float* w_hh0, const float* w_ih1, const float* bias1, const float* w_hh1,
              const float* resid, float* y, int32_t T, int32_t H, int32_t* workspace, v2a_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* V2A_CFM_H */
