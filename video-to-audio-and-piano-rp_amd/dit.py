"""Host side of the three-stream DiT (audio / CLIP-text / piano-frames) on MI355X.

Mirrors `Transformer.forward` (x3:941-1143) + `transformer_with_pred_head` (x3:1993-2088)
of the reference as a fixed launch sequence over the C-ABI kernels of include/v2a_cfm.h.
`x3` = /root/reference/src/e2_tts_pytorch/e2_tts_crossatt3.py.

What is restructured relative to the reference (results unchanged):
  * cond + null CFG passes run as ONE batch of 2B sequences (x3:2099,2104 run them in turn);
  * the FLAN-T5 output is an input (`context`), its per-layer cross-attention K/V (+RoPE on K)
    are computed once per sample() instead of 2*(steps-1) times (x3:2057);
  * cross-attention is skipped for the null half: context == 0 and bias-free to_k/to_v/to_out
    make its output exactly 0 (x3:2058-2062);
  * every AdaptiveRMSNorm / AdaLNZero modulation vector of every layer and every Euler grid
    point is one fp32 GEMM at prepare() time (they depend on t only, x3:966-971);
  * layer 0's text and frames blocks do not depend on x or t (cross-conditioning happens
    after them, x3:1081-1104) and are hoisted out of the Euler loop;
  * torch.cat of cross-condition / skip inputs is replaced by multi-segment GEMM operands.

HBM layout: residual streams are fp32 (Bt, N, d) row-major with N = registers + frames;
GEMM operands (norm outputs, q|k|v|gate, attention outputs, GEGLU hidden) are in the compute
dtype (bf16, or fp32 in parity mode) and live in per-stream scratch buffers sized once per
(B, T) plan, so a whole Euler step is allocation-free and hipGraph-capturable.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, asdict

import torch

from . import _lib as L

NOTES = 51  # x3:70


@dataclass
class DiTConfig:
    """predict.py:118-154 / x3:707-738 defaults."""
    dim: int = 1024
    dim_text: int = 1280
    dim_frames: int = 512
    depth: int = 12
    heads: int = 16
    dim_head: int = 64
    frames_heads: int = 8           # x3:914
    ff_mult: int = 4
    kernel_size: int = 31
    num_registers: int = 32
    num_channels: int = 128
    notes: int = NOTES
    max_seq_len: int = 8192
    dim_context: int | None = None
    cond_proj_in: bool = False      # E2TTS(if_cond_proj_in=...), x3:1365: the audio-prompt projection (False in every shipped config)

    @property
    def ctx_dim(self):
        return self.dim if self.dim_context is None else self.dim_context

    def to_dict(self):
        return asdict(self)


def _ru(x, m):
    return (x + m - 1) // m * m


def pack_weight(w, dev, cd, split=False, ksegs=None):
    """nn.Linear weight [N][K] -> device operand.  Plain modes: one tensor in the compute dtype.  split (bf16x3 mode): [N][2K] bf16 =
    [W_hi | W_lo] with W_hi = bf16(W), W_lo = bf16(W - W_hi): against A rows [A_hi | A_lo] the split GEMM (v2a_gemm, a_dtype
    V2A_BF16_SPLIT) sums A_lo W_hi + A_hi W_lo + A_hi W_hi in fp32 per K step, over all logical K segments in one launch (`ksegs` is
    kept for the callers' documentation of the concatenated inputs; the layout does not depend on it)."""
    w = w.float()
    if not split:
        return w.to(dev, cd).contiguous()
    hi = w.bfloat16()
    lo = (w - hi.float()).bfloat16()
    return torch.cat([hi, lo], 1).contiguous().to(dev)


class _Attn:
    """Packed weights of one x-transformers Attention (A1, A5 of SURVEY 8c)."""

    def __init__(self, sd, prefix, dim, heads, dh, cd, dev, cross=False, split=False):
        inner = heads * dh
        wq, wk, wv = sd[f"{prefix}.to_q.weight"], sd[f"{prefix}.to_k.weight"], sd[f"{prefix}.to_v.weight"]
        wg, bg = sd[f"{prefix}.to_v_head_gate.weight"], sd[f"{prefix}.to_v_head_gate.bias"]
        self.heads, self.inner = heads, inner
        parts = [wq, wg] if cross else [wq, wk, wv, wg]
        n = sum(p.shape[0] for p in parts)
        self.n_pad = _ru(n, 16)
        w = torch.zeros(self.n_pad, dim, dtype=torch.float32)
        w[:n] = torch.cat([p.float() for p in parts], 0)
        b = torch.zeros(self.n_pad, dtype=torch.float32)
        self.gate_col = n - heads
        b[self.gate_col:n] = bg.float()
        self.w_in = pack_weight(w, dev, cd, split)      # [q|k|v|gate] or [q|gate] rows
        self.b_in = b.to(dev)
        self.w_out = pack_weight(sd[f"{prefix}.to_out.weight"], dev, cd, split)
        self.wk, self.wv = (wk, wv) if cross else (None, None)


class _FF:
    """Packed GEGLU feed-forward (A8): W1 rows regrouped [16 value | 16 gate] per 16 outputs."""

    def __init__(self, sd, prefix, dim, cd, dev, split=False):
        w1, b1 = sd[f"{prefix}.ff.0.proj.weight"].float(), sd[f"{prefix}.ff.0.proj.bias"].float()
        inner = w1.shape[0] // 2
        assert inner % 16 == 0
        self.inner = inner
        wv, wg = w1[:inner].reshape(inner // 16, 1, 16, dim), w1[inner:].reshape(inner // 16, 1, 16, dim)
        self.w1 = pack_weight(torch.cat([wv, wg], 1).reshape(2 * inner, dim), dev, cd, split)
        bv, bgt = b1[:inner].reshape(inner // 16, 1, 16), b1[inner:].reshape(inner // 16, 1, 16)
        self.b1 = torch.cat([bv, bgt], 1).reshape(2 * inner).to(dev).contiguous()
        self.w2 = pack_weight(sd[f"{prefix}.ff.2.weight"], dev, cd, split)
        self.b2 = sd[f"{prefix}.ff.2.bias"].float().to(dev).contiguous()


class _Conv:
    def __init__(self, sd, prefix, dev):
        w = sd[f"{prefix}.dw_conv1d.0.weight"].float()          # (d, 1, k)
        self.k = w.shape[-1]
        self.wt = w[:, 0, :].t().contiguous().to(dev)            # [k][d]
        self.b = sd[f"{prefix}.dw_conv1d.0.bias"].float().to(dev).contiguous()


class PackedWeights:
    """Reference state_dict (key layout of x3:824-933) -> device blobs."""

    def __init__(self, cfg: DiTConfig, sd: dict, dev, cd: torch.dtype, split: bool = False):
        c = cfg
        pk = lambda w, ksegs=None: pack_weight(w, dev, cd, split, ksegs)
        T = "transformer"
        f32 = lambda k: sd[k].float().to(dev).contiguous()
        self.pos_emb = f32(f"{T}.abs_pos_emb.weight")
        self.regs = f32(f"{T}.registers")
        self.text_regs = f32(f"{T}.text_registers")
        self.frames_regs = f32(f"{T}.frames_registers")
        self.fourier_w = f32(f"{T}.time_cond_mlp.0.weights")
        self.time_wt = sd[f"{T}.time_cond_mlp.1.weight"].float().t().contiguous().to(dev)   # [d+1][d]
        self.time_b = f32(f"{T}.time_cond_mlp.1.bias")
        self.final_g = f32(f"{T}.final_norm.g")
        self.pin_wt = sd["proj_in.weight"].float().t().contiguous().to(dev)                  # [C][d]
        self.pin_b = f32("proj_in.bias")
        self.pf_wt = sd["proj_frames.weight"].float().t().contiguous().to(dev)               # [51][df]
        self.pf_b = f32("proj_frames.bias")
        self.pred_w = pk(sd["to_pred.weight"])
        self.pred_b = f32("to_pred.bias")
        if c.cond_proj_in:          # audio-prompt projection (x3:2034); K zero-padded to the GEMM's K granule
            wc = sd["cond_proj_in.weight"].float()
            self.cond_k = _ru(c.num_channels, 64)
            wp = torch.zeros(wc.shape[0], self.cond_k)
            wp[:, :c.num_channels] = wc
            self.cond_w = pk(wp)
            self.cond_b = (sd["cond_proj_in.bias"].float() if "cond_proj_in.bias" in sd else torch.zeros(wc.shape[0])).to(dev).contiguous()
        self.layers = []
        ng, gw, gb, kw, vw = [], [], [], [], []
        for i in range(c.depth):
            P = f"{T}.layers.{i}"
            ly = {}
            if i >= c.depth // 2:
                ly["skip"] = pk(sd[f"{P}.0.0.weight"], [c.dim, c.dim])
                if cd == torch.bfloat16:
                    # cross-condition and U-Net skip of the second half as ONE GEMM (bf16 mode, and since round 5 bf16x3: both Linears are bias-free):
                    #   skip_proj(cat(x + W1 [x; t; f], s)) = Ws_x (I + W1_a) x + Ws_s s + (Ws_x W1_t) t + (Ws_x W1_f) f
                    # K = 3840 instead of 2816 + 2048, one launch instead of two on the audio stream's critical path.  Column
                    # order [x | s | t | f]: x and s are the two halves of one 2048-wide bf16 operand buffer (DiTEngine).
                    ws = sd[f"{P}.0.0.weight"].double()
                    w1 = sd[f"{P}.1.5.text_frames_to_audio.weight"].double()
                    wsx, wss = ws[:, :c.dim], ws[:, c.dim:]
                    fused = torch.cat([wsx @ w1[:, :c.dim] + wsx, wss, wsx @ w1[:, c.dim:c.dim + c.dim_text], wsx @ w1[:, c.dim + c.dim_text:]], 1)
                    ly["x_skip"] = pk(fused.float(), [2 * c.dim, c.dim_text, c.dim_frames])
            ly["a_conv"] = _Conv(sd, f"{P}.0.1", dev)
            ly["a_attn"] = _Attn(sd, f"{P}.0.3", c.dim, c.heads, c.dim_head, cd, dev, split=split)
            ly["a_attn2"] = _Attn(sd, f"{P}.0.6", c.dim, c.heads, c.dim_head, cd, dev, cross=True, split=split)
            ly["a_ff"] = _FF(sd, f"{P}.0.9", c.dim, cd, dev, split)
            ng += [sd[f"{P}.0.2.to_gamma.weight"], sd[f"{P}.0.5.to_gamma.weight"], sd[f"{P}.0.8.to_gamma.weight"]]
            gw += [sd[f"{P}.0.4.to_gamma.weight"], sd[f"{P}.0.7.to_gamma.weight"], sd[f"{P}.0.10.to_gamma.weight"]]
            gb += [sd[f"{P}.0.4.to_gamma.bias"], sd[f"{P}.0.7.to_gamma.bias"], sd[f"{P}.0.10.to_gamma.bias"]]
            kw.append(ly["a_attn2"].wk)
            vw.append(ly["a_attn2"].wv)
            ly["t_conv"] = _Conv(sd, f"{P}.1.0", dev)
            ly["t_g1"] = f32(f"{P}.1.1.g")
            ly["t_attn"] = _Attn(sd, f"{P}.1.2", c.dim_text, c.heads, c.dim_head, cd, dev, split=split)
            ly["t_g2"] = f32(f"{P}.1.3.g")
            ly["t_ff"] = _FF(sd, f"{P}.1.4", c.dim_text, cd, dev, split)
            ly["x_tfa"] = pk(sd[f"{P}.1.5.text_frames_to_audio.weight"], [c.dim, c.dim_text, c.dim_frames])
            if i != c.depth - 1:
                ly["x_at"] = pk(sd[f"{P}.1.5.audio_to_text.weight"], [c.dim, c.dim_text])
                ly["x_af"] = pk(sd[f"{P}.1.5.audio_to_frames.weight"], [c.dim, c.dim_frames])
            ly["f_conv"] = _Conv(sd, f"{P}.2.0", dev)
            ly["f_g1"] = f32(f"{P}.2.1.g")
            ly["f_attn"] = _Attn(sd, f"{P}.2.2", c.dim_frames, c.frames_heads, c.dim_head, cd, dev, split=split)
            ly["f_g2"] = f32(f"{P}.2.3.g")
            ly["f_ff"] = _FF(sd, f"{P}.2.4", c.dim_frames, cd, dev, split)
            self.layers.append(ly)
        # modulation tables: rows ordered (layer, slot) with slot = attn / cross-attn / ff
        self.norm_gamma_w = torch.cat([w.float() for w in ng], 0).to(dev).contiguous()       # (L*3*d, d) fp32
        self.norm_gamma_b = torch.ones(self.norm_gamma_w.shape[0], device=dev)               # the "+1" of A10
        self.gate_w = torch.cat([w.float() for w in gw], 0).to(dev).contiguous()
        self.gate_b = torch.cat([b.float() for b in gb], 0).to(dev).contiguous()
        # cross-attention K/V projections of every layer stacked: [K_0..K_{L-1} | V_0..V_{L-1}]
        self.ctx_kv_w = pk(torch.cat([w.float() for w in kw] + [w.float() for w in vw], 0))
        for ly in self.layers:
            ly["a_attn2"].wk = ly["a_attn2"].wv = None


# ---- measured tile choices, in ONE place -----------------------------------------------------------------------------------------
# key: (compute mode, regime, (dim, dim_text, dim_frames, ff_mult)); regime as DiTEngine._regime(): 0 = a launch cannot fill the chip (one
# clip at these widths), 1 = about fills it (two clips), 2 = fills it several times over.  value: {(stream, op): (tile, log)} where
# `tile` is a tile configuration of v2a_gemm (bf16: the k of tile_hint = k + 1; bf16x3: the split-operand shape 1..5 = tile_hint) and `log`
# names the A/B record under profiles/ that justifies the entry.  streams a / t / f; ops: x_tfa skip qkv out q2 out2 ff1 ff2 (audio),
# cross qkv out ff1 ff2 (text, frames).  A (mode, regime, widths) without a row, and an op without an entry, take the stream's policy
# tile (DiTEngine.side_tile / main_tile: fat 128x256 tiles on the side streams, the library's choice on the audio stream) -- correct
# everywhere, measured only where a row exists.  bench.py --side-tiles / --big-tiles / --main-tile override entries for A/B runs.
SHIPPED_WIDTHS = (1024, 1280, 512, 4)
_R2, _R3 = "profiles/r02 A/B runs (DESIGN.md appendix A, `scripts/gpu_ci.sh sidetiles`)", "profiles/r03_tile_sweep.txt"
_R5 = "profiles/r05_bf16x3_tile_sweeps.txt"
TUNED_TILES = {
    ("bf16", 0, SHIPPED_WIDTHS): {
        # text stream co-critical with the audio stream: its 1280-wide GEMMs on 128x128 tiles with eight waves (tile 12), feed-forward in on
        # the 256x256 8-phase kernel (tile 6)
        ("t", "cross"): (12, _R3), ("t", "out"): (12, _R3), ("t", "ff2"): (12, _R3), ("t", "ff1"): (6, _R2),
        ("f", "cross"): (12, _R3), ("f", "out"): (12, _R3), ("f", "ff2"): (12, _R3), ("f", "ff1"): (6, _R2), ("f", "qkv"): (6, _R2),
        # audio stream: narrow outputs on 128x64 tiles with eight waves (tile 14), QKV on 128x128 (tile 1), feed-forward in on the 8-phase kernel
        ("a", "x_tfa"): (14, _R3), ("a", "skip"): (14, _R3), ("a", "out"): (14, _R3), ("a", "out2"): (14, _R3), ("a", "ff2"): (14, _R3),
        ("a", "qkv"): (1, _R2), ("a", "ff1"): (6, _R2),
    },
    # two clips: the one-clip table costs 4 % here (6214 vs 6467 mel-frames/s, round 2): policy tiles only
    ("bf16", 1, SHIPPED_WIDTHS): {},
    # from three clips on every kernel fills the chip and the library's stand-alone choice wins: an 8-clip sweep of every entry measured <= 0
    ("bf16", 2, SHIPPED_WIDTHS): {},
    # bf16x3, one clip: QKV of the audio / text streams (N = 3088: 91 tiles of 256x256) on the 8-phase kernel's three-segment form instead
    # of the 2-deep 128x128 split ring (90 us per launch): +1 %; everything else by shape (profiles/r03_split_gemm_probe.txt)
    # round 5: the FRAMES stream (5 % of the flops, 17 % of the CU-time on 64x64 / 64x128 tiles: K = 512 .. 2048) on fat tiles with 32-wide K
    # stages -- narrow GEMMs and feed-forward in on 128x256 (26 / 208 workgroups), QKV on 128x128: its chain has the slack (124 us per layer
    # stand-alone against the audio chain's 345), the CU-time it frees goes to the audio chain: +2.7 % (profiles/r05_bf16x3_tile_sweeps.txt,
    # sweeps 4 - 6).  The same on the text stream loses 0.6 - 3 %: its chain is as long as the audio chain's.
    # ... and, once the 8-phase kernel's K loop had lost its address bookkeeping, the frames stream's two WIDE GEMMs on it: QKV
    # (N = 1552: 49 workgroups instead of 91) and feed-forward in (N = 4096: 112 instead of 208), the cheapest tile per flop in chip-time: +0.9 %
    # (sweep 7 of the same file).  Its narrow GEMMs stay on the ring: 14 tiles of 256x256 take 125 us each and the chain's slack is gone.
    ("bf16x3", 0, SHIPPED_WIDTHS): {("a", "qkv"): (5, "profiles/r04_bf16x3_qkv_8phase_ab.txt"), ("t", "qkv"): (5, "profiles/r04_bf16x3_qkv_8phase_ab.txt"),
                                    ("f", "qkv"): (5, _R5), ("f", "ff1"): (5, _R5), ("f", "cross"): (6, _R5), ("f", "out"): (6, _R5), ("f", "ff2"): (6, _R5)},
    # bf16x3, launches that fill the chip several times over (8 clips per GPU): no entry.  The QKV projections on 128x256 tiles with 32-wide K
    # stages are 4-12 % faster stand-alone than on the 8-phase kernel (228 / 280 / 80 against 240 / 290 / 91 us: 637 tiles of 256x256 are 2.49
    # rounds, the 13th tile column holds 16 gate columns) and 1.2 % SLOWER in the sampler (3566 against 3608 mel-frames/s,
    # profiles/r05_bf16x3_8clips_ab.txt): beside the other two queues the tail rounds are filled anyway and the 8-phase tile is the cheaper one per flop
    ("bf16x3", 2, SHIPPED_WIDTHS): {},
}


def tuned_tiles(mode: str, regime: int, widths) -> dict:
    """{(stream, op): tile} of a measured (mode, regime, widths) row, {} when there is none."""
    return {k: v[0] for k, v in TUNED_TILES.get((mode, regime, tuple(widths)), {}).items()}


_STREAMS: dict = {}


def process_streams(dev):
    """(text side stream, frames side stream, graph-capture stream) of a device, created ONCE per process.

    torch hands out streams from a pool of 32 per device and wraps around: allocating two side streams per plan made a later
    plan's side stream alias torch's graph-capture stream after ~16 plans, and replaying a graph captured that way crashed
    inside hipGraphLaunch.  Three fixed, distinct streams for the whole process cannot alias."""
    dev = torch.device(dev)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    s = _STREAMS.get(key)
    if s is None:
        s = _STREAMS[key] = tuple(torch.cuda.Stream(device=dev) for _ in range(3))
        assert len({x.cuda_stream for x in s}) == 3
    return s


class DiTEngine:
    """One plan = fixed (Bt sequences, T frames, nc context tokens); all buffers preallocated."""

    def __init__(self, cfg: DiTConfig, state_dict: dict, device="cuda", compute: str = "bf16",
                 rope_layout: str = "interleaved", rope_cross: bool = False,
                 zero_masked_queries: bool = True, softclamp: float = 50.0, multi_stream: bool = True):
        assert compute in ("bf16", "fp32", "bf16x3")
        # "bf16x3": GEMM operands as bf16 hi | lo planes, three bf16 MFMA products per fp32 product (hi*hi + hi*lo + lo*hi);
        # attention, norms, conv, residual streams in fp32 as in parity mode -- fp32-grade results at a third of bf16 speed
        self.split = compute == "bf16x3"
        self.multi_stream = multi_stream
        # GEMMs of the text / frames blocks: tile configuration 0 (128x256, one 144 KB workgroup per CU) when they run beside the
        # audio block on side streams -- few fat workgroups that own whole CUs disturb the critical path less than many small
        # ones spread over every CU (+3.5 % end to end); -1 = the library's stand-alone choice
        self.side_tile = 0 if multi_stream else -1
        # per-(stream, op) exceptions to the fat-tile policy.  The text stream is co-critical with the audio stream (its block
        # plus its cross-condition GEMM must fit inside one audio layer): its three 1280-wide GEMMs have only 65 tiles of
        # 128x256 (FF2: 77 us alone) -- on 128x128 tiles (130 workgroups) the sampler is 3.5 % faster at one clip.
        # ... and its feed-forward GEMM (N = 10240: 7 x 40 tiles of 256x256) on the phase-interleaved kernel: another 1.5 %.
        # ("a", op) entries apply to the audio stream's GEMMs (ops: x_tfa skip qkv out q2 out2 ff1 ff2): its feed-forward
        # GEMM on the phase-interleaved kernel too (+2 %), its QKV GEMM on 128x128 tiles (+0.4 %); frames feed-forward on the
        # phase-interleaved kernel (+0.6 %), frames QKV too (49 tiles: +1.7 %).  Each entry A/B-ed alone and in combination on one box (bench.py --side-tiles).
        self.fuse_skip = True           # bf16 mode: cross-condition + skip projection of layers >= depth/2 as one GEMM (PackedWeights)
        # per-(stream, op) tile choices: the measured rows of TUNED_TILES (module level: each entry with the A/B log that justifies it);
        # these dicts are the engine's working copies, which bench.py overrides for A/B runs.  An op without an entry takes the stream's
        # policy tile: `side_tile` on the text / frames streams (set above), `main_tile` on the audio stream's narrow GEMMs (-1 = library).
        widths = (cfg.dim, cfg.dim_text, cfg.dim_frames, cfg.ff_mult)
        one_clip = tuned_tiles("bf16", 0, widths)
        self.side_tiles = {k: v for k, v in one_clip.items() if not (k[0] == "a" and v == 14)}
        self.main_tile = 14 if any(k[0] == "a" and v == 14 for k, v in one_clip.items()) else -1
        self.big_tiles = tuned_tiles("bf16", 2, widths)
        self.split_tiles = tuned_tiles("bf16x3", 0, widths)      # bf16x3 mode: split-operand tile shape 1..7 of v2a_gemm per (stream, op), up to two clips
        self.split_big_tiles = tuned_tiles("bf16x3", 2, widths)  # ... for launches that fill the chip several times over
        # bf16 mode: the RMSNorms of the layer stack are folded into the kernel before them (gamma on the bf16 operand it writes,
        # sums of squares per 32 columns) and the GEMM after them (1 / rms per row in the epilogue): see _fold / _fold_gemm
        self.fold_norm = True
        self.fold_gemm_all = False      # A/B: norms folded into GEMM epilogues at every batch size (default: up to two clips, see _fold_gemm)
        # bf16 and bf16x3 modes, up to three clips (bf16 stand-alone: 19.5 vs 23.2 us at three, 26.7 vs 25.7 at four): the audio stream's cross-attention as ONE launch (v2a_qproj_xattn: q-projection, RoPE, attention over
        # the <= 64 context keys and the head gate without the [q | gate] buffer in between); equal bit for bit to the two launches
        self.fuse_xattn = True
        # x_at / x_af cross-condition GEMMs on the side streams (True: all three on the main stream, -2.2 %)
        self.cross_on_main = False
        # capture order: audio {cross .. self-attention}, sides {conv, norm, attention}, audio {cross-attention, feed-forward},
        # sides {norm, feed-forward} instead of audio block, text block, frames block
        self.interleave_capture = False
        # RoPE rides in the QKV GEMM epilogue when pairs are lane-local (interleaved layout, bf16 DMA kernel)
        self._fuse_rope = compute != "fp32" and rope_layout == "interleaved"
        assert cfg.dim_head == 64, "kernels are built for dim_head = 64 (x3:717)"
        self.cfg = cfg
        self.dev = torch.device(device)
        self.cd = torch.float32 if compute == "fp32" else torch.bfloat16      # dtype of GEMM operand buffers
        self.cdc = L.F32 if compute == "fp32" else L.BF16                      # compute dtype of the GEMMs
        self.ad = torch.bfloat16 if compute == "bf16" else torch.float32      # dtype of q | k | v | gate and attention outputs
        # attention arithmetic: bf16 MFMA / exact fp32 on the VALU / fp32 tensors with split-bf16 MFMA products (bf16x3)
        self.adc = {"bf16": L.BF16, "fp32": L.F32, "bf16x3": L.BF16_SPLIT}[compute]
        self.rope_layout = {"interleaved": 0, "half": 1}[rope_layout]
        self.rope_cross = rope_cross
        self.zero_masked_queries = zero_masked_queries
        self.softclamp = float(softclamp)
        L.lib()  # fail loudly now if the HIP library is absent
        self.W = PackedWeights(cfg, state_dict, self.dev, self.cd, self.split)
        # plans (buffers + the hipGraphs captured on them, see E2TTS._run_steps) by shape, least recently used first: captions and
        # durations vary per clip (predict.py:210-237), and a plan switch otherwise costs ~0.5 GB of allocations, an eager warm-up
        # evaluation and a re-capture.  A B = 1 plan is < 0.5 GB, 8 clips ~4 GB: nothing next to 288 GB of HBM.
        self.plans: OrderedDict = OrderedDict()
        self.max_plans = 4
        self.plan = None

    # ------------------------------------------------------------------------------ planning
    def setup(self, B: int, T: int, nc: int, S: int, cfg_mode: bool = True):
        """B clips, T latent frames, nc context tokens, S time points (Euler evaluations)."""
        key = (B, T, nc, S, cfg_mode)
        p = self.plans.get(key)
        if p is not None:
            self.plans.move_to_end(key)
            self.plan = p
            return p
        c, dev, cd = self.cfg, self.dev, self.cd
        R = c.num_registers
        N = T + R
        if T > c.max_seq_len:       # embed() indexes abs_pos_emb with the frame position (x3:957-960: the reference asserts the same)
            raise ValueError(f"{T} latent frames exceed max_seq_len = {c.max_seq_len} of the position table")
        Bt = 2 * B if cfg_mode else B
        rows = Bt * N
        e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=dev)
        p = dict(key=key, B=B, Bt=Bt, T=T, N=N, nc=nc, S=S, rows=rows, cfg_mode=cfg_mode, graphs={})
        D, Dt, Df = c.dim, c.dim_text, c.dim_frames
        p["xA"], p["xB"], p["xS"] = e(Bt, N, D), e(Bt, N, D), e(Bt, N, D)
        p["skips"] = [e(Bt, N, D) for _ in range(c.depth // 2)]
        p["tA"], p["tB"], p["t0"], p["tL0"] = e(Bt, N, Dt), e(Bt, N, Dt), e(Bt, N, Dt), e(Bt, N, Dt)
        p["fA"], p["fB"], p["f0"], p["fL0"] = e(Bt, N, Df), e(Bt, N, Df), e(Bt, N, Df), e(Bt, N, Df)
        W0 = self.W.layers[0]
        w2 = 2 if self.split else 1           # split operand buffers hold hi | lo planes
        for s, d, attn, ff in (("a", D, W0["a_attn"], W0["a_ff"]), ("t", Dt, W0["t_attn"], W0["t_ff"]),
                               ("f", Df, W0["f_attn"], W0["f_ff"])):
            p[f"hn_{s}"] = e(rows, w2 * d, dt=cd)
            # folded RMSNorm: sums of squares per 32 columns of the row to be normed; rows padded with zeros to whole float4
            p[f"ssq_{s}"] = torch.zeros(rows, (d // 32 + 3) // 4 * 4, device=dev)
            p[f"qkv_{s}"] = e(rows, attn.n_pad, dt=self.ad)
            p[f"ao_{s}"] = e(rows, w2 * attn.inner, dt=cd)
            p[f"ffh_{s}"] = e(rows, w2 * ff.inner, dt=cd)
        # bf16 shadows of the fp32 residual streams (written by the producing GEMM epilogues): the
        # operands of the cross-condition / skip GEMMs, so those run on the LDS-DMA bf16 kernel too
        p["shadow"] = {}
        p["lo_off"] = {}            # bf16x3: split operand views whose lo plane is NOT k elements behind the hi plane (by data pointer)
        if cd == torch.bfloat16:
            mk = lambda t: torch.empty(*t.shape[:-1], w2 * t.shape[-1], dtype=cd, device=dev)
            for name in ("xA", "xB", "tA", "tB", "tL0", "fA", "fB", "fL0"):
                p["shadow"][p[name].data_ptr()] = mk(p[name])
            # [x entering layer depth-1-j | skip j] as one 2048-wide operand buffer per skip: the fused cross-condition +
            # skip GEMM reads both halves as ONE K segment; the halves are written by the FF2 epilogue of layer
            # depth-2-j (x's shadow) and by the cross-condition epilogue of layer j (skip j's shadow).  bf16x3: rows are
            # [x_hi | s_hi | x_lo | s_lo] -- the lo plane of either half lies 2 D further, not D (`lo_off`: v2a_gemm_args.a_lo_offset /
            # out_bf16_lo_offset), so that the buffer is a K = 2 D split segment as a whole and a K = D one half by half
            p["wide"] = [torch.empty(Bt, N, w2 * 2 * D, dtype=cd, device=dev) for _ in p["skips"]]
            for sk, wd in zip(p["skips"], p["wide"]):
                p["shadow"][sk.data_ptr()] = wd[..., D:2 * D]
                if self.split:
                    p["lo_off"][wd.data_ptr()] = 2 * D                       # the whole buffer, or its x half
                    p["lo_off"][wd[..., D:2 * D].data_ptr()] = 2 * D         # its skip half
        p["q2"] = e(B * N, W0["a_attn2"].n_pad, dt=self.ad)
        inner = c.heads * c.dim_head
        p["ctx_kv"] = e(B * nc, 2 * c.depth * inner, dt=self.ad)
        p["ctx"] = e(B * nc, w2 * c.ctx_dim, dt=cd)
        p["ctx_len"] = torch.full((B,), nc, dtype=torch.int32, device=dev)
        p["pred"] = e(Bt, N, c.num_channels)
        p["tc"] = e(S, D)
        p["norm_tab"] = e(S, c.depth, 3, D)
        p["gate_tab"] = e(S, c.depth, 3, D)
        p["t_pts"] = e(S)
        p["dt"] = e(S)
        p["step"] = torch.zeros(1, dtype=torch.int32, device=dev)
        p["apg"] = torch.zeros(2 * B, dtype=torch.float64, device=dev)
        # latent frames of the CALL (<= T when the plan is padded to a shape bucket): the APG sums of x3:162-173 run over the call's own
        # (b, n, C) tensor, not over padding rows.  A device int -- a captured graph bakes scalar arguments, and one graph serves a bucket
        p["valid_T"] = torch.full((1,), T, dtype=torch.int32, device=dev)
        p["seq_len"] = torch.full((Bt,), N, dtype=torch.int32, device=dev)
        p["ragged"] = False
        # rotary table (A6): cos/sin of pos * 10000^(-2i/64), computed like the oracle (CPU fp32)
        inv = 1.0 / (10000 ** (torch.arange(0, c.dim_head, 2).float() / c.dim_head))
        # N + nc rows: with a bucketed plan (E2TTS(bucket_ctx=...)) the cross-attention keys sit at the positions of the
        # UNPADDED lengths (prepare(rope_len=, rope_ctx_len=)), and the masked padding keys behind them still index the table
        ang = torch.arange(N + nc).float()[:, None] * inv[None, :]
        p["rope"] = torch.stack((ang.cos(), ang.sin()), -1).contiguous().to(dev)            # (N + nc, 32, 2)
        p["per_sample_t"] = False
        p["has_cond"] = False
        if c.cond_proj_in:          # operand of the audio-prompt GEMM: (B, N, Kp) rows incl. zero register rows; pos_emb + bias table
            p["condbuf"] = torch.zeros(B * N, w2 * self.W.cond_k, dtype=cd, device=dev)
            p["padd"] = e(T, D)
        # side streams: text block l+1 and frames block l+1 run beside the audio block l (see forward())
        if self.dev.type == "cuda" and self.multi_stream:
            p["st"], p["sf"], _ = process_streams(self.dev)
        self.plan = p
        self.plans[key] = p
        while len(self.plans) > max(1, self.max_plans):
            self.plans.popitem(last=False)          # drops the plan's buffers and graphs
        return p

    def launch_signature(self):
        """Everything besides the plan's shape that changes the LAUNCH SEQUENCE of euler_step (part of the key of a captured graph):
        the tuning knobs on this object and the per-call state prepare() leaves in the plan."""
        p = self.plan
        return (p["ragged"], p["has_cond"], p["per_sample_t"], self.multi_stream, self.side_tile, tuple(sorted(self.side_tiles.items())),
                tuple(sorted(self.big_tiles.items())), tuple(sorted(self.split_tiles.items())), tuple(sorted(self.split_big_tiles.items())), self.main_tile, self.fold_norm, self.fold_gemm_all, self.fuse_skip, self.fuse_xattn, self.cross_on_main,
                self.interleave_capture, self.rope_cross, self.zero_masked_queries)

    # --------------------------------------------------------------------------- primitives
    def _sh(self, buf):
        """bf16 shadow of an fp32 stream buffer (None in fp32 mode)."""
        return self.plan["shadow"].get(buf.data_ptr())

    def _opnd(self, buf):
        """GEMM A operand for a residual stream: its bf16 shadow, or the fp32 buffer itself in fp32 mode."""
        s = self._sh(buf)
        return buf if s is None else s

    def _mm(self, segs, W, out, **kw):
        """GEMM on logical K segments [(operand buffer, lda, k)].  Plain modes: one v2a_gemm.  bf16x3: the operand buffers hold
        hi | lo planes (rows of 2k bf16) and W is [W_hi | W_lo]: one v2a_gemm with split operands (three MFMA products per fp32
        product inside the kernel); a requested bf16 shadow of the result is written as hi | lo planes by the same epilogue."""
        if kw.get("out_bf16") is not None and "ld_out_bf16" not in kw:
            kw["ld_out_bf16"] = kw["out_bf16"].stride(-2)        # a shadow may be half of a wider operand buffer
        if not self.split:
            return L.gemm(segs, W, out, compute=self.cdc, **kw)
        lo = self.plan["lo_off"]
        segs = [(buf, buf.stride(-2), k, lo.get(buf.data_ptr(), 0)) for buf, _, k in segs]
        if kw.get("out_bf16") is not None:
            kw["out_bf16_split"] = True
            kw["out_bf16_lo_offset"] = lo.get(kw["out_bf16"].data_ptr(), 0)
        return L.gemm(segs, W, out, compute=L.BF16, a_split=True, **kw)

    def _norm_plain(self, x, hn, rows, d, g):
        L.rmsnorm(x, hn, rows=rows, d=d, gamma=g, split=self.split)

    def _norm_ada(self, x, hn, rows, d, layer, slot):
        p = self.plan
        tab = p["norm_tab"][0, layer, slot]
        ss = p["norm_tab"].stride(0)
        if p["per_sample_t"]:
            L.rmsnorm(x, hn, rows=rows, d=d, gamma=tab, gamma_batch_stride=ss, rows_per_batch=p["N"], split=self.split)
        else:
            L.rmsnorm(x, hn, rows=rows, d=d, gamma=tab, step=p["step"], gamma_step_stride=ss, rows_per_batch=p["N"], split=self.split)

    def _fuse_skip(self):
        return self.fuse_skip and self.cd == torch.bfloat16 and "x_skip" in self.W.layers[-1]

    def _fold(self):
        c = self.cfg
        return (self.fold_norm and self.cd == torch.bfloat16
                and all(d % 32 == 0 and d <= 1280 for d in (c.dim, c.dim_text, c.dim_frames)))

    def _regime(self):
        """How a launch of this plan fills the chip, from the tile count of its narrow GEMMs (128x128 tiles over rows x dim) against
        the 256 CUs: 0 = a launch cannot fill the chip (one clip at the shipped dims: 104 tiles), 1 = about fills it (two clips: 200),
        2 = fills it several times over.  The per-(stream, op) tile table was measured in regime 0 at the shipped dims only and
        applies there only; other widths take the stream's default hint or the library's choice."""
        c = self.cfg
        tiles = -(-self.plan["rows"] // 128) * max(1, c.dim // 128)
        return 0 if tiles <= 140 else (1 if tiles <= 281 else 2)

    def _tuned_dims(self):
        c = self.cfg
        return ("bf16", 0, (c.dim, c.dim_text, c.dim_frames, c.ff_mult)) in TUNED_TILES

    def _fold_gemm(self):
        """... into a GEMM epilogue only up to two clips: the extra bf16 row pieces cost an out-projection launch 2 us of 24 at
        one clip (a norm launch: 7.7 us) but 20-26 us of 58 at 8 clips per GPU, more than the 16.5 us norm launch they replace
        (the conv fold wins at every size: 46 us against 49 + 16.5).  Re-measured in the bf16x3 mode in round 5 (`bench.py --fold-gemm-all`, 8 clips,
        alternating): 3960 / 3933, 3950 / 3924, 3949 / 3930 mel-frames/s -- the rule stands (-0.6 %)."""
        return self._fold() and (self._regime() < 2 or self.fold_gemm_all)

    def _nprod_ada(self, layer, slot, switch_row=0):
        """Producer side of a folded AdaptiveRMSNorm (audio stream): kwargs for the RESID / GATE_RESID GEMM or the conv that
        writes the rows to be normed.  switch_row: rows from there on take the NEXT slot's gamma (the null half of a CFG
        batch skips cross-attention, so its next norm is the feed-forward's)."""
        p, D = self.plan, self.cfg.dim
        ss = p["norm_tab"].stride(0)
        kw = dict(norm_gamma=p["norm_tab"][0, layer, slot], norm_ssq=p["ssq_a"], rows_per_batch=p["N"])
        if p["per_sample_t"]:
            kw["norm_batch_stride"] = ss
        else:
            kw.update(step=p["step"], norm_step_stride=ss)
        if switch_row:
            kw.update(norm_switch_row=switch_row, norm_switch_offset=D)
        return kw

    def _ncons(self, s, d):
        """Consumer side: the GEMM whose A operand is the un-normalised, gamma-scaled bf16 copy applies 1 / rms per row."""
        return dict(row_ssq=self.plan[f"ssq_{s}"], row_norm_dim=d) if self._fold() else {}

    def _gate_kw(self, layer, slot):
        p = self.plan
        tab = p["gate_tab"][0, layer, slot]
        ss = p["gate_tab"].stride(0)
        if p["per_sample_t"]:
            return dict(gate=tab, gate_batch_stride=ss, rows_per_batch=p["N"])
        return dict(gate=tab, step=p["step"], gate_step_stride=ss, rows_per_batch=p["N"])

    def _self_attn(self, A: _Attn, x, s, nseq, d, out_kw, in_kw={}):
        """x += epilogue(to_out(attend(rope(q), rope(k), v) * sigmoid(gate))), operand hn_s already normed."""
        p = self.plan
        N, rows = p["N"], nseq * p["N"]
        hn, qkv, ao = p[f"hn_{s}"], p[f"qkv_{s}"], p[f"ao_{s}"]
        if self._fuse_rope:       # RoPE of the q and k heads inside the QKV GEMM epilogue
            self._mm([(hn, d, d)], A.w_in, qkv, M=rows, N=A.n_pad, bias=A.b_in, ldo=A.n_pad,
                     rope_table=p["rope"], rope_cols=2 * A.inner, rope_pos_offset=0, rows_per_batch=N, **in_kw)
        else:
            self._mm([(hn, d, d)], A.w_in, qkv, M=rows, N=A.n_pad, bias=A.b_in, ldo=A.n_pad, **in_kw)
            L.rope(qkv, rows=rows, row_stride=A.n_pad, nheads=2 * A.heads, rows_per_batch=N, pos_offset=0,
                   table=p["rope"], layout=self.rope_layout)
        es = qkv.element_size()
        base = qkv.data_ptr()
        lens = p["seq_len"] if p["ragged"] else None
        aw = ao.stride(-2)                      # inner, or 2 * inner in bf16x3 mode: the kernel writes hi | lo planes itself
        # (the QKV epilogue writing q, k, v as hi | lo planes for the split attention kernel -- no conversion of the K / V tiles there -- is bit-equal
        # and 1.5-2 % SLOWER end to end: profiles/r05_attn_planes_ab.txt; not kept)
        L.attention(base, base + A.inner * es, base + 2 * A.inner * es, base + A.gate_col * es, ao.data_ptr(),
                    strides=(A.n_pad, A.n_pad, A.n_pad, A.n_pad, aw,
                             N * A.n_pad, N * A.n_pad, N * A.n_pad, N * A.n_pad, N * aw),
                    B=nseq, H=A.heads, Nq=N, Nk=N, kv_len=lens,
                    q_len=lens if self.zero_masked_queries else None,
                    scale=self.cfg.dim_head ** -0.5, softclamp=self.softclamp, dtype=self.adc, out_split=self.split)
        self._mm([(ao, A.inner, A.inner)], A.w_out, x, M=rows, N=d, resid=x, ldo=d, ldr=d, **out_kw)

    def _ff(self, Fw: _FF, x, s, nseq, d, out_kw, in_kw={}):
        p = self.plan
        rows = nseq * p["N"]
        hn, ffh = p[f"hn_{s}"], p[f"ffh_{s}"]
        # (A row split -- whole 256-row bands on the phase-interleaved kernel in one round, the 28-row tail as its own launch -- takes
        # the text stream's GEMM from 66.9 to 51 us alone and LOSES 2 % in the sampler, profiles/r03_rowsplit_probe.txt: not done.)
        self._mm([(hn, d, d)], Fw.w1, ffh, M=rows, N=2 * Fw.inner, epilogue=L.EPI_GEGLU, bias=Fw.b1, ldo=ffh.stride(-2),
                 **(dict(out_split=True) if self.split else {}), **in_kw)
        out_kw = dict(out_kw)
        shadow = out_kw.pop("out_bf16", self._sh(x))         # the audio stream redirects it into a wide operand buffer (forward)
        self._mm([(ffh, Fw.inner, Fw.inner)], Fw.w2, x, M=rows, N=d, bias=Fw.b2, resid=x, ldo=d, ldr=d, out_bf16=shadow, **out_kw)

    def _side_hint(self, stream="t", op="qkv"):
        """tile_hint of a text / frames GEMM (op: cross, qkv, out, ff1, ff2) while one launch cannot fill the chip anyway (up to
        two clips: M <= 3128 rows); with more rows every kernel fills all CUs and the library's stand-alone choice is faster
        (8 clips: text feed-forward 325 us on the 256x256 kernel against 556 us on forced 128x256 tiles).  `side_tiles` maps
        (stream, op) to a tile configuration of v2a_tuning.gemm_force_tile; missing entries take `side_tile`."""
        if self.split:                      # split-operand GEMMs have their own tile shapes (v2a_gemm): 0 = by shape
            return (self.split_big_tiles if self._regime() == 2 else self.split_tiles).get((stream, op), 0)
        if self.side_tile < 0:
            return 0
        r = self._regime()
        if r == 2:
            return self.big_tiles.get((stream, op), -1) + 1
        if r == 1 or not self._tuned_dims():        # two clips: the table below was tuned at one clip and costs 4 % here (6214 vs 6467)
            return self.side_tile + 1
        return self.side_tiles.get((stream, op), self.side_tile) + 1

    def _main_hint(self, op=None):
        if self.split:
            t = (self.split_big_tiles if self._regime() == 2 else self.split_tiles).get(("a", op), 0)
            return dict(tile_hint=t) if t else {}
        r = self._regime()
        if r == 2:
            t = self.big_tiles.get(("a", op), -1)
            return dict(tile_hint=t + 1) if t >= 0 else {}
        if r == 1 or not self._tuned_dims():
            return {}
        t = self.side_tiles.get(("a", op), self.main_tile if op in ("x_tfa", "skip", "out", "out2", "ff2") else -1)
        return dict(tile_hint=t + 1) if t >= 0 else {}

    def _side_block(self, ly, s, src, dst, nseq, d, parts=(0, 1, 2)):
        """text / frames stream block (x3:1081-1086, 1097-1101): conv, attention, feed-forward.  `parts` selects
        0 = conv + norm, 1 = attention, 2 = norm + feed-forward, so that the caller can interleave the CAPTURE order of the
        two side blocks and the audio block (a replayed graph hands kernels to the queues in capture order)."""
        p = self.plan
        N, rows = p["N"], nseq * p["N"]
        lens = p["seq_len"] if p["ragged"] else None
        cv = ly[f"{s}_conv"]
        hq, ho, h1, h2 = (dict(tile_hint=self._side_hint(s, op)) for op in ("qkv", "out", "ff1", "ff2"))
        fold, fold2 = self._fold(), self._fold_gemm()
        hn, ssq = p[f"hn_{s}"], p[f"ssq_{s}"]
        nc = self._ncons(s, d)
        cons = dict(**hq, **nc)
        cons2 = dict(**h1, **(nc if fold2 else {}))
        if 0 in parts:
            if fold:
                L.dwconv(src, dst, cv.wt, cv.b, B=nseq, N=N, d=d, ksize=cv.k, lens=lens,
                         norm=dict(out_bf16=hn, ld_out_bf16=hn.stride(-2), gamma=ly[f"{s}_g1"], ssq=ssq, split=self.split))
            else:
                L.dwconv(src, dst, cv.wt, cv.b, B=nseq, N=N, d=d, ksize=cv.k, lens=lens)
                self._norm_plain(dst, hn, rows, d, ly[f"{s}_g1"])
        if 1 in parts:
            prod = dict(out_bf16=hn, ld_out_bf16=hn.stride(-2), norm_gamma=ly[f"{s}_g2"], norm_ssq=ssq) if fold2 else {}
            self._self_attn(ly[f"{s}_attn"], dst, s, nseq, d, dict(epilogue=L.EPI_RESID, **ho, **prod), cons)
        if 2 in parts:
            if not fold2:
                self._norm_plain(dst, hn, rows, d, ly[f"{s}_g2"])
            self._ff(ly[f"{s}_ff"], dst, s, nseq, d, dict(epilogue=L.EPI_RESID, **h2), cons2)

    def _audio_cross_attention(self, i, ly, x, nctx, cons2, fold2, lens):
        """x[:nctx] += gate * to_out(attend(q(x), K_ctx, V_ctx)) of layer i (x3:1126-1133) on the current stream: one launch
        (v2a_qproj_xattn) + the out-projection, or q-projection, (RoPE,) attention, out-projection."""
        p, c, W = self.plan, self.cfg, self.W
        N, D = p["N"], c.dim
        inner = c.heads * c.dim_head
        nkv = 2 * c.depth * inner
        mh = self._main_hint
        r2 = nctx * N
        A2 = ly["a_attn2"]
        if not fold2:
            self._norm_ada(x, p["hn_a"], r2, D, i, 1)
        q2 = p["q2"]
        es = q2.element_size()
        kb = p["ctx_kv"].data_ptr() + i * inner * es
        vb = p["ctx_kv"].data_ptr() + (c.depth + i) * inner * es
        aw = p["ao_a"].stride(-2)
        # one launch while its 64-token x one-head workgroups stay under ~2.75 per CU (three clips at the shipped dims: 624; four clips
        # -- 832 -- are faster as two launches on larger tiles, profiles/r03_xattn_probe.txt)
        one_launch = (self.fuse_xattn and self.adc in (L.BF16, L.BF16_SPLIT) and nctx * -(-N // 64) * A2.heads <= 704 and p["nc"] <= 64 and D % 512 == 0
                      and A2.gate_col == A2.inner and (self._fuse_rope or not self.rope_cross))
        if one_launch:
            rk = dict(rope_table=p["rope"], rope_cols=A2.inner, rope_pos_offset=0) if self.rope_cross else {}
            nk = dict(row_ssq=cons2["row_ssq"], row_norm_dim=cons2["row_norm_dim"]) if cons2 else {}
            L.qproj_xattn(p["hn_a"], p["hn_a"].stride(-2) if self.split else D, D, A2.w_in, bias=A2.b_in, M=r2, N=A2.n_pad, rows_per_batch=N,
                          k=kb, v=vb, out=p["ao_a"].data_ptr(), split=self.split,
                          kv_strides=(nkv, nkv, p["nc"] * nkv, p["nc"] * nkv), out_strides=(aw, N * aw), B=nctx, H=A2.heads, Nk=p["nc"],
                          kv_len=p["ctx_len"], q_len=lens if self.zero_masked_queries else None, scale=c.dim_head ** -0.5,
                          softclamp=self.softclamp, **rk, **nk)
        elif self.rope_cross and self._fuse_rope:
            self._mm([(p["hn_a"], D, D)], A2.w_in, q2, M=r2, N=A2.n_pad, bias=A2.b_in, ldo=A2.n_pad,
                     rope_table=p["rope"], rope_cols=A2.inner, rope_pos_offset=0, rows_per_batch=N, **cons2, **mh("q2"))
        else:
            self._mm([(p["hn_a"], D, D)], A2.w_in, q2, M=r2, N=A2.n_pad, bias=A2.b_in, ldo=A2.n_pad, **cons2, **mh("q2"))
            if self.rope_cross:
                L.rope(q2, rows=r2, row_stride=A2.n_pad, nheads=A2.heads, rows_per_batch=N, pos_offset=0,
                       table=p["rope"], layout=self.rope_layout)
        if not one_launch:
            L.attention(q2.data_ptr(), kb, vb, q2.data_ptr() + A2.gate_col * es, p["ao_a"].data_ptr(),
                        strides=(A2.n_pad, nkv, nkv, A2.n_pad, aw,
                                 N * A2.n_pad, p["nc"] * nkv, p["nc"] * nkv, N * A2.n_pad, N * aw),
                        B=nctx, H=A2.heads, Nq=N, Nk=p["nc"], kv_len=p["ctx_len"],
                        q_len=lens if self.zero_masked_queries else None,
                        scale=c.dim_head ** -0.5, softclamp=self.softclamp, dtype=self.adc, out_split=self.split)
        prod2 = {}
        if fold2:
            n2 = {k: v for k, v in self._nprod_ada(i, 2).items() if k not in ("step", "rows_per_batch")}
            prod2 = dict(out_bf16=p["hn_a"], ld_out_bf16=p["hn_a"].stride(-2), **n2)
        self._mm([(p["ao_a"], inner, inner)], A2.w_out, x, M=r2, N=D, resid=x, ldo=D, ldr=D,
                 epilogue=L.EPI_GATE_RESID, **self._gate_kw(i, 1), **mh("out2"), **prod2)

    # ------------------------------------------------------------------------------ prepare
    def prepare(self, text, frames_roll, context, context_mask, t_points, *, lens=None,
                drop_text=None, drop_ctx=None, dt=None, step_cond=None, rope_len=None, rope_ctx_len=None):
        """Everything that is constant over the Euler loop.
        text (B,T,Dt) f32 CLIP features, frames_roll (B,T,51) f32, context (B,nc,ctx) f32,
        context_mask (B,nc) bool (prefix form), t_points (S,) f32 (host or device),
        lens (B,) valid latent frames per clip or None, drop_text / drop_ctx: per-sequence
        bool lists of length Bt / B (cfg_mode fills the null half itself).
        step_cond (B,T,C) f32 or None: the masked audio prompt of the infilling branch (x3:2228), already zeroed where dropped;
        it enters every evaluation as x += cond_proj_in(step_cond) on the conditional half and + bias on the null half
        (x3:2015-2035: a dropped cond is zero, the bias stays).
        rope_len / rope_ctx_len: sequence length (registers + frames) and context length the REFERENCE call would have had when
        the plan is padded to a bucket (defaults: the plan's N and nc): the cross-attention keys are rotated with the last
        rope_ctx_len positions of a rope_len-long table (A7), which padding must not move."""
        p, c, W, dev = self.plan, self.cfg, self.W, self.dev
        B, Bt, T, N, nc, S = p["B"], p["Bt"], p["T"], p["N"], p["nc"], p["S"]
        R, D, Dt, Df = c.num_registers, c.dim, c.dim_text, c.dim_frames
        assert text.shape == (B, T, Dt) and frames_roll.shape == (B, T, c.notes) and context.shape == (B, nc, c.ctx_dim)
        p["valid_T"].fill_(T if rope_len is None else min(T, int(rope_len) - R))
        # -- time conditioning + modulation tables for every grid point
        p["t_pts"].copy_(t_points.to(dev, torch.float32))
        if dt is not None:
            p["dt"].copy_(dt.to(dev, torch.float32))
        p["step"].zero_()
        L.time_cond(p["t_pts"], W.fourier_w, W.time_wt, W.time_b, p["tc"], S=S, d=D)
        nt = c.depth * 3 * D
        L.gemm([(p["tc"], D, D)], W.norm_gamma_w, p["norm_tab"], M=S, N=nt, compute=L.F32, bias=W.norm_gamma_b, ldo=nt)
        L.gemm([(p["tc"], D, D)], W.gate_w, p["gate_tab"], M=S, N=nt, compute=L.F32, epilogue=L.EPI_SIGMOID,
               bias=W.gate_b, ldo=nt)
        # -- sequence lengths (mask = lens_to_mask(duration) padded by R registers, x3:978-979, 2216)
        if lens is not None and not bool((torch.as_tensor(lens) == T).all()):
            sl = (torch.as_tensor(lens).to(torch.int32) + R).to(dev)
            p["seq_len"].copy_(sl.repeat(Bt // B))
            p["ragged"] = True
        else:
            p["seq_len"].fill_(N)
            p["ragged"] = False
        # -- step-invariant stream inputs: [registers ; features]
        if drop_text is None:
            drop_text = [False] * B + [True] * (Bt - B)
        if drop_ctx is None:
            drop_ctx = [False] * B
        L.fill_registers(p["t0"], W.text_regs, B=Bt, R=R, d=Dt, out_batch_stride=N * Dt)
        L.fill_registers(p["f0"], W.frames_regs, B=Bt, R=R, d=Df, out_batch_stride=N * Df)
        tx = text.to(dev, torch.float32)
        for s in range(Bt):                                   # device-memory plumbing, once per sample()
            if drop_text[s]:
                p["t0"][s, R:].zero_()                        # x3:2042-2044
            else:
                p["t0"][s, R:].copy_(tx[s % B])
        fr = frames_roll.to(dev, torch.float32).contiguous()
        L.linear_small(fr, W.pf_wt, W.pf_b, None, p["f0"], M=B * T, K=c.notes, T=T, out_batch_stride=N * Df,
                       row_off=R, d=Df, dup=(B if Bt > B else 0))                           # x3:2069
        # -- audio prompt (x3:2015-2035)
        p["has_cond"] = step_cond is not None
        if step_cond is not None:
            assert c.cond_proj_in, "step_cond needs cond_proj_in weights (E2TTS(if_cond_proj_in=True))"
            assert step_cond.shape == (B, T, c.num_channels)
            cb = torch.zeros(B, N, W.cond_k, device=dev)
            cb[:, R:, :c.num_channels] = step_cond.to(dev, torch.float32)
            cb = cb.reshape(B * N, W.cond_k)
            if self.split:
                L.split_bf16(cb, p["condbuf"], rows=B * N, d=W.cond_k)
            else:
                p["condbuf"].copy_(cb)
            p["padd"].copy_(W.pos_emb[:T] + W.cond_b)
        # -- cross-attention K/V of every layer from the (possibly dropped) context
        cx = context.to(dev, torch.float32).clone()
        for b in range(B):
            if drop_ctx[b]:
                cx[b] = 0                                     # x3:2058-2062
        if self.split:
            L.split_bf16(cx.reshape(B * nc, -1).contiguous(), p["ctx"], rows=B * nc, d=c.ctx_dim)
        else:
            p["ctx"].copy_(cx.reshape(B * nc, -1))
        cm = context_mask.to(torch.bool).cpu()
        cl = cm.sum(-1).to(torch.int32)
        assert bool((cm == (torch.arange(nc)[None] < cl[:, None])).all()), "context_mask must be a prefix mask"
        p["ctx_len"].copy_(cl)
        inner = c.heads * c.dim_head
        nkv = 2 * c.depth * inner
        self._mm([(p["ctx"], c.ctx_dim, c.ctx_dim)], W.ctx_kv_w, p["ctx_kv"], M=B * nc, N=nkv, ldo=nkv)
        if self.rope_cross:                                   # A7: keys take the LAST nc table rows
            off = (N if rope_len is None else int(rope_len)) - (nc if rope_ctx_len is None else int(rope_ctx_len))
            assert 0 <= off and off + nc <= p["rope"].shape[0], (off, nc, N)
            L.rope(p["ctx_kv"], rows=B * nc, row_stride=nkv, nheads=c.depth * c.heads, rows_per_batch=nc,
                   pos_offset=off, table=p["rope"], layout=self.rope_layout)
        # -- hoisted layer-0 text / frames blocks
        ly = W.layers[0]
        self._side_block(ly, "t", p["t0"], p["tL0"], Bt, Dt)
        self._side_block(ly, "f", p["f0"], p["fL0"], Bt, Df)

    # ------------------------------------------------------------------------------ forward
    def embed(self, y):
        """x0 = [registers ; proj_in(y) + abs_pos_emb]  (x3:2027, 957-960, 975-976) into xA."""
        p, c, W = self.plan, self.cfg, self.W
        B, Bt, T, N, D = p["B"], p["Bt"], p["T"], p["N"], c.dim
        L.linear_small(y, W.pin_wt, W.pin_b, p["padd"] if p["has_cond"] else W.pos_emb, p["xA"], M=B * T, K=c.num_channels, T=T,
                       out_batch_stride=N * D, row_off=c.num_registers, d=D, dup=(B if Bt > B else 0),
                       regs=W.regs, out_bf16=None if self.split else self._sh(p["xA"]))
        if self.split:              # operand planes of x0 (both halves); the conditional half is rewritten by the GEMM below
            L.split_bf16(p["xA"], self._sh(p["xA"]), rows=Bt * N, d=D)
        if p["has_cond"]:           # conditional half: x += cond_proj_in.weight @ step_cond (the bias sits in the position table)
            self._mm([(p["condbuf"], W.cond_k, W.cond_k)], W.cond_w, p["xA"], M=B * N, N=D, epilogue=L.EPI_RESID, resid=p["xA"],
                     ldo=D, ldr=D, out_bf16=self._sh(p["xA"]))

    def forward(self, n_ctx_seqs: int | None = None):
        """Transformer.forward over the plan's Bt sequences starting from xA; result in plan['pred'].

        Dependency structure of one layer i (x3:1081-1137): the text block T_i and frames block F_i need only
        their own stream; the cross-condition reads the PRE-update x, text, frames; the audio block A_i needs
        the cross-conditioned x.  Hence T_{i+1} and F_{i+1} run beside A_i.  With side streams this is
        expressed by events (captured as hipGraph edges); without them everything is issued in order on
        the current stream.  Hand-offs per layer: side blocks done (eT, eF -> main's cross-condition GEMM of the next
        layer), x of the layer ready (eA -> the side streams' own cross-condition GEMMs) and main's cross-condition GEMM
        done (eX -> the side blocks may overwrite the text / frames buffers it read).  All are forward edges: the main
        stream (the critical path) waits only for eT / eF, which the side streams reach with slack."""
        p, c, W = self.plan, self.cfg, self.W
        B, Bt, N, rows = p["B"], p["Bt"], p["N"], p["rows"]
        D, Dt, Df = c.dim, c.dim_text, c.dim_frames
        nctx = B if n_ctx_seqs is None else n_ctx_seqs
        lens = p["seq_len"] if p["ragged"] else None
        half = c.depth // 2
        fz = self._fuse_skip()
        inner = c.heads * c.dim_head
        nkv = 2 * c.depth * inner
        xc, xo = p["xA"], p["xB"]
        tc_, fc_ = p["tL0"], p["fL0"]
        tbuf, fbuf = [p["tA"], p["tB"]], [p["fA"], p["fB"]]

        main = torch.cuda.current_stream() if self.dev.type == "cuda" else None
        st, sf = (p.get("st"), p.get("sf")) if main is not None else (None, None)
        multi = st is not None

        class _On:                      # run a block of launches on a side stream (or inline)
            def __init__(self, s):
                self.s = s
            def __enter__(self):
                if multi:
                    self.ctx = torch.cuda.stream(self.s)
                    self.ctx.__enter__()
            def __exit__(self, *a):
                if multi:
                    self.ctx.__exit__(*a)

        def rec(stream):
            if not multi:
                return None
            e = torch.cuda.Event()
            e.record(stream)
            return e

        def wait(stream, *evs):
            if multi:
                for e in evs:
                    if e is not None:
                        stream.wait_event(e)

        eT = eF = None                  # layer 0's side blocks were hoisted into prepare()
        eA = rec(main) if not self.cross_on_main else None      # x of layer 0 (embed output) is ready; forks the side streams
        for i, ly in enumerate(W.layers):
            last = i == c.depth - 1
            # Cross condition (x3:686-702): three GEMMs, each reading the PRE-update x, text, frames of this layer.  x of this
            # layer (xc) is never overwritten during the layer -- the conv output goes to the other buffer -- so nothing has to
            # wait for the side streams to have read it.  (A kernel trace of the first version showed the main stream idle for
            # ~100 us per layer across two cross-stream hand-offs, each costing 10-50 us of queue-to-queue latency.)
            wait(main, eT, eF)
            xn = p["skips"][i] if i < half else xo
            ax, at_, af_ = self._opnd(xc), self._opnd(tc_), self._opnd(fc_)
            mh = self._main_hint
            fused = fz and i >= half
            if fused:
                # second half, bf16 mode: cross-condition + skip projection in one GEMM over [x | skip | text | frames]; x's
                # bf16 copy was written into the left half of the skip's wide buffer by the previous layer's FF2 epilogue
                wd = p["wide"][c.depth - 1 - i]
                ax = wd[..., :D]
                self._mm([(wd, 2 * D, 2 * D), (at_, Dt, Dt), (af_, Df, Df)], ly["x_skip"], p["xS"], M=rows, N=D, ldo=D, **mh("x_tfa"))
            else:
                self._mm([(ax, D, D), (at_, Dt, Dt), (af_, Df, Df)], ly["x_tfa"], xn, M=rows, N=D,
                         epilogue=L.EPI_RESID, resid=xc, ldo=D, ldr=D, out_bf16=self._sh(xn), **mh("x_tfa"))
            ax_ld = ax.stride(-2)
            if not last:
                nxt = W.layers[i + 1]
                on_side = multi and not self.cross_on_main
                hint_t = self._side_hint("t", "cross") if on_side else 0
                hint = self._side_hint("f", "cross") if on_side else 0
                def cross_t():
                    self._mm([(ax, ax_ld, D), (at_, Dt, Dt)], ly["x_at"], tbuf[0], M=rows, N=Dt,
                             epilogue=L.EPI_RESID, resid=tc_, ldo=Dt, ldr=Dt, tile_hint=hint_t)
                def cross_f():
                    self._mm([(ax, ax_ld, D), (af_, Df, Df)], ly["x_af"], fbuf[0], M=rows, N=Df,
                             epilogue=L.EPI_RESID, resid=fc_, ldo=Df, ldr=Df, tile_hint=hint)
                if self.cross_on_main or not multi:
                    cross_t()
                    cross_f()
                    eX = rec(main)
                else:
                    # eX: main has read this layer's text / frames buffers (x_tfa) -- the side blocks below overwrite them
                    eX = rec(main)
                    with _On(st):
                        wait(st, eA)
                        cross_t()
                    with _On(sf):
                        wait(sf, eA)
                        cross_f()
            # U-Net skip (x3:1108-1117).  First half: the cross-condition output buffer IS the saved skip.  Second half:
            # skip_proj(cat(x, skip)) -> spare buffer.  The conv output (and the whole audio block after it) goes to xo.
            if i < half:
                src = xn
            elif fused:
                src = p["xS"]
            else:
                src = p["xS"]
                sk = self._opnd(p["skips"][c.depth - 1 - i])          # bf16 mode: the right half of a wide operand buffer
                self._mm([(self._opnd(xn), D, D), (sk, sk.stride(-2), D)], ly["skip"], src, M=rows, N=D, ldo=D, **mh("skip"))
            dst = xo
            # audio stream (x3:1121-1137)
            cv = ly["a_conv"]
            fold, fold2 = self._fold(), self._fold_gemm()       # norm after the conv / norms after GEMM epilogues
            cons = self._ncons("a", D)
            cons2 = cons if fold2 else {}
            x = dst
            r2 = nctx * N
            prod1 = {}
            if fold:
                n0 = self._nprod_ada(i, 0)
                L.dwconv(src, dst, cv.wt, cv.b, B=Bt, N=N, d=D, ksize=cv.k, lens=lens,
                         norm=dict(out_bf16=p["hn_a"], ld_out_bf16=p["hn_a"].stride(-2), gamma=n0["norm_gamma"], ssq=p["ssq_a"], step=n0.get("step"),
                                   step_stride=n0.get("norm_step_stride", 0), batch_stride=n0.get("norm_batch_stride", 0), split=self.split))
            else:
                L.dwconv(src, dst, cv.wt, cv.b, B=Bt, N=N, d=D, ksize=cv.k, lens=lens)
                self._norm_ada(x, p["hn_a"], rows, D, i, 0)
            if fold2:
                # the norm after self-attention is cross-attention's (slot 1) for the rows that have a context, the
                # feed-forward's (slot 2) for the rest
                n1 = self._nprod_ada(i, 1, switch_row=r2 if r2 < rows else 0) if nctx > 0 else self._nprod_ada(i, 2)
                n1 = {k: v for k, v in n1.items() if k not in ("step", "rows_per_batch")}       # the gate already passes them
                prod1 = dict(out_bf16=p["hn_a"], ld_out_bf16=p["hn_a"].stride(-2), **n1)
            self._self_attn(ly["a_attn"], x, "a", Bt, D, dict(epilogue=L.EPI_GATE_RESID, **self._gate_kw(i, 0), **mh("out"), **prod1), dict(**cons, **mh("qkv")))
            if not last and self.interleave_capture:
                for part in (0, 1):
                    with _On(st):
                        if part == 0:
                            wait(st, eX)
                        self._side_block(nxt, "t", tbuf[0], tbuf[1], Bt, Dt, (part,))
                    with _On(sf):
                        if part == 0:
                            wait(sf, eX)
                        self._side_block(nxt, "f", fbuf[0], fbuf[1], Bt, Df, (part,))
            if nctx > 0:
                self._audio_cross_attention(i, ly, x, nctx, cons2, fold2, lens)
            # (holding the side streams' feed-forward launches back until here -- this out-projection's 208 small workgroups take 118-134 us
            # instead of 13.6 behind the 280 + 208 fat workgroups of those launches -- only moves the bubble: the text feed-forward then takes
            # 265 us and the audio queue waits for the text chain, -3.4 %: profiles/r05_hold_side_ff_ab.txt)
            if not fold2:
                self._norm_ada(x, p["hn_a"], rows, D, i, 2)
            # the bf16 copy of this layer's output: into the wide buffer of the next layer's skip when that layer is fused
            xsh = dict(out_bf16=p["wide"][c.depth - 2 - i][..., :D]) if (fz and half <= i + 1 < c.depth) else {}
            if last and fold2:      # the final RMSNorm folded like the others: gamma on the operand copy, 1 / rms per row in to_pred's epilogue
                xsh = dict(out_bf16=p["hn_a"], ld_out_bf16=p["hn_a"].stride(-2), norm_gamma=W.final_g, norm_ssq=p["ssq_a"])
            self._ff(ly["a_ff"], x, "a", Bt, D, dict(epilogue=L.EPI_GATE_RESID, **self._gate_kw(i, 2), **mh("ff2"), **xsh), dict(**cons2, **mh("ff1")))
            if not last:
                if not self.cross_on_main:
                    eA = rec(main)             # x of the next layer is ready
                # side streams: the NEXT layer's text / frames block beside this layer's audio block.  Issued AFTER the audio
                # block in program order: a replayed hipGraph hands its kernels to the queues in capture order (a few us
                # each), so whatever is captured first is submitted first -- with the side blocks in front, the audio
                # stream (the critical path) sat idle for ~75 us per layer until 18 side kernels had been handed over.
                rest = (2,) if self.interleave_capture else (0, 1, 2)
                with _On(st):
                    if not self.interleave_capture:
                        wait(st, eX)
                    self._side_block(nxt, "t", tbuf[0], tbuf[1], Bt, Dt, rest)
                    eT = rec(st)
                with _On(sf):
                    if not self.interleave_capture:
                        wait(sf, eX)
                    self._side_block(nxt, "f", fbuf[0], fbuf[1], Bt, Df, rest)
                    eF = rec(sf)
                tc_, fc_ = tbuf[1], fbuf[1]
            xc, xo = xo, xc
        # final norm over all rows (registers are dropped by the consumer) + to_pred (x3:1141-1143, 2083)
        if self._fold_gemm():
            self._mm([(p["hn_a"], D, D)], W.pred_w, p["pred"], M=rows, N=c.num_channels, bias=W.pred_b, ldo=c.num_channels, **self._ncons("a", D))
        else:
            L.rmsnorm(xc, p["hn_a"], rows=rows, d=D, gamma=W.final_g, split=self.split)
            self._mm([(p["hn_a"], D, D)], W.pred_w, p["pred"], M=rows, N=c.num_channels, bias=W.pred_b, ldo=c.num_channels)
        return p["pred"]

    def euler_step(self, y, cfg_strength: float, remove_parallel_component: bool = False, keep_parallel_frac: float = 0.0):
        """One CFG + Euler evaluation, in place on y (B,T,C): the unit that is hipGraph-captured."""
        p, c = self.plan, self.cfg
        self.embed(y)
        self.forward()
        kw = dict(B=p["B"], T=p["T"], C_=c.num_channels, pred_batch_stride=p["N"] * c.num_channels, row_off=c.num_registers)
        apg = None
        if remove_parallel_component:
            L.apg_reduce(p["pred"], p["apg"], valid_rows=p["valid_T"], **kw)
            apg = p["apg"]
        L.cfg_euler(y, p["pred"], cfg_strength=cfg_strength, dt=p["dt"], step=p["step"], apg=apg, keep=keep_parallel_frac, **kw)
        L.step_advance(p["step"])
