"""Drop-in `E2TTS` for the sampling path: same constructor keywords, `sample()`,
`transformer_with_pred_head()`, `cfg_transformer_with_pred_head()` and checkpoint key layout as
the reference class (x3:1275-1318, 1993-2113, 2127-2305; `x3` =
/root/reference/src/e2_tts_pytorch/e2_tts_crossatt3.py), running on the HIP kernels of
include/v2a_cfm.h.  Callers: predict.py:266, app.py:267, src/inference_v2a.py:183,
src/inference_v2p.py:183.

Scope: the Euler/CFG loop over the DiT.  The encoders that feed it run
once per clip and are out of scope, so their outputs are passed in through keyword-only
extensions (`text_embed`, `context`, `context_mask`, `frames_embed`, `y0`) or produced by
user-supplied callables (`video_encoder_fn`, `text_encoder_fn`, `frames_encoder_fn`).  The
reference draws the initial noise on the device inside sample() (x3:2248); `y0` makes it
injectable so results can be compared across devices.

There is no fallback path: without libv2a_cfm.so (gfx950) construction raises.
"""
from __future__ import annotations

from collections import namedtuple
from pathlib import Path
from typing import Callable

import torch

from . import _lib as L
from .dit import DiTConfig, DiTEngine, NOTES, process_streams

_IncompatibleKeys = namedtuple("_IncompatibleKeys", ["missing_keys", "unexpected_keys"])
_V2R_PREFIX = "video2roll_net."


def lens_to_mask(t: torch.Tensor, length: int | None = None) -> torch.Tensor:
    """x3:296-305."""
    if length is None:
        length = int(t.amax())
    seq = torch.arange(length, device=t.device)
    return seq[None, :] < t[:, None]


def sway_grid(steps: int, sway_sampling: bool = True) -> torch.Tensor:
    """x3:2250-2252, evaluated in fp32 on the host (same op order as the reference)."""
    t = torch.linspace(0, 1, steps)
    if sway_sampling:
        t = t + -1.0 * (torch.cos(torch.pi / 2 * t) - 1 + t)
    return t


def expected_state_dict_shapes(cfg: DiTConfig) -> dict[str, tuple]:
    """Key layout of the sampled path inside a reference checkpoint (nested ModuleList
    indices of x3:824-933; SURVEY section 5 'Checkpoint / resume')."""
    d, dt, df, k = cfg.dim, cfg.dim_text, cfg.dim_frames, cfg.kernel_size
    inner, ih, fh = cfg.heads * cfg.dim_head, cfg.heads, cfg.frames_heads
    s: dict[str, tuple] = {}

    def attn(p, dim, heads, ctx=None):
        inn = heads * cfg.dim_head
        s[f"{p}.to_q.weight"] = (inn, dim)
        s[f"{p}.to_k.weight"] = (inn, ctx or dim)
        s[f"{p}.to_v.weight"] = (inn, ctx or dim)
        s[f"{p}.to_v_head_gate.weight"] = (heads, dim)
        s[f"{p}.to_v_head_gate.bias"] = (heads,)
        s[f"{p}.to_out.weight"] = (dim, inn)

    def ff(p, dim, mult):
        s[f"{p}.ff.0.proj.weight"] = (2 * dim * mult, dim)
        s[f"{p}.ff.0.proj.bias"] = (2 * dim * mult,)
        s[f"{p}.ff.2.weight"] = (dim, dim * mult)
        s[f"{p}.ff.2.bias"] = (dim,)

    def conv(p, dim):
        s[f"{p}.dw_conv1d.0.weight"] = (dim, 1, k)
        s[f"{p}.dw_conv1d.0.bias"] = (dim,)

    T = "transformer"
    s[f"{T}.abs_pos_emb.weight"] = (cfg.max_seq_len, d)
    s[f"{T}.registers"] = (cfg.num_registers, d)
    s[f"{T}.text_registers"] = (cfg.num_registers, dt)
    s[f"{T}.frames_registers"] = (cfg.num_registers, df)
    s[f"{T}.time_cond_mlp.0.weights"] = (d // 2,)
    s[f"{T}.time_cond_mlp.1.weight"] = (d, d + 1)
    s[f"{T}.time_cond_mlp.1.bias"] = (d,)
    for i in range(cfg.depth):
        P = f"{T}.layers.{i}"
        if i >= cfg.depth // 2:
            s[f"{P}.0.0.weight"] = (d, 2 * d)
        conv(f"{P}.0.1", d)
        for j in (2, 5, 8):
            s[f"{P}.0.{j}.to_gamma.weight"] = (d, d)
        for j in (4, 7, 10):
            s[f"{P}.0.{j}.to_gamma.weight"] = (d, d)
            s[f"{P}.0.{j}.to_gamma.bias"] = (d,)
        attn(f"{P}.0.3", d, ih)
        attn(f"{P}.0.6", d, ih, cfg.ctx_dim)
        ff(f"{P}.0.9", d, cfg.ff_mult)
        conv(f"{P}.1.0", dt)
        s[f"{P}.1.1.g"] = (dt,)
        attn(f"{P}.1.2", dt, ih)
        s[f"{P}.1.3.g"] = (dt,)
        ff(f"{P}.1.4", dt, cfg.ff_mult)
        s[f"{P}.1.5.text_frames_to_audio.weight"] = (d, d + dt + df)
        if i != cfg.depth - 1:
            s[f"{P}.1.5.audio_to_text.weight"] = (dt, d + dt)
            s[f"{P}.1.5.audio_to_frames.weight"] = (df, d + df)
        conv(f"{P}.2.0", df)
        s[f"{P}.2.1.g"] = (df,)
        attn(f"{P}.2.2", df, fh)
        s[f"{P}.2.3.g"] = (df,)
        ff(f"{P}.2.4", df, 4)
    s[f"{T}.final_norm.g"] = (d,)
    s["proj_in.weight"] = (d, cfg.num_channels)
    s["proj_in.bias"] = (d,)
    s["to_pred.weight"] = (cfg.num_channels, d)
    s["to_pred.bias"] = (cfg.num_channels,)
    s["proj_frames.weight"] = (df, cfg.notes)
    s["proj_frames.bias"] = (df,)
    if cfg.cond_proj_in:                      # x3:1365 (bias optional: cond_proj_in_bias)
        s["cond_proj_in.weight"] = (d, cfg.num_channels)
        s["cond_proj_in.bias"] = (d,)
    return s


class E2TTS:
    def __init__(
        self,
        transformer: dict | None = None,
        duration_predictor=None,
        odeint_kwargs: dict = dict(method="euler"),
        audiocond_drop_prob=0.30,
        cond_drop_prob=0.20,
        prompt_drop_prob=0.10,
        num_channels=None,
        mel_spec_module=None,
        char_embed_kwargs: dict = dict(),
        mel_spec_kwargs: dict = dict(),
        frac_lengths_mask=(0.7, 1.0),
        audiocond_snr=None,
        concat_cond=False,
        interpolated_text=False,
        text_num_embeds=None,
        tokenizer="char_utf8",
        use_vocos=True,
        pretrained_vocos_path="charactr/vocos-mel-24khz",
        sampling_rate=None,
        frame_size: int = 320,
        velocity_consistency_weight=-1e-5,
        if_cond_proj_in=True,
        cond_proj_in_bias=True,
        if_embed_text=True,
        if_text_encoder2=True,
        if_clip_encoder=False,
        video_encoder="clip_vit",
        *,
        # ---- build-side extensions (keyword-only) ----
        compute_dtype: str = "bf16x3",       # "bf16x3" (default: split-bf16 GEMMs, inside 1e-3 of the fp32 reference path, config.yaml:7) | "bf16" (2x faster, ~5e-2 off) | "fp32" (exact-fp32 MFMA)
        device="cuda",
        rope_layout: str = "interleaved",
        rope_cross: bool = False,            # x-transformers 1.37.4 ignores rotary_pos_emb when a context is given; True = the other reading
        use_graph: bool = True,              # capture the Euler step in a hipGraph
        bucket_frames: int = 0,              # > 0: plans are padded to a multiple of this many latent frames (ragged masks hide the padding)
        bucket_ctx: int = 0,                 # > 0: ... and to a multiple of this many context tokens (context_mask hides the padding)
        video_encoder_fn: Callable | None = None,   # (video_paths, n) -> (b, n, dim_text)
        text_encoder_fn: Callable | None = None,    # (prompts) -> ((b, nc, ctx) float, (b, nc) bool)
        frames_encoder_fn: Callable | None = None,  # (frames, n) -> (b, n, NOTES)
    ):
        if not isinstance(transformer, dict):
            raise TypeError("transformer must be the keyword dict of the reference (predict.py:120-134)")
        if odeint_kwargs.get("method", "euler") != "euler":
            raise NotImplementedError("only the fixed-grid Euler solver of the shipped config is built (x3:1282-1287)")
        if concat_cond:
            raise NotImplementedError("concat_cond=True is not used by any shipped caller")
        tk = dict(transformer)
        for flag in ("if_text_modules", "if_cross_attn", "if_audio_conv", "if_text_conv"):
            if not tk.pop(flag, flag != "if_text_conv"):
                raise NotImplementedError(f"{flag}=False: only the shipped configuration (all True, predict.py:126-129) is built")
        tk.pop("cond_on_time", None)
        if num_channels is None:
            raise ValueError("num_channels is required (predict.py:152)")
        self.cfg = DiTConfig(num_channels=num_channels, cond_proj_in=bool(if_cond_proj_in), **tk)
        self.cond_proj_in_bias = bool(cond_proj_in_bias)
        self.audiocond_snr = audiocond_snr
        self.dim, self.dim_text = self.cfg.dim, self.cfg.dim_text
        self.num_channels = num_channels
        self.sampling_rate = sampling_rate
        self.frame_size = frame_size
        self.audiocond_drop_prob, self.cond_drop_prob, self.prompt_drop_prob = audiocond_drop_prob, cond_drop_prob, prompt_drop_prob
        self.duration_predictor = duration_predictor
        self.mel_spec = mel_spec_module
        self.vocos = None
        self.video_encoder = video_encoder
        self.training = False
        self._device = torch.device(device)
        if compute_dtype not in ("bf16", "fp32", "bf16x3"):
            raise ValueError(f"compute_dtype must be 'bf16', 'fp32' or 'bf16x3', got {compute_dtype!r}")
        self._compute = compute_dtype
        self._rope = (rope_layout, rope_cross)
        self._use_graph = use_graph
        # Shape buckets: captions and durations vary per clip (predict.py:210-237), and every new (frames, context) shape costs a
        # plan, an eager warm-up evaluation and a graph capture.  With buckets, sample() pads the latent frames and the context to
        # the bucket and lets the length masks (lens_to_mask x3:296-305, context_mask) hide the padding; results of the valid
        # frames do not change (tests/test_sampler_gpu.py::test_plan_cache_and_buckets).  0 = exact shapes.
        self.bucket_frames, self.bucket_ctx = int(bucket_frames), int(bucket_ctx)
        self.graph_captures = 0              # hipGraph captures so far (a plan / graph cache hit leaves it unchanged)
        self.video_encoder_fn, self.text_encoder_fn, self.frames_encoder_fn = video_encoder_fn, text_encoder_fn, frames_encoder_fn
        self._shapes = expected_state_dict_shapes(self.cfg)
        self._sd: dict[str, torch.Tensor] = {}
        self._engine: DiTEngine | None = None
        self._v2r_sd, self._v2r = None, None      # optional Video2Roll frame encoder (video2roll_net.*, x3:1523)
        L.lib()  # no library -> no sampler

    # ---- nn.Module-like surface used by the callers (predict.py:156-170) -------------------
    @property
    def device(self):
        return self._device

    def to(self, device):
        self._device = torch.device(device)
        self._engine = None
        self._v2r = None
        return self

    def eval(self):
        self.training = False
        return self

    def parameters(self):
        return iter(self._sd.values())

    def state_dict(self):
        sd = dict(self._sd)
        if self._v2r_sd is not None:
            sd.update({_V2R_PREFIX + k: v for k, v in self._v2r_sd.items()})
        return sd

    def load_state_dict(self, state_dict, strict: bool = True):
        """Accepts a reference checkpoint's `model_state_dict` (predict.py:168, strict=False there):
        keys of the sampled path are taken, encoder / vocoder / training-only keys are reported as
        unexpected."""
        missing, unexpected, new = [], [], {}
        for k, shp in self._shapes.items():
            if k == "cond_proj_in.bias" and not self.cond_proj_in_bias:
                new[k] = torch.zeros(shp)
                continue
            if k in state_dict:
                v = state_dict[k]
                if tuple(v.shape) != tuple(shp):
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(shp)}")
                new[k] = v.detach().to("cpu", torch.float32).contiguous()
            else:
                missing.append(k)
        # optional: the Video2Roll frame encoder (`video2roll_net.*`, x3:1523) -- taken when the checkpoint holds it whole
        v2r = {k[len(_V2R_PREFIX):]: v.detach().to("cpu") for k, v in state_dict.items() if k.startswith(_V2R_PREFIX)}
        if v2r:
            from .video2roll import expected_state_dict_shapes as v2r_shapes
            lacking = [k for k in v2r_shapes() if k not in v2r and not k.endswith("num_batches_tracked")]
            if lacking:
                raise RuntimeError(f"load_state_dict: {_V2R_PREFIX}* is incomplete, e.g. {lacking[:3]}")
            self._v2r_sd, self._v2r = v2r, None
        unexpected = [k for k in state_dict if k not in self._shapes and not k.startswith(_V2R_PREFIX)]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict(strict=True): missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        self._sd.update(new)
        self._engine = None          # plans and their graphs go with the engine
        return _IncompatibleKeys(missing, unexpected)

    def engine(self) -> DiTEngine:
        if self._engine is None:
            absent = [k for k in self._shapes if k not in self._sd]
            if any(k.startswith("cond_proj_in.") for k in absent):
                # a checkpoint of the shipped configuration has no audio-prompt projection (predict.py:144): the sampler runs
                # without it and the infilling branch raises, as the reference's `None(cond)` would (x3:2034)
                absent = [k for k in absent if not k.startswith("cond_proj_in.")]
                self.cfg.cond_proj_in = False
            if absent:
                raise RuntimeError(f"{len(absent)} parameters were never loaded (e.g. {absent[0]}): call load_state_dict first")
            self._engine = DiTEngine(self.cfg, {k: v for k, v in self._sd.items() if self.cfg.cond_proj_in or not k.startswith("cond_proj_in.")},
                                     self._device, compute=self._compute,
                                     rope_layout=self._rope[0], rope_cross=self._rope[1])
        return self._engine

    # ---- conditioning helpers ---------------------------------------------------------------
    def encode_frames(self, x, l: int):
        """x3:1525-1553: (b, 1, t, 100, 900) grey frames -> piano-roll probabilities (b, l, 51) on the HIP Video2Roll
        encoder (video2roll.py); needs the `video2roll_net.*` weights of the checkpoint (load_state_dict)."""
        if self._v2r_sd is None:
            raise RuntimeError("encode_frames: no `video2roll_net.*` weights were loaded (load_state_dict), pass "
                               "frames_embed= or frames_encoder_fn= instead")
        if self._v2r is None:
            from .video2roll import Video2RollEngine
            self._v2r = Video2RollEngine(self._v2r_sd, self._device, compute="bf16" if self._compute == "bf16" else "fp32")
        return self._v2r.encode_frames(x, l)

    def _get_context(self, prompt, context, context_mask, b):
        if context is None:
            if prompt is None:
                raise ValueError("pass `prompt` (with text_encoder_fn) or precomputed `context`/`context_mask`")
            if self.text_encoder_fn is None:
                raise NotImplementedError("FLAN-T5 encoding is outside the accelerated path (SURVEY 8): supply "
                                          "`context`/`context_mask` or construct with text_encoder_fn=")
            context, context_mask = self.text_encoder_fn(list(prompt))     # encode_text x3:1648-1657
        if context_mask is None:
            context_mask = torch.ones(context.shape[:2], dtype=torch.bool)
        assert context.shape[0] == b
        return context, context_mask

    # ---- reference API: one forward ----------------------------------------------------------
    @torch.no_grad()
    def transformer_with_pred_head(self, x, cond=None, times=None, mask=None, text=None, frames_embed=None,
                                   prompt=None, video_drop_prompt=None, audio_drop_prompt=None,
                                   drop_audio_cond: bool | None = None, drop_text_cond: bool | None = None,
                                   drop_text_prompt: bool | None = None, return_drop_conditions=False,
                                   *, context=None, context_mask=None):
        """x3:1993-2088.  x (b,n,C); times (b,) or 0-dim; mask (b,n) bool prefix mask or None;
        text (b,n,dim_text) float; frames_embed (b,n,NOTES).  Returns (b,n,C) on x.device."""
        if cond is not None and not self.cfg.cond_proj_in:
            raise NotImplementedError("cond != None needs cond_proj_in (E2TTS(if_cond_proj_in=True), x3:1365): with the shipped "
                                      "configuration (predict.py:144) the reference has no such layer either (x3:2034)")
        if self.training:
            raise NotImplementedError("training-time random condition dropping is out of scope")
        b, n, _ = x.shape
        dtc, dtp = bool(drop_text_cond), bool(drop_text_prompt)
        eng = self.engine()
        context, context_mask = self._get_context(prompt, context, context_mask, b)
        times = torch.as_tensor(times, dtype=torch.float32)
        if times.ndim == 0:
            times = times.repeat(b)
        eng.setup(b, n, context.shape[1], b, cfg_mode=False)
        p = eng.plan
        p["per_sample_t"] = True
        lens = None if mask is None else _mask_to_lens(mask)
        if frames_embed is None:
            frames_embed = torch.zeros(b, n, self.cfg.notes)
        drop_ctx = [dtp or bool(video_drop_prompt is not None and video_drop_prompt[i]) for i in range(b)]
        step_cond = None
        if cond is not None:                                  # x3:2015-2020: dropped prompts are zeroed (in place in the reference)
            step_cond = cond.detach().to(torch.float32).clone()
            for i in range(b):
                if bool(drop_audio_cond) or (audio_drop_prompt is not None and audio_drop_prompt[i]):
                    step_cond[i] = 0
        eng.prepare(text, frames_embed, context, context_mask, times, lens=lens,
                    drop_text=[dtc] * b, drop_ctx=drop_ctx, step_cond=step_cond)
        eng.embed(x.to(self._device, torch.float32).contiguous())
        pred = eng.forward(n_ctx_seqs=b)
        out = pred[:, self.cfg.num_registers:, :].to(x.device).clone()
        p["per_sample_t"] = False
        if return_drop_conditions:
            return out, [bool(drop_audio_cond)] * b, dtc, [dtp] * b
        return out

    @torch.no_grad()
    def cfg_transformer_with_pred_head(self, *args, cfg_strength: float = 1.0, remove_parallel_component: bool = True,
                                       keep_parallel_frac: float = 0.0, **kwargs):
        """x3:2090-2113 (two forwards; sample() uses the batched in-engine form instead)."""
        pred = self.transformer_with_pred_head(*args, drop_audio_cond=False, drop_text_cond=False, drop_text_prompt=False, **kwargs)
        if cfg_strength < 1e-5:
            return pred
        null = self.transformer_with_pred_head(*args, drop_audio_cond=True, drop_text_cond=True, drop_text_prompt=True, **kwargs)
        upd = pred - null
        if remove_parallel_component:
            shp = upd.shape
            xd, yd = upd.reshape(shp[0], -1).double(), pred.reshape(shp[0], -1).double()
            unit = torch.nn.functional.normalize(yd, dim=-1)
            par = (xd * unit).sum(-1, keepdim=True) * unit
            upd = ((xd - par) + par * keep_parallel_frac).reshape(shp).to(pred.dtype)
        return pred + upd * cfg_strength

    # ---- reference API: the sampler ------------------------------------------------------------
    @torch.no_grad()
    def sample(self, cond, *, text=None, lens=None, duration=None, steps=32, cfg_strength=1.0,
               remove_parallel_component=True, sway_sampling=True, max_duration=4096, vocoder=None,
               return_raw_output=None, save_to_filename=None, prompt=None, video_drop_prompt=None,
               audio_drop_prompt=None, video_paths=None, frames=None, midis=None,
               # build-side extensions
               y0=None, text_embed=None, context=None, context_mask=None, frames_embed=None, trajectory_out=None):
        """x3:2127-2305.  With lens == duration (every shipped call, predict.py:261-263) `cond` (b, n, C) only fixes shape and
        device.  With lens[0] != duration[0] it is the audio prompt of the infilling branch (x3:2196-2231, 2260-2261; needs
        if_cond_proj_in=True): zero-padded to the longest duration, masked to lens, added through cond_proj_in at every
        evaluation (dropped in the null pass), and returned unchanged in the first lens[b] frames.
        `trajectory_out`: optional list that receives a device copy of y at every grid point (the `trajectory` of
        x3:2255, of which the reference keeps only [-1]); test aid, adds a copy per step."""
        self.eval()
        if cond.ndim == 2:
            raise NotImplementedError("raw-wave `cond` needs mel_spec_module, which the shipped config does not set")
        batch, cond_seq_len = cond.shape[:2]
        out_device = cond.device
        cfgm = self.cfg
        # -- frames / piano roll (x3:2164-2176)
        if frames_embed is None:
            if frames is None or isinstance(frames, (int, float)):
                # no frames (V2A): all-zero roll (x3:2164-2165).  predict.py:270 also lets a float placeholder through, which the
                # reference's encode_frames would fail on (`x.shape`, x3:1527); here it means "no frames" instead of reaching the encoder
                frames_embed = torch.zeros(batch, cond_seq_len, cfgm.notes)
            elif self.frames_encoder_fn is not None:
                frames_embed = self.frames_encoder_fn(frames, cond_seq_len)
            elif self._v2r_sd is not None:
                frames_embed = self.encode_frames(frames, cond_seq_len)     # x3:2169
            else:
                raise NotImplementedError("`frames` needs the Video2Roll encoder: load a checkpoint holding `video2roll_net.*`, "
                                          "or pass frames_embed= / frames_encoder_fn=")
        if lens is None:
            lens = torch.full((batch,), cond_seq_len, dtype=torch.long)
        lens = torch.as_tensor(lens).cpu().long()
        # -- CLIP conditioning (x3:2183-2184)
        if text_embed is None:
            if video_paths is not None and self.video_encoder_fn is not None:
                text_embed = self.video_encoder_fn(video_paths, cond_seq_len)
            elif video_paths is not None:
                # cached CLIP features next to the videos, resampled to the latent rate (encode_video's cache branch,
                # x3:1796-1813); the CLIP encoder itself is outside the accelerated path
                from .features import encode_video_cached
                text_embed = encode_video_cached(video_paths, cond_seq_len, dim=cfgm.dim_text, video_encoder=self.video_encoder,
                                                 sampling_rate=self.sampling_rate or 24000, frame_size=self.frame_size)
            elif torch.is_tensor(text) and text.ndim == 3:
                text_embed = text
            else:
                raise NotImplementedError("pass video_paths (with cached .npz CLIP features), text_embed= (b, n, dim_text) "
                                          "or video_encoder_fn=")
        # -- duration (x3:2196-2216)
        if duration is None:
            duration = lens.clone()
        elif isinstance(duration, int):
            duration = torch.full((batch,), duration, dtype=torch.long)
        duration = torch.maximum(lens, torch.as_tensor(duration).cpu().long()).clamp(max=max_duration)
        assert duration.shape[0] == batch
        n = int(duration.amax())
        if frames_embed.shape[1] < n:
            pad = torch.zeros(batch, n - frames_embed.shape[1], cfgm.notes, dtype=frames_embed.dtype, device=frames_embed.device)
            frames_embed = torch.cat([frames_embed, pad], 1)
        frames_embed = frames_embed[:, :n]
        if text_embed.shape[1] != n:
            raise ValueError(f"text_embed has {text_embed.shape[1]} frames, the longest duration is {n}")
        # -- audio prompt (x3:2196-2231): the reference decides on clip 0 for the whole batch (x3:2224)
        step_cond = cond_mask = condp = None
        if int(lens[0]) != int(duration[0]):
            if not cfgm.cond_proj_in:
                raise NotImplementedError("lens != duration (audio-prompted infilling) needs cond_proj_in (E2TTS(if_cond_proj_in=True), "
                                          "x3:1365): with the shipped configuration the reference fails at x3:2034 as well")
            if self.audiocond_snr is not None:
                # (the reference cannot get through this branch either: add_noise indexes signal[mask] with the (b, n, 1) cond_mask of
                # x3:2213 on the (b, n, C) prompt, which torch refuses -- IndexError at x3:2124 -- so there is no behaviour to mirror)
                raise NotImplementedError("audiocond_snr: the reference adds fresh device noise to the prompt at every step (x3:2115-2125, 2228)")
            condp = torch.nn.functional.pad(cond.detach().to("cpu", torch.float32)[:, :n], (0, 0, 0, max(0, n - cond_seq_len)))   # x3:2212
            cond_mask = lens_to_mask(lens, n)[..., None]                                                            # x3:2196, 2213-2214
            step_cond = torch.where(cond_mask, condp, torch.zeros_like(condp))                                      # x3:2228
            for i in range(batch):                                                                                  # x3:2018-2020
                if audio_drop_prompt is not None and audio_drop_prompt[i]:
                    step_cond[i] = 0
        elif n != cond_seq_len:
            raise ValueError(f"cond has {cond_seq_len} frames but the longest duration is {n}")
        context, context_mask = self._get_context(prompt, context, context_mask, batch)
        drop_ctx = [bool(video_drop_prompt is not None and video_drop_prompt[i]) for i in range(batch)]
        # -- grid (x3:2250-2252) and noise (x3:2248)
        t = sway_grid(steps, sway_sampling)
        S = steps - 1
        if y0 is None:
            y0 = torch.randn(batch, n, cfgm.num_channels, device=self._device)
        # -- shape buckets (constructor): pad frames / context, the masks hide the padding; RoPE positions of the cross-attention
        #    keys stay those of the unpadded call (DiTEngine.prepare)
        nc = context.shape[1]
        n_plan = -(-n // self.bucket_frames) * self.bucket_frames if self.bucket_frames > 0 else n
        if n_plan > cfgm.max_seq_len:       # the bucket would run past the position table: exact shape for this call
            n_plan = n
        nc_plan = -(-nc // self.bucket_ctx) * self.bucket_ctx if self.bucket_ctx > 0 else nc
        pad_t = lambda x: x if x is None or x.shape[1] == n_plan else torch.nn.functional.pad(x, (0, 0, 0, n_plan - x.shape[1]))
        if nc_plan != nc:
            context = torch.nn.functional.pad(context, (0, 0, 0, nc_plan - nc))
            context_mask = torch.nn.functional.pad(context_mask.to(torch.bool), (0, nc_plan - nc))
        eng = self.engine()
        eng.setup(batch, n_plan, nc_plan, S, cfg_mode=True)
        p = eng.plan
        if "y" not in p:
            p["y"] = torch.empty(batch, n_plan, cfgm.num_channels, dtype=torch.float32, device=self._device)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        eng.prepare(pad_t(text_embed), pad_t(frames_embed), context, context_mask, t[:-1], lens=duration,
                    drop_ctx=drop_ctx, dt=t[1:] - t[:-1], step_cond=pad_t(step_cond),
                    rope_len=cfgm.num_registers + n, rope_ctx_len=nc)
        ev[1].record()
        self._run_steps(eng, pad_t(y0), S, float(cfg_strength), bool(remove_parallel_component), trajectory_out, n)
        ev[2].record()
        self._phase_events = ev          # device-side phase marks of this call (no host sync here): see phase_ms()
        out = p["y"][:, :n].to(out_device).clone()
        if step_cond is not None:
            out = torch.where(cond_mask.to(out_device), condp.to(out_device), out)        # x3:2260-2261: the prompt frames come back unchanged
        if return_raw_output:
            return out
        # -- waveform decode (x3:2270-2305), only if the caller attached a vocoder module
        mask = lens_to_mask(duration)
        if vocoder is not None:
            return vocoder(out.transpose(1, 2))
        if self.vocos is None:
            return out
        audio = [self.vocos.decode(mel[m].transpose(0, 1)[None]).squeeze(0) for mel, m in zip(out, mask)]
        if save_to_filename is not None:
            import torchaudio  # optional dependency of the caller's environment
            path = Path(save_to_filename)
            path.parents[0].mkdir(exist_ok=True, parents=True)
            for ind, one in enumerate(audio):
                name = path.name if len(audio) == 1 else f"{ind + 1}.{path.name}"
                torchaudio.save(str(path.parents[0] / name), one.detach().cpu()[None], sample_rate=self.sampling_rate)
        return audio

    def phase_ms(self):
        """(prepare_ms, euler_loop_ms) of the last sample() call, from events on the launch stream; synchronises."""
        ev = getattr(self, "_phase_events", None)
        if ev is None:
            return None
        ev[2].synchronize()
        return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])

    def _run_steps(self, eng: DiTEngine, y0, S, cfg_strength, apg, traj=None, n_valid=None):
        """steps-1 Euler evaluations (A12 of SURVEY 8c).  cfg_strength < 1e-5 (x3:2101) still
        runs the batched pass; the null half then has weight 0."""
        p = eng.plan
        y = p["y"]
        y.copy_(y0.to(self._device, torch.float32))
        p["step"].zero_()
        if traj is not None:
            traj.append(y[:, :n_valid].clone())
        if not self._use_graph:
            for _ in range(S):
                eng.euler_step(y, cfg_strength, apg)
                if traj is not None:
                    traj.append(y[:, :n_valid].clone())
            return
        # graphs live in the plan they were captured on (DiTEngine keeps the last few plans, least recently used first); the key
        # holds everything that changes the captured launch sequence: embed() and forward() issue different launches with an
        # audio prompt (has_cond), ragged lengths, or any tuning knob of the engine
        key = (cfg_strength, apg, eng.launch_signature())
        g = p["graphs"].get(key)
        if g is None:
            # warm-up on the real buffers (first-launch attribute calls must not happen under capture)
            keep = y.clone()
            eng.euler_step(y, cfg_strength, apg)
            torch.cuda.synchronize()
            y.copy_(keep)
            p["step"].zero_()
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: only this thread's calls are policed, so a communication library's watchdog thread
            # (RCCL under torch.distributed) polling its events meanwhile cannot invalidate the capture
            with torch.cuda.graph(g, stream=process_streams(self._device)[2], capture_error_mode="thread_local"):
                eng.euler_step(y, cfg_strength, apg)
            p["graphs"][key] = g
            self.graph_captures += 1
        for _ in range(S):
            g.replay()
            if traj is not None:
                traj.append(y[:, :n_valid].clone())


def _mask_to_lens(mask: torch.Tensor) -> torch.Tensor:
    mask = mask.to(torch.bool).cpu()
    lens = mask.sum(-1)
    if not bool((mask == lens_to_mask(lens, mask.shape[1])).all()):
        raise NotImplementedError("only prefix masks (lens_to_mask form, x3:296-305) are supported")
    return lens
