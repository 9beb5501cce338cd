"""Synthetic weights and conditioning of the BASELINE shape (no checkpoints or datasets are
reachable offline: every *.pt / *.mp4 in the reference tree is a Git-LFS stub).
Used by bench.py and __graft_entry__.smoke(); distributions follow SURVEY section 8(d)."""
from __future__ import annotations

import math

import torch

from .dit import DiTConfig
from .e2tts import expected_state_dict_shapes


def random_state_dict(cfg: DiTConfig, seed: int = 0, device="cpu") -> dict[str, torch.Tensor]:
    """Random-init weights in the reference's checkpoint key layout.  Matrix weights ~ N(0, 1/fan_in)
    (activations stay O(1) through 12 layers); the parameters the reference zero-initialises
    (to_gamma, AdaLNZero, TextAudioCrossCondition) are non-zero so every kernel does real work."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for k, shp in expected_state_dict_shapes(cfg).items():
        r = lambda: torch.randn(shp, generator=g, device=device, dtype=torch.float32)
        if k.endswith(".g"):
            v = 1.0 + 0.1 * r()
        elif k.endswith("time_cond_mlp.0.weights"):
            v = r()
        elif k.endswith("registers") or k.endswith("abs_pos_emb.weight"):
            v = 0.5 * r()
        elif k.endswith("to_v_head_gate.bias"):
            v = 1.0 + r()
        elif k.endswith(".bias"):
            v = 0.1 * r()
        elif k.endswith("dw_conv1d.0.weight"):
            v = r() / math.sqrt(shp[-1])
        elif "to_gamma.weight" in k:
            v = r() * (0.5 / math.sqrt(shp[-1]))
        elif any(t in k for t in ("text_frames_to_audio", "audio_to_text", "audio_to_frames")):
            v = r() * (0.3 / math.sqrt(shp[-1]))
        else:
            v = r() / math.sqrt(shp[-1])
        sd[k] = v
    return sd


def synthetic_conditioning(cfg: DiTConfig, b: int, n: int = 750, nc: int = 16, seed: int = 0, piano: bool = False, device="cpu"):
    """(y0, clip_embed, roll, context, context_mask): CLIP features piecewise constant over ~31-frame
    runs (24 fps frames nearest-neighbour resampled to 75 Hz, x3:1803-1805), T5 context of nc tokens,
    zero piano roll for V2A (x3:2164-2165) or a ~5 % dense roll for V2P."""
    g = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, device=device)
    y0 = r(b, n, cfg.num_channels)
    run = 31
    nseg = (n + run - 1) // run
    text = (0.5 * r(b, nseg, cfg.dim_text)).repeat_interleave(run, dim=1)[:, :n].contiguous()
    context = 0.2 * r(b, nc, cfg.ctx_dim)
    context_mask = torch.ones(b, nc, dtype=torch.bool)
    if piano:
        u = torch.rand(b, n, cfg.notes, generator=g, device=device)
        roll = torch.where(u > 0.95, u, torch.zeros((), device=device))
    else:
        roll = torch.zeros(b, n, cfg.notes, device=device)
    return y0, text, roll, context, context_mask
