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


def random_video2roll_state_dict(seed: int = 0) -> dict[str, torch.Tensor]:
    """Seeded weights in the key layout of `Video2RollNet.resnet18(num_classes=51)` (no piano checkpoint is reachable
    offline: `./ckpts/piano5_4_2_8000.pt`, predict.py:68).  numpy's legacy RandomState stream, so the same seed gives the
    same tensors on any machine / torch build (the golden vectors of tests/golden/video2roll_*.npz depend on that).
    Conv weights follow the reference init N(0, sqrt(2 / (k*k*out))) (Video2RollNet.py:160-163); BatchNorm scale, shift
    and running statistics are non-trivial so the folded-BatchNorm path is exercised."""
    import numpy as np

    from .video2roll import expected_state_dict_shapes

    rs = np.random.RandomState(seed)
    sd = {}
    for k, shp in expected_state_dict_shapes().items():
        if k.endswith("num_batches_tracked"):
            v = np.array(1000, dtype=np.int64)
        elif k.endswith("running_var"):
            v = rs.uniform(0.5, 1.5, shp).astype(np.float32)
        elif k.endswith("running_mean"):
            v = (0.1 * rs.standard_normal(shp)).astype(np.float32)
        elif ("bn" in k or "downsample.1" in k) and k.endswith(".weight"):
            v = (1.0 + 0.1 * rs.standard_normal(shp)).astype(np.float32)
        elif k.endswith(".bias"):
            v = (0.1 * rs.standard_normal(shp)).astype(np.float32)
        elif len(shp) == 4:
            v = (rs.standard_normal(shp) * math.sqrt(2.0 / (shp[2] * shp[3] * shp[0]))).astype(np.float32)
        else:
            v = (rs.standard_normal(shp) / math.sqrt(shp[-1])).astype(np.float32)
        sd[k] = torch.from_numpy(v) if v.ndim else torch.tensor(int(v))
    return sd


def synthetic_piano_frames(b: int, t: int, H: int = 100, W: int = 900, seed: int = 0) -> torch.Tensor:
    """(b, 1, t, H, W) grey frames in [0, 1] (ToTensor range, x3:1878-1890): a static keyboard-like stripe pattern plus a
    few moving bright blobs and noise.  numpy RandomState, reproducible across machines."""
    import numpy as np

    rs = np.random.RandomState(seed)
    xs = np.arange(W, dtype=np.float32)[None, :]
    ys = np.arange(H, dtype=np.float32)[:, None]
    base = 0.55 + 0.35 * np.sign(np.sin(xs * (2 * np.pi / 17.3))) * (ys > 0.35 * H)
    out = np.empty((b, 1, t, H, W), dtype=np.float32)
    for bi in range(b):
        cx = rs.uniform(0, W, 6)
        vx = rs.uniform(-6, 6, 6)
        cy = rs.uniform(0.3 * H, H, 6)
        for ti in range(t):
            f = base.copy()
            for j in range(6):
                f += 0.4 * np.exp(-(((xs - (cx[j] + vx[j] * ti) % W) / 14.0) ** 2 + ((ys - cy[j]) / 9.0) ** 2))
            f += 0.03 * rs.standard_normal((H, W)).astype(np.float32)
            out[bi, 0, ti] = np.clip(f, 0.0, 1.0)
    return torch.from_numpy(out)


def random_encodec_decoder_state_dict(seed: int = 0) -> dict[str, torch.Tensor]:
    """Seeded weights in the key layout of `EncodecModel(EncodecConfig()).decoder.state_dict()` (weight_norm g / v pairs),
    the facebook/encodec_24khz architecture the reference loads by name (x3:421-423; unreachable offline).  numpy
    RandomState stream: same tensors on any machine.  Scales keep activations O(1) through the stack and the LSTM gates
    out of saturation."""
    import numpy as np

    from .encodec import expected_state_dict_shapes

    rs = np.random.RandomState(seed)
    sd = {}
    for k, shp in expected_state_dict_shapes().items():
        if k.endswith("original0"):
            v = rs.uniform(0.8, 1.6, shp)
        elif k.endswith("original1"):
            v = rs.standard_normal(shp) / math.sqrt(shp[1] * shp[2])
        elif "lstm.weight" in k:
            v = rs.uniform(-1.0, 1.0, shp) / math.sqrt(shp[1])
        elif "lstm.bias" in k:
            v = rs.uniform(-0.1, 0.1, shp)
        else:
            v = 0.05 * rs.standard_normal(shp)
        sd[k] = torch.from_numpy(v.astype(np.float32))
    return sd
