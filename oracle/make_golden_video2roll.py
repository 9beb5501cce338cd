"""Generates tests/golden/video2roll_*.npz by running the REFERENCE module.          *** TEST INFRASTRUCTURE ***

`src/audeo/Video2RollNet.py` of the reference imports only torch + math, so it is loaded here straight from
/root/reference (read-only, never copied) with seeded weights and inputs; its outputs pin BOTH the CPU restatement
(oracle/video2roll_oracle.py) and the HIP path.  `E2TTS.encode_frames` itself lives in a file that cannot be imported
(SURVEY 8c), so the window / repeat / pad logic around the network is pinned by re-running its few tensor lines
(x3:1531-1553) literally on the reference network's outputs below.

Fixtures hold outputs only (inputs and weights are regenerated from seeds by v2a_amd.synth, numpy RandomState):
  video2roll_forward.npz  logits of 3 windows (3, 51) + per-tap statistics and sampled entries of x1..x4, x5, x2_/x3_/x4_
  video2roll_encode.npz   encode_frames output for b=2, t=4, l=10 and l=14 (crop and zero-pad branches)

Usage:  python oracle/make_golden_video2roll.py  [--reference /root/reference]
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import v2a_amd  # noqa: E402,F401
from v2a_amd.synth import random_video2roll_state_dict, synthetic_piano_frames  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
PARAM_SEED, INPUT_SEED = 4321, 77
NOTES = 51


def load_reference_net(ref_root):
    path = os.path.join(ref_root, "src", "audeo", "Video2RollNet.py")
    spec = importlib.util.spec_from_file_location("ref_Video2RollNet", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    net = mod.resnet18(num_classes=NOTES)
    net.load_state_dict(random_video2roll_state_dict(PARAM_SEED), strict=True)      # every key present, none extra
    net.eval()
    return net


def sample_idx(shape, k=64, seed=5):
    rs = np.random.RandomState(seed)
    return np.stack([rs.randint(0, s, k) for s in shape], 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    net = load_reference_net(args.reference)
    os.makedirs(OUT, exist_ok=True)

    # ---- forward: 3 explicit windows, with taps captured by forward hooks on the reference module ----
    frames = synthetic_piano_frames(1, 7, seed=INPUT_SEED)          # (1, 1, 7, 100, 900)
    idx = (torch.arange(7)[:, None] + torch.arange(-2, 3)[None, :]).clamp(0, 6)
    windows = frames[0, 0][idx][[0, 3, 6]]                           # first (clamped), middle, last (clamped) window
    taps = {}
    hooks = []
    for name in ("layer1", "layer2", "layer3", "layer4", "toplayer_relu", "FTB2_2", "FTB3", "FTB4"):
        hooks.append(getattr(net, name).register_forward_hook(lambda m, i, o, name=name: taps.__setitem__(name, o.detach().clone())))
    logits = net(windows)
    for h in hooks:
        h.remove()
    rename = dict(layer1="x1", layer2="x2", layer3="x3", layer4="x4", toplayer_relu="x5", FTB2_2="x2_", FTB3="x3_", FTB4="x4_")
    rec = dict(logits=logits.numpy(), window_ids=np.array([0, 3, 6]))
    for k, v in taps.items():
        a = v.numpy()
        ii = sample_idx(a.shape)
        rec[f"{rename[k]}_shape"] = np.array(a.shape)
        rec[f"{rename[k]}_stats"] = np.array([a.mean(dtype=np.float64), np.abs(a).mean(dtype=np.float64), a.max(), a.min()])
        rec[f"{rename[k]}_idx"] = ii
        rec[f"{rename[k]}_val"] = a[tuple(ii.T)]
    np.savez_compressed(os.path.join(OUT, "video2roll_forward.npz"), **rec)
    print("forward: logits", logits.shape, "abs mean", float(logits.abs().mean()), "range", float(logits.min()), float(logits.max()))

    # ---- encode_frames: the tensor lines of x3:1531-1553 re-run on the reference network ----
    x = synthetic_piano_frames(2, 4, seed=INPUT_SEED + 1)            # (b=2, 1, t=4, 100, 900)
    enc = {}
    for l in (10, 14):
        b, c, t, w, h = x.shape
        x_all = []
        for i in range(t):
            fr = []
            for j in [-2, -1, 0, 1, 2]:
                f = min(max(i + j, 0), t - 1)
                fr.append(x[:, :, f:f + 1, :, :])
            x_all.append(torch.cat(fr, dim=2))
        xx = torch.cat(x_all, dim=1).reshape(b * t, 5, w, h)
        y = torch.sigmoid(net(xx))
        y = y.reshape(b, t, 1, NOTES).repeat(1, 1, 3, 1).reshape(b, t * 3, NOTES)
        d = y.shape[1]
        if d > l:
            y = y[:, :l, :]
        elif d < l:
            y = torch.cat((y, torch.zeros(b, l - d, NOTES)), 1)
        enc[f"roll_l{l}"] = y.numpy()
    np.savez_compressed(os.path.join(OUT, "video2roll_encode.npz"), **enc)
    print("encode:", {k: v.shape for k, v in enc.items()}, "mean prob", float(enc["roll_l10"].mean()))


if __name__ == "__main__":
    main()
