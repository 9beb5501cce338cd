"""Generates tests/golden/encodec_*.npz by running the third-party library the reference calls.   *** TEST INFRASTRUCTURE ***

The reference's vocoder is `EncodecModel.from_pretrained("facebook/encodec_24khz").decoder` (x3:421-423, 434-437).  The
checkpoint cannot be fetched offline, but the library is installed (transformers 5.15.0; the reference pins 4.46.0), so
`EncodecModel(EncodecConfig())` -- the same architecture, default config == encodec_24khz -- is instantiated here, loaded
with seeded weights (v2a_amd.synth.random_encodec_decoder_state_dict, numpy RandomState) and its decoder's outputs are
committed.  They pin both the CPU restatement (oracle/encodec_oracle.py) and the HIP path.

  encodec_small.npz   T = 24 latent frames: the full 7 680-sample waveform + LSTM / stage taps (sampled)
  encodec_full.npz    T = 750 (the BASELINE clip): statistics + 4 096 sampled samples of the 240 000-sample waveform

Usage:  python oracle/make_golden_encodec.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import v2a_amd  # noqa: E402,F401
from v2a_amd.synth import random_encodec_decoder_state_dict  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
PARAM_SEED, INPUT_SEED = 2468, 135


def latents(T, seed):
    """Sampler-like latents (b=1, 128, T): smooth O(1) trajectories, numpy RandomState."""
    rs = np.random.RandomState(seed)
    base = rs.standard_normal((128, T // 4 + 2)).astype(np.float32)
    x = np.repeat(base, 4, axis=1)[:, :T] + 0.3 * rs.standard_normal((128, T)).astype(np.float32)
    return torch.from_numpy(x[None])


def main():
    from transformers import EncodecConfig, EncodecModel
    torch.set_grad_enabled(False)
    model = EncodecModel(EncodecConfig()).eval()
    res = model.decoder.load_state_dict(random_encodec_decoder_state_dict(PARAM_SEED), strict=True)
    dec = model.decoder
    os.makedirs(OUT, exist_ok=True)
    for name, T in (("small", 24), ("full", 750)):
        emb = latents(T, INPUT_SEED + T)
        taps = {}
        hooks = []
        for idx, key in ((1, "lstm"), (4, "stage8"), (7, "stage5"), (10, "stage4"), (13, "stage2")):
            hooks.append(dec.layers[idx].register_forward_hook(lambda m, i, o, key=key: taps.__setitem__(key, o.detach().clone())))
        wav = dec(emb)
        for h in hooks:
            h.remove()
        w = wav[0, 0].numpy()
        rec = dict(stats=np.array([w.mean(dtype=np.float64), np.abs(w).mean(dtype=np.float64), w.max(), w.min()]), length=np.array(w.shape[0]))
        rs = np.random.RandomState(9)
        if name == "small":
            rec["wav"] = w
        else:
            ii = np.sort(rs.randint(0, w.shape[0], 4096))
            rec["wav_idx"], rec["wav_val"] = ii, w[ii]
        for k, v in taps.items():
            a = v[0].numpy()                                     # (C, L)
            ii = np.stack([rs.randint(0, a.shape[0], 256), rs.randint(0, a.shape[1], 256)], 1)
            rec[f"{k}_shape"], rec[f"{k}_idx"], rec[f"{k}_val"] = np.array(a.shape), ii, a[tuple(ii.T)]
            rec[f"{k}_absmean"] = np.array(np.abs(a).mean(dtype=np.float64))
        np.savez_compressed(os.path.join(OUT, f"encodec_{name}.npz"), **rec)
        print(name, "T", T, "->", w.shape, "abs mean %.4f max %.3f" % (np.abs(w).mean(), np.abs(w).max()),
              {k: float(np.abs(v.numpy()).mean()) for k, v in taps.items()})


if __name__ == "__main__":
    main()
