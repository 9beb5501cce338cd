"""Golden vectors of the audio-prompted (infilling) branch of sample() from the CPU oracle.   *** TEST INFRASTRUCTURE ***

x3:2015-2035 (cond -> cond_proj_in, dropped in the null pass), x3:2196-2231 (cond padded to the longest duration, masked to
the prompt length) and x3:2260-2261 (the prompt frames are returned unchanged); `x3` =
/root/reference/src/e2_tts_pytorch/e2_tts_crossatt3.py.  No shipped caller reaches this branch (lens == duration at
predict.py:261-263, and the shipped config builds no cond_proj_in), so these vectors pin the oracle's restatement only.

Small config of oracle/make_golden.py plus cond_proj_in; clips of 40 and 33 frames with prompts of 12 and 20 frames given as a
24-frame `cond` (shorter than the longest duration: exercises the padding of x3:2212).

Usage:  python oracle/make_golden_prompt.py      (writes tests/golden/sample_small_prompt.npz)
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import e2_cfm_oracle as O  # noqa: E402
from oracle.make_golden import SMALL, PARAM_SEED, INPUT_SEED, B, T, NC, param_fingerprint  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sample_small_prompt.npz")


def main():
    cfg = O.DiTConfig(**SMALL, cond_proj_in=True)
    P = O.init_params(cfg, PARAM_SEED)
    base = O.init_params(O.DiTConfig(**SMALL), PARAM_SEED)
    assert all(torch.equal(P[k], v) for k, v in base.items())          # the switch only appends parameters
    y0, text, roll, ctx, cm = O.synthetic_inputs(cfg, B, T, nc=NC, seed=INPUT_SEED, piano=True)
    g = torch.Generator().manual_seed(77)
    cond = torch.randn(B, 24, cfg.num_channels, generator=g)
    lens, duration = torch.tensor([12, 20]), torch.tensor([T, 33])
    kw = dict(steps=4, cfg_strength=2.0, sway_sampling=True, remove_parallel_component=False, duration=duration, cond=cond, lens=lens)
    out = dict(y0=y0.numpy(), text=text.numpy(), roll=roll.numpy(), ctx=ctx.numpy(), ctx_mask=cm.numpy(), cond=cond.numpy(),
               lens=lens.numpy(), duration=duration.numpy(), param_fingerprint=param_fingerprint(P),
               cond_proj_in_weight=P["cond_proj_in.weight"].numpy(), cond_proj_in_bias=P["cond_proj_in.bias"].numpy(),
               meta=json.dumps(dict(cfg={**SMALL, "cond_proj_in": True}, param_seed=PARAM_SEED, input_seed=INPUT_SEED)))
    with torch.no_grad():
        out["y_prompt"] = O.sample(P, cfg, y0, text, roll, ctx, cm, **kw).numpy()
        out["y_prompt_audio_drop"] = O.sample(P, cfg, y0, text, roll, ctx, cm, audio_drop_prompt=[False, True], **kw).numpy()
        # one conditional and one null forward with the prompt (transformer_with_pred_head, x3:1993-2088)
        mask = O.lens_to_mask(duration, T)
        condp = torch.nn.functional.pad(cond, (0, 0, 0, T - cond.shape[1]))
        sc = torch.where(O.lens_to_mask(lens, T)[..., None], condp, torch.zeros_like(condp))
        for name, drop in (("pred_cond", False), ("pred_null", True)):
            out[name] = O.transformer_with_pred_head(P, cfg, y0, torch.tensor(0.37), mask, text, roll, ctx, cm, drop_text_cond=drop,
                                                     drop_text_prompt=drop, cond=sc, drop_audio_cond=drop).numpy()
    out["step_cond"] = sc.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith(("y_", "pred"))})


if __name__ == "__main__":
    main()
