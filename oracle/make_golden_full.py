"""Full-shape golden vectors from the CPU oracle.                 *** TEST INFRASTRUCTURE ***

BASELINE.json configs[1] inputs (depth 12, dim 1024/1280/512, T = 750, nc = 16, 776.6 M parameters;
`init_params(cfg, 0)`, `synthetic_inputs(cfg, 1, 750, nc=16, seed=0)`), so that the GPU box never has to run
62 CPU forwards to check a 32-step sample:

  pred_cond_t037      one conditional forward at t = 0.37                      (750, 128)
  y_steps4            4-point sway grid, CFG 2.0 (configs[0] plumbing shape)   (750, 128)
  y_steps4_piano      the same with the V2P roll (`piano=True`, configs[3])    (750, 128)
  y_steps32           32-point sway grid, CFG 2.0 (configs[1])                 (750, 128)
  traj32_sub          every grid point of that run on every 8th latent frame   (32, 94, 128)
  y_steps64_piano     configs[3] at its real size: V2P roll on the 64-point grid of src/inference_v2p.py:183   (750, 128)
  y_steps4_ropecross, y_steps32_ropecross, traj32_sub_ropecross
                      the other reading of A7 (`OracleOptions(rope_cross=True)`: rotary applied in cross-attention);
                      the vectors above use the default reading (x-transformers 1.37.4 ignores it, oracle header)

Like every fixture of this path these pin the ORACLE (parity of the sampler stays "unpinned": the reference holds
no vectors and cannot be imported).  About 270 full-size forwards: ~30 min on 8 cores.

Usage:  python oracle/make_golden_full.py        (writes tests/golden/sample_full.npz)
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import e2_cfm_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SUB = 8


def main():
    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", max(1, (os.cpu_count() or 2) - 2))))
    cfg = O.DiTConfig()
    P = O.init_params(cfg, 0)
    t0 = time.time()
    out = {}
    y0, text, roll, ctx, cm = O.synthetic_inputs(cfg, 1, 750, nc=16, seed=0)
    kw = dict(cfg_strength=2.0, remove_parallel_component=False, sway_sampling=True)
    with torch.no_grad():
        out["pred_cond_t037"] = O.transformer_with_pred_head(P, cfg, y0, torch.tensor(0.37), None, text, roll, ctx, cm,
                                                             drop_text_cond=False, drop_text_prompt=False)[0].numpy()
        print("forward done %.0f s" % (time.time() - t0), flush=True)
        out["y_steps4"] = O.sample(P, cfg, y0, text, roll, ctx, cm, steps=4, **kw)[0].numpy()
        print("steps4 done %.0f s" % (time.time() - t0), flush=True)
        y0p, textp, rollp, ctxp, cmp_ = O.synthetic_inputs(cfg, 1, 750, nc=16, seed=0, piano=True)
        assert torch.equal(y0p, y0) and float(rollp.abs().sum()) > 0
        out["y_steps4_piano"] = O.sample(P, cfg, y0p, textp, rollp, ctxp, cmp_, steps=4, **kw)[0].numpy()
        print("steps4 piano done %.0f s" % (time.time() - t0), flush=True)
        y, traj = O.sample(P, cfg, y0, text, roll, ctx, cm, steps=32, return_trajectory=True, **kw)
        out["y_steps32"] = y[0].numpy()
        out["traj32_sub"] = torch.stack([p[0, ::SUB] for p in traj]).numpy()
        print("steps32 done %.0f s" % (time.time() - t0), flush=True)
        rc = O.OracleOptions(rope_cross=True)
        out["y_steps4_ropecross"] = O.sample(P, cfg, y0, text, roll, ctx, cm, steps=4, opts=rc, **kw)[0].numpy()
        y, traj = O.sample(P, cfg, y0, text, roll, ctx, cm, steps=32, return_trajectory=True, opts=rc, **kw)
        out["y_steps32_ropecross"] = y[0].numpy()
        out["traj32_sub_ropecross"] = torch.stack([p[0, ::SUB] for p in traj]).numpy()
        print("steps4 + steps32 with rope in cross-attention done %.0f s" % (time.time() - t0), flush=True)
        out["y_steps64_piano"] = O.sample(P, cfg, y0p, textp, rollp, ctxp, cmp_, steps=64, **kw)[0].numpy()
        print("steps64 piano done %.0f s" % (time.time() - t0), flush=True)
    out["meta"] = json.dumps(dict(param_seed=0, input_seed=0, T=750, nc=16, cfg_strength=2.0, traj_frame_stride=SUB, rope_cross_default=O.OracleOptions().rope_cross,
                                  torch=torch.__version__))
    np.savez_compressed(os.path.join(OUT, "sample_full.npz"), **out)
    print("wrote sample_full.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
