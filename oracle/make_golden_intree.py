"""Golden vectors produced by the REFERENCE'S OWN CODE for the in-tree blocks of the path.   *** TEST INFRASTRUCTURE ***

`src/e2_tts_pytorch/e2_tts_crossatt3.py` (x3) cannot be imported here (13 import-time dependencies are absent),
but it starts with `from __future__ import annotations`, so the definitions below -- which need nothing but torch and
einops, both installed -- can be taken out of the file's syntax tree AT GENERATION TIME and executed against the real
`torch` / `einops` names.  Nothing of the reference is copied into this repository: the script reads
/root/reference/... when it runs (in the build container only) and commits INPUTS and OUTPUTS.

  definitions executed as written          x3 lines    pins oracle function        pins HIP row (tests -m gpu)
  project + pack_one_with_inverse          147-173     apg_project                 a2  v2a_apg_reduce / v2a_cfg_euler
  AdaLNZero                                532-551     adaln_zero                  a12 EPI_SIGMOID table + EPI_GATE_RESID
  TextAudioCrossCondition                  664-702     cross_condition             a13 3 multi-segment v2a_gemm
  statements executed as written (sliced out of their enclosing method by line number)
  sway-warped grid of sample()             2250-2252   sway_grid                   a1  E2TTS.sample host grid
  CLIP frame resampling loop               1800-1808   --                          N3  features.resample_indices

What stays unpinned: everything whose arithmetic lives in x-transformers / torchdiffeq / einx (Attention, FeedForward,
RMSNorm, AdaptiveRMSNorm, RotaryEmbedding, odeint, and the einx-based DepthwiseConv / RandomFourierEmbed / lens_to_mask):
those packages are absent and no stand-ins are written for them.

Usage:  python oracle/make_golden_intree.py      (writes tests/golden/intree_blocks.npz)
"""
from __future__ import annotations

import ast
import json
import os
import types

import numpy as np
import torch
import torch.nn.functional as F
from einops import pack, rearrange, unpack
from torch import nn
from torch.nn import Module

X3 = "/root/reference/src/e2_tts_pytorch/e2_tts_crossatt3.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "intree_blocks.npz")
DEFS = ["exists", "default", "divisible_by", "pack_one_with_inverse", "project", "AdaLNZero", "TextAudioCrossCondition"]


def reference_namespace(tree):
    """Top-level definitions of x3 named in DEFS, compiled from the reference's own syntax tree."""
    nodes = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in DEFS]
    assert sorted(n.name for n in nodes) == sorted(DEFS), [n.name for n in nodes]
    # `from __future__ import annotations` (x3:11) keeps the jaxtyping-style annotations unevaluated, as in the reference
    mod = ast.Module(body=[ast.ImportFrom(module="__future__", names=[ast.alias(name="annotations")], level=0)] + nodes, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = dict(torch=torch, F=F, nn=nn, Module=Module, pack=pack, unpack=unpack, rearrange=rearrange)
    exec(compile(mod, X3, "exec"), ns)
    return ns, {n.name: (n.lineno, n.end_lineno) for n in nodes}


def statements(tree, first_line, last_line):
    """The statements of x3 that start within [first_line, last_line] at the shallowest nesting level that has any."""
    best = None
    for node in ast.walk(tree):
        body = getattr(node, "body", None)
        if not isinstance(body, list):
            continue
        for attr in ("body", "orelse"):
            seq = getattr(node, attr, None)
            if not isinstance(seq, list):
                continue
            hit = [s for s in seq if isinstance(s, ast.stmt) and first_line <= s.lineno <= last_line]
            if hit and (best is None or hit[0].col_offset < best[0].col_offset):
                best = hit
    assert best, (first_line, last_line)
    mod = ast.Module(body=best, type_ignores=[])
    ast.fix_missing_locations(mod)
    return compile(mod, X3, "exec"), (best[0].lineno, best[-1].end_lineno)


def main():
    src = open(X3).read()
    tree = ast.parse(src)
    assert "from __future__ import annotations" in src.split("import torch")[0]
    ns, where = reference_namespace(tree)
    g = torch.Generator().manual_seed(2024)
    r = lambda *s: torch.randn(*s, generator=g)
    out, meta = {}, dict(source=X3, lines=where, torch=torch.__version__)

    # ---- a2: project (APG decomposition of the CFG update), fp32 in -> fp64 inside -> fp32 out
    x, y = r(3, 40, 16), r(3, 40, 16)
    par, orth = ns["project"](x, y)
    out.update(project_x=x.numpy(), project_y=y.numpy(), project_parallel=par.numpy(), project_orthogonal=orth.numpy())

    # ---- a12: AdaLNZero with seeded (non-zero) to_gamma
    d = 128
    m = ns["AdaLNZero"](d)
    with torch.no_grad():
        m.to_gamma.weight.copy_(r(d, d) * 0.1)
        m.to_gamma.bias.copy_(r(d) * 0.5 - 2.0)
        xa, cond = r(2, 44, d), r(2, d)
        out.update(adaln_w=m.to_gamma.weight.numpy().copy(), adaln_b=m.to_gamma.bias.numpy().copy(), adaln_x=xa.numpy(), adaln_cond=cond.numpy(),
                   adaln_out=m(xa, condition=cond).numpy())

    # ---- a13: TextAudioCrossCondition, both settings of cond_audio_to_text (x3:888-890: False on the last layer)
    da, dt, df = 128, 192, 64
    a, t, f = r(2, 44, da), r(2, 44, dt), r(2, 44, df)
    out.update(cc_audio=a.numpy(), cc_text=t.numpy(), cc_frames=f.numpy())
    for flag in (True, False):
        m = ns["TextAudioCrossCondition"](da, dt, df, cond_audio_to_text=flag)
        with torch.no_grad():
            for name, p in m.named_parameters():
                p.copy_(r(*p.shape) * 0.05)
                out[f"cc_{int(flag)}_{name.replace('.', '_')}"] = p.numpy().copy()
            oa, ot, of = m(a, t, f)
        out.update({f"cc_{int(flag)}_out_audio": oa.numpy(), f"cc_{int(flag)}_out_text": ot.numpy(), f"cc_{int(flag)}_out_frames": of.numpy()})

    # ---- a1: the sway-warped time grid, statements of sample() as written
    code, span = statements(tree, 2250, 2252)
    meta["sway_lines"] = span
    for steps in (2, 4, 25, 32, 64):
        for sway in (True, False):
            env = dict(torch=torch, self=types.SimpleNamespace(device="cpu"), steps=steps, sway_sampling=sway)
            exec(code, env)
            out[f"sway_{steps}_{int(sway)}"] = env["t"].numpy()

    # ---- N3: nearest-frame resampling of cached CLIP embeddings to the latent rate, loop of encode_video as written
    code, span = statements(tree, 1799, 1808)
    meta["resample_lines"] = span
    cases = [(240, 10.0, 750, 0, None), (251, 10.04, 750, 0, None), (30, 1.3, 750, 0, None), (300, 10.0, 400, 0, None), (2, 0.5, 750, 0, None),
             (240, 10.0, 750, 24000, 120000), (97, 3.97, 300, 0, 48000)]
    meta["resample_cases"] = cases
    for i, (nf, dur, l, start, mx) in enumerate(cases):
        env = dict(torch=torch, self=types.SimpleNamespace(sampling_rate=24000, frame_size=320), duration=dur, l=l, start_sample=start,
                   max_sample=mx, image_embeddings=torch.arange(nf, dtype=torch.float32)[:, None], min=min, round=round, range=range, len=len, int=int)
        exec(code, env)
        out[f"resample_{i}"] = env["interpolated"][:, 0].to(torch.int64).numpy()      # x3:1808 already concatenated the rows

    out["meta"] = json.dumps(meta)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays; definitions at", where, "sway", meta["sway_lines"], "resample", meta["resample_lines"])


if __name__ == "__main__":
    main()
