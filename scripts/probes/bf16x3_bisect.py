"""First frames-QKV GEMM of a bf16x3 sample under two tile hints: are the inputs equal, are the outputs?  (Debug aid: this is
how the fused-RoPE epilogue was found to round differently in the 8-phase and ring instantiations -- the rotation is now
written with explicit fmaf and `test_gemm_tile_hint_three_segments_bitwise` pins it.)"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import e2_cfm_oracle as O
from conftest import make_model
from v2a_amd import _lib as L

cfg = O.DiTConfig()
P = O.init_params(cfg, 0)
y0, text, roll, ctx, cm = O.synthetic_inputs(cfg, 1, 750, nc=16, seed=0, piano=True)
rec = {}
orig = L.gemm


def spy(a_segs, w, out, **kw):
    orig(a_segs, w, out, **kw)
    if kw["N"] == 1552 and "replayed" not in rec:
        rec["replayed"] = True
        torch.cuda.synchronize()
        res = {}
        for h in (7, 1, 4, 2):
            o2 = torch.full_like(out, float("nan"))
            k2 = dict(kw)
            k2["tile_hint"] = h
            orig(a_segs, w, o2, **k2)
            torch.cuda.synchronize()
            res[h] = o2
        print("in-situ hint", kw.get("tile_hint"), "| replays: 7==1", torch.equal(res[7], res[1]), "7==4", torch.equal(res[7], res[4]), "1==4", torch.equal(res[1], res[4]),
              "2==4", torch.equal(res[2], res[4]), "| in-situ == replay7", torch.equal(out, res[7]), "== replay1", torch.equal(out, res[1]),
              "| kw", {k: (v if not torch.is_tensor(v) else tuple(v.shape)) for k, v in kw.items()},
              "| segs", [(tuple(t.shape), t.stride(), l, k, t.data_ptr() % 4096) for t, l, k in a_segs], "w", tuple(w.shape), w.stride(), flush=True)
    tag = rec["tag"]
    key = (kw["M"], kw["N"], sum(k for _, _, k in a_segs), kw.get("epilogue", 0))
    lst = rec.setdefault(tag, [])
    if len(lst) < 40:
        torch.cuda.synchronize()
        lst.append((key, kw.get("tile_hint", 0), [t.clone() for t, _, _ in a_segs], out.clone()))


L.gemm = spy
for name, v in (("hint7", 6), ("fat", 0)):
    rec["tag"] = name
    m = make_model(cfg, P, "bf16x3", use_graph=False)
    e = m.engine()
    e.multi_stream = False
    e.side_tiles = {("f", "qkv"): v}
    m.sample(torch.zeros(1, 750, 128), y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll,
             return_raw_output=True, steps=2, cfg_strength=2.0, remove_parallel_component=False)
for n, (ra, rb) in enumerate(zip(rec["hint7"], rec["fat"])):
    same_in = all(torch.equal(x, y) for x, y in zip(ra[2], rb[2]))
    same_out = torch.equal(ra[3], rb[3])
    flag = "" if (same_in and same_out) else "   <-----"
    print(f"call {n:2d} {ra[0]} hints {ra[1]}/{rb[1]}: inputs equal {same_in}, outputs equal {same_out}, max out diff {float((ra[3].float() - rb[3].float()).abs().max()):.3e}{flag}")
    if not same_out and same_in:
        d = (ra[3].float() - rb[3].float()).abs()
        idx = torch.nonzero(d > 0)
        print("   differing elements:", idx.shape[0], "rows", int(idx[:, 0].min()), "-", int(idx[:, 0].max()), "cols", int(idx[:, 1].min()), "-", int(idx[:, 1].max()))
        break
