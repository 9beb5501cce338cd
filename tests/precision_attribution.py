#!/usr/bin/env python3
"""Where does bf16 mode's |delta mel| come from?  (needs the instrumented library:
`bash video-to-audio-and-piano-rp_amd/csrc/build.sh --probe`)

The full-shape 32-point sample (BASELINE configs[1], the inputs of tests/golden/sample_full.npz) runs in bf16x3 mode -- every
product hi*hi + hi*lo + lo*hi, max |delta mel| ~ 8e-5 against the CPU restatement -- with ONE class of products at a time
reduced to its hi*hi term, i.e. to exactly the arithmetic bf16 mode uses for that class (bf16 operands, fp32 accumulate);
everything else stays split.  The table therefore attributes the bf16 error class by class, and its last rows show what the
cheapest mixed modes (all classes bf16 except one or two split) would land at.  Classes:
  qkv     fused q|k|v|gate projections of the three streams      attn    QK^T and PV inside the attention kernels
  out     attention out-projections                              ff1     GEGLU feed-forward in
  ff2     feed-forward out                                       cross   TextAudioCrossCondition + U-Net skip GEMMs
  xattn   cross-attention q / out projections + context K/V      pred    to_pred
Lives under tests/ because it borrows the checker's seeded parameters and inputs (oracle/ is test infrastructure).
The probe library reads the switch from v2a_tuning.reserved[0] at launch time, so the sampler runs eagerly (no hipGraph).
usage: python tests/precision_attribution.py [--steps 32]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import v2a_amd  # noqa: E402
from v2a_amd import _lib  # noqa: E402
from oracle import e2_cfm_oracle as O  # noqa: E402  (checker inputs only: config, seeded parameters, synthetic conditioning)

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libv2a_cfm_probe.so")
CLASSES = ["qkv", "attn", "out", "ff1", "ff2", "cross", "xattn", "pred"]


def set_dbg(v):
    t = _lib.Tuning(-1, 0, 1, 0, 0, 0, 0)
    t.reserved[0] = v
    _lib.check(_lib.lib().v2a_set_tuning(C.byref(t)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_precision_attribution.json"))
    a = ap.parse_args()
    cfg = O.DiTConfig()
    P = O.init_params(cfg, 0)
    y0, text, roll, ctx, cm = O.synthetic_inputs(cfg, 1, 750, nc=16, seed=0)
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "sample_full.npz"), allow_pickle=False))
    ref = torch.from_numpy(g["y_steps32" if a.steps == 32 else "y_steps4"])
    m = v2a_amd.E2TTS(transformer=dict(dim=cfg.dim, dim_text=cfg.dim_text, dim_frames=cfg.dim_frames, depth=cfg.depth, heads=cfg.heads,
                                       dim_head=cfg.dim_head, frames_heads=cfg.frames_heads, num_registers=cfg.num_registers,
                                       max_seq_len=cfg.max_seq_len, if_text_modules=True, if_cross_attn=True, if_audio_conv=True, if_text_conv=True),
                      num_channels=cfg.num_channels, sampling_rate=24000, if_cond_proj_in=False, compute_dtype="bf16x3", use_graph=False)
    m.load_state_dict(P, strict=False)
    eng = m.engine()
    # weight tensor -> class
    cls = {}
    for ly in eng.W.layers:
        for s in "atf":
            A, F = ly[f"{s}_attn"], ly[f"{s}_ff"]
            cls[A.w_in.data_ptr()] = "qkv"
            cls[A.w_out.data_ptr()] = "out"
            cls[F.w1.data_ptr()] = "ff1"
            cls[F.w2.data_ptr()] = "ff2"
        cls[ly["a_attn2"].w_in.data_ptr()] = "xattn"
        cls[ly["a_attn2"].w_out.data_ptr()] = "xattn"
        for k in ("x_tfa", "x_at", "x_af", "skip"):
            if k in ly:
                cls[ly[k].data_ptr()] = "cross"
    cls[eng.W.ctx_kv_w.data_ptr()] = "xattn"
    cls[eng.W.pred_w.data_ptr()] = "pred"
    active = set()
    counts = {}
    orig_gemm, orig_attn = _lib.gemm, _lib.attention

    def gemm(a_segs, w, out, **kw):
        c = cls.get(w.data_ptr())
        if kw.get("a_split") and c is None:
            raise RuntimeError("unclassified split GEMM")
        on = c in active
        counts[c] = counts.get(c, 0) + 1
        if on:
            set_dbg(32)
        orig_gemm(a_segs, w, out, **kw)
        if on:
            set_dbg(0)

    def attention(*args, **kw):
        on = "attn" in active
        if on:
            set_dbg(32)
        orig_attn(*args, **kw)
        if on:
            set_dbg(0)

    _lib.gemm, _lib.attention = gemm, attention

    def run(degraded):
        active.clear()
        active.update(degraded)
        t0 = time.time()
        y = m.sample(torch.zeros(1, 750, 128), y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll,
                     steps=a.steps, cfg_strength=2.0, remove_parallel_component=False, return_raw_output=True)
        d = (y[0] - ref).abs()
        return float(d.max()), float(d.mean()), time.time() - t0

    rows = []
    def rec(name, degraded):
        mx, mean, dt = run(degraded)
        rows.append(dict(case=name, bf16_classes=sorted(degraded), max_abs_delta_mel=mx, mean_abs_delta_mel=mean))
        print(f"{name:46s} max |delta mel| {mx:.3e}  mean {mean:.3e}   ({dt:.1f} s)", flush=True)

    rec("all split (bf16x3 mode)", [])
    for c in CLASSES:
        rec(f"only {c} in bf16", [c])
    rec("all classes in bf16 (= bf16 products everywhere)", CLASSES)
    for c in CLASSES:
        rec(f"all bf16 except {c} split", [x for x in CLASSES if x != c])
    for pair in (("ff1", "ff2"), ("qkv", "attn"), ("qkv", "out"), ("cross", "ff2"), ("qkv", "ff1"), ("ff1", "cross")):
        rec(f"all bf16 except {pair[0]} + {pair[1]} split", [x for x in CLASSES if x not in pair])
    json.dump(dict(steps=a.steps, shape="BASELINE configs[1]: 1 clip x 750 frames, CFG 2.0", gemm_launches_by_class=counts, rows=rows),
              open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
