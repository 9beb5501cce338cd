"""Generates tests/golden/*.npz from the CPU oracle.          *** TEST INFRASTRUCTURE ***

The reference holds no golden vectors for this path and cannot be imported here,
so these fixtures pin the ORACLE (against regressions and against the torch build on the GPU
box), not the reference: parity stays "unpinned" in the sense of the task statement.

Every zero-initialised parameter of the reference is given seeded non-trivial values
(init_params), both rotary pair layouts (A6) are emitted, and one fixture exercises ragged
durations (mask paths), one the APG projection branch of CFG.

Usage:  python oracle/make_golden.py        (writes tests/golden/)
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import e2_cfm_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

SMALL = dict(dim=128, dim_text=192, dim_frames=64, depth=4, heads=2, frames_heads=1, num_registers=4,
             num_channels=16, max_seq_len=256)
PARAM_SEED, INPUT_SEED = 1234, 99
B, T, NC = 2, 40, 5


def param_fingerprint(P):
    """Order-stable checksum so a consumer can verify its seeded regeneration of the weights."""
    keys = sorted(P)
    return np.array([[float(P[k].double().sum()), float(P[k].double().abs().sum())] for k in keys])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg = O.DiTConfig(**SMALL)
    P = O.init_params(cfg, PARAM_SEED)
    y0, text, roll, ctx, cm = O.synthetic_inputs(cfg, B, T, nc=NC, seed=INPUT_SEED, piano=True)
    cm[1, -2:] = False                                   # ragged context mask on clip 1
    meta = dict(cfg=SMALL, param_seed=PARAM_SEED, input_seed=INPUT_SEED, B=B, T=T, nc=NC)
    common = dict(y0=y0.numpy(), text=text.numpy(), roll=roll.numpy(), ctx=ctx.numpy(), ctx_mask=cm.numpy(),
                  param_fingerprint=param_fingerprint(P), meta=json.dumps(meta))

    # 1. one forward (cond + null), both rotary layouts, with per-layer taps
    fw = dict(common)
    tval = torch.tensor(0.37)
    for lay in ("interleaved", "half"):
        opts = O.OracleOptions(rope_layout=lay)
        taps = {}
        with torch.no_grad():
            pc = O.transformer_with_pred_head(P, cfg, y0, tval, None, text, roll, ctx, cm, drop_text_cond=False,
                                              drop_text_prompt=False, opts=opts, taps=taps)
            pn = O.transformer_with_pred_head(P, cfg, y0, tval, None, text, roll, ctx, cm, drop_text_cond=True,
                                              drop_text_prompt=True, opts=opts)
        fw[f"pred_cond_{lay}"] = pc.numpy()
        fw[f"pred_null_{lay}"] = pn.numpy()
        if lay == "interleaved":
            for k, v in taps.items():
                fw[f"tap_{k}"] = v.numpy()
    fw["t"] = np.float32(0.37)
    np.savez_compressed(os.path.join(OUT, "forward_small.npz"), **fw)

    # 2. 4-step CFG samples: full length, ragged durations, video_drop_prompt, APG branch, the other A7 reading (rope in cross-attention)
    sm = dict(common)
    kw = dict(steps=4, cfg_strength=2.0, sway_sampling=True)
    with torch.no_grad():
        sm["y_full"] = O.sample(P, cfg, y0, text, roll, ctx, cm, remove_parallel_component=False, **kw).numpy()
        sm["y_half_layout"] = O.sample(P, cfg, y0, text, roll, ctx, cm, remove_parallel_component=False,
                                       opts=O.OracleOptions(rope_layout="half"), **kw).numpy()
        sm["y_ragged"] = O.sample(P, cfg, y0, text, roll, ctx, cm, duration=[T, 29], remove_parallel_component=False, **kw).numpy()
        sm["y_dropprompt"] = O.sample(P, cfg, y0, text, roll, ctx, cm, remove_parallel_component=False,
                                      video_drop_prompt=[False, True], **kw).numpy()
        sm["y_apg"] = O.sample(P, cfg, y0, text, roll, ctx, cm, remove_parallel_component=True, **kw).numpy()
        sm["y_rope_cross"] = O.sample(P, cfg, y0, text, roll, ctx, cm, remove_parallel_component=False,
                                      opts=O.OracleOptions(rope_cross=True), **kw).numpy()
        sm["y_steps8_nosway"] = O.sample(P, cfg, y0, text, roll, ctx, cm, steps=8, cfg_strength=3.0,
                                         sway_sampling=False, remove_parallel_component=False).numpy()
    sm["ragged_duration"] = np.array([T, 29])
    sm["sway_grid_4"] = O.sway_grid(4).numpy()
    sm["sway_grid_32"] = O.sway_grid(32).numpy()
    np.savez_compressed(os.path.join(OUT, "sample_small.npz"), **sm)

    # 3. block-level vectors (one per kernel family)
    g = torch.Generator().manual_seed(7)
    blk = {}
    x = torch.randn(2, 44, 128, generator=g)
    c = torch.randn(2, 128, generator=g)
    mask = O.lens_to_mask(torch.tensor([44, 30]), 44)
    L0 = "transformer.layers.0"
    freqs = O.rotary_freqs(44, 64, "interleaved")
    with torch.no_grad():
        blk["x"], blk["c"], blk["mask"] = x.numpy(), c.numpy(), mask.numpy()
        blk["fourier"] = O.fourier_embed(torch.tensor([0.0, 0.37, 1.0]), P["transformer.time_cond_mlp.0.weights"]).numpy()
        blk["time_cond"] = O.time_cond(P, torch.tensor([0.0, 0.37, 1.0])).numpy()
        blk["dwconv"] = O.depthwise_conv(x, P[f"{L0}.0.1.dw_conv1d.0.weight"], P[f"{L0}.0.1.dw_conv1d.0.bias"], mask).numpy()
        blk["ada_rmsnorm"] = O.adaptive_rmsnorm(x, P[f"{L0}.0.2.to_gamma.weight"], c).numpy()
        blk["adaln_zero"] = O.adaln_zero(x, P[f"{L0}.0.4.to_gamma.weight"], P[f"{L0}.0.4.to_gamma.bias"], c).numpy()
        blk["self_attn"] = O.attention(P, f"{L0}.0.3", x, cfg.heads, 64, freqs, mask, O.OracleOptions()).numpy()
        ctxb = torch.randn(2, 5, 128, generator=g)
        cmb = torch.tensor([[True] * 5, [True] * 3 + [False] * 2])
        blk["ctx"], blk["ctx_mask"] = ctxb.numpy(), cmb.numpy()
        blk["cross_attn"] = O.attention(P, f"{L0}.0.6", x, cfg.heads, 64, freqs, mask, O.OracleOptions(),
                                        context=ctxb, context_mask=cmb).numpy()
        blk["cross_attn_rope"] = O.attention(P, f"{L0}.0.6", x, cfg.heads, 64, freqs, mask, O.OracleOptions(rope_cross=True),
                                             context=ctxb, context_mask=cmb).numpy()
        blk["feedforward"] = O.feedforward(P, f"{L0}.0.9", x).numpy()
        tx = torch.randn(2, 44, 192, generator=g)
        fr = torch.randn(2, 44, 64, generator=g)
        a2, t2, f2 = O.cross_condition(P, f"{L0}.1.5", x, tx, fr, True)
        blk["tx"], blk["fr"] = tx.numpy(), fr.numpy()
        blk["cc_audio"], blk["cc_text"], blk["cc_frames"] = a2.numpy(), t2.numpy(), f2.numpy()
        q = torch.randn(1, 2, 44, 64, generator=g)
        blk["rope_in"] = q.numpy()
        blk["rope_interleaved"] = O.apply_rope(q, O.rotary_freqs(44, 64, "interleaved"), "interleaved").numpy()
        blk["rope_half"] = O.apply_rope(q, O.rotary_freqs(44, 64, "half"), "half").numpy()
        blk["rope_last5_interleaved"] = O.apply_rope(q[:, :, :5], O.rotary_freqs(44, 64, "interleaved"), "interleaved").numpy()
    np.savez_compressed(os.path.join(OUT, "blocks_small.npz"), **blk)

    # 4. full-shape summary statistics (one forward at the BASELINE shape; values, not arrays)
    if "--full" in sys.argv:
        cfgF = O.DiTConfig()
        PF = O.init_params(cfgF, 0)
        y0F, textF, rollF, ctxF, cmF = O.synthetic_inputs(cfgF, 1, 750, nc=16, seed=0)
        with torch.no_grad():
            pc = O.transformer_with_pred_head(PF, cfgF, y0F, torch.tensor(0.37), None, textF, rollF, ctxF, cmF,
                                              drop_text_cond=False, drop_text_prompt=False)
        stats = dict(mean=float(pc.mean()), std=float(pc.std()), absmean=float(pc.abs().mean()),
                     first8=[float(v) for v in pc[0, 0, :8]], last8=[float(v) for v in pc[0, -1, -8:]],
                     n_params=int(sum(v.numel() for v in PF.values())))
        with open(os.path.join(OUT, "forward_full_stats.json"), "w") as f:
            json.dump(stats, f, indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
