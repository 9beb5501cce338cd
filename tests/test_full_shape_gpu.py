"""GPU parity at BASELINE.json's full shape (depth 12, dim 1024/1280/512, T=750, nc=16; 776.6 M params) against the
committed oracle vectors tests/golden/sample_full.npz (oracle/make_golden_full.py), one test per BASELINE config:

  configs[0]  steps=4 plumbing case, fp32                         test_full_sample_steps4_fp32_vs_golden
  configs[1]  32-point grid, every grid point, fp32 < 1e-3        test_full_sample_steps32_fp32_every_grid_point
              same grid in the benchmarked bf16 mode (reported)   test_full_sample_steps32_bf16_report
  configs[2]  8 clips per GPU, clip 3 == the B=1 vector           test_config2_eight_clips_per_gpu (steps=4, fp32 + bf16)
              ... on the full 32-point grid, bf16x3 < 1e-3        test_config2_eight_clips_per_gpu_32_steps
  configs[3]  V2P roll, steps=4 vs golden + the 64-point grid of src/inference_v2p.py:183 vs `y_steps64_piano`
              (fp32 and bf16x3 < 1e-3, bf16 reported)              test_config3_v2p_roll_and_64_steps
  A7          the other reading (`rope_cross=True`, rotary in cross-attention): 4 and 32 points, fp32 and bf16x3 < 1e-3
                                                                  test_full_sample_rope_cross_reading
  configs[4]  3 cascaded passes x 4 clips == independent calls    test_config4_cascade_equals_independent_calls

The tolerance for fp32 mode is north_star's |delta mel| < 1e-3; bf16 numbers are printed and bounded at ~1.5x what the mode measures today."""
import os

import numpy as np
import pytest
import torch

from oracle import e2_cfm_oracle as O
from conftest import GOLDEN, make_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    cfg = O.DiTConfig()
    P = O.init_params(cfg, 0)
    y0, text, roll, ctx, cm = O.synthetic_inputs(cfg, 1, 750, nc=16, seed=0)
    g = dict(np.load(os.path.join(GOLDEN, "sample_full.npz"), allow_pickle=False))
    return dict(cfg=cfg, P=P, y0=y0, text=text, roll=roll, ctx=ctx, cm=cm, g={k: torch.from_numpy(v) for k, v in g.items() if k != "meta"})


@pytest.fixture(scope="module")
def m32(full):
    return make_model(full["cfg"], full["P"], "fp32")


@pytest.fixture(scope="module")
def mbf(full):
    return make_model(full["cfg"], full["P"], "bf16")


def _sample(m, f, steps, **over):
    kw = dict(y0=f["y0"], text_embed=f["text"], context=f["ctx"], context_mask=f["cm"], frames_embed=f["roll"], return_raw_output=True,
              steps=steps, cfg_strength=2.0, remove_parallel_component=False)
    kw.update(over)
    return m.sample(torch.zeros(kw["y0"].shape[0], 750, 128), **kw)


def test_full_forward_fp32_vs_golden_and_stats(full, m32, golden):
    f = full
    ref = f["g"]["pred_cond_t037"]
    st = golden["full_stats"]
    assert abs(float(ref.std()) - st["std"]) < 1e-3 and np.allclose(ref[0, :8].numpy(), st["first8"], atol=1e-3)
    got = m32.transformer_with_pred_head(f["y0"], times=torch.tensor(0.37), text=f["text"], frames_embed=f["roll"], context=f["ctx"],
                                         context_mask=f["cm"], drop_text_cond=False, drop_text_prompt=False)
    err = float((got[0] - ref).abs().max())
    print(f"full-shape fp32 forward: max |delta| = {err:.3e}")
    assert err < 1e-3


def test_full_sample_steps4_fp32_vs_golden(full, m32):
    """configs[0]-shaped plumbing case on the GPU: steps=4 (3 evaluations, 6 forwards), CFG 2.0."""
    f = full
    got = _sample(m32, f, 4)
    err = float((got[0] - f["g"]["y_steps4"]).abs().max())
    print(f"full-shape fp32 4-step sample: max |delta mel| = {err:.3e}")
    assert err < 1e-3
    # CFG algebra: strength 0 == the conditional pass alone (null half weight 0), x3:2101-2113
    g0 = _sample(m32, f, 2, cfg_strength=0.0)
    pc = m32.transformer_with_pred_head(f["y0"], times=torch.tensor(0.0), text=f["text"], frames_embed=f["roll"], context=f["ctx"],
                                        context_mask=f["cm"], drop_text_cond=False, drop_text_prompt=False)
    t = O.sway_grid(2)
    assert float((g0 - (f["y0"] + (t[1] - t[0]) * pc)).abs().max()) < 1e-4


def test_full_sample_steps4_vocoder(full, m32):
    """latents -> waveform through the HIP vocoder attached as `vocos` (predict.py:171-172, x3:2277-2289): 240 000 samples."""
    f = full
    from oracle import encodec_oracle as EO
    from v2a_amd import EncodecDecoder
    from v2a_amd.synth import random_encodec_decoder_state_dict
    vsd = random_encodec_decoder_state_dict(7)
    m32.vocos = EncodecDecoder(vsd, "cuda")
    try:
        audio = _sample(m32, f, 4, return_raw_output=None)
    finally:
        m32.vocos = None
    assert isinstance(audio, list) and len(audio) == 1 and audio[0].shape == (240000,)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        wref = EO.decode(vsd, f["g"]["y_steps4"][None].transpose(1, 2))[0]
    werr = float((audio[0].cpu() - wref).abs().max())
    print(f"full-shape fp32 sample + vocoder: max |delta wav| = {werr:.3e} (|wav| max {float(wref.abs().max()):.2f})")
    assert werr < 2e-3


def test_full_sample_steps32_fp32_every_grid_point(full, m32):
    """configs[1] in parity mode: the whole 32-point sway grid (31 CFG evaluations, 62 forwards), every grid point checked
    on every 8th latent frame and the final latents in full, against the CPU restatement: |delta mel| < 1e-3."""
    f = full
    traj = []
    got = _sample(m32, f, 32, trajectory_out=traj)
    assert len(traj) == 32
    sub = torch.stack([t[0, ::8].cpu() for t in traj])
    per_point = (sub - f["g"]["traj32_sub"]).abs().amax(dim=(1, 2))
    err = float((got[0] - f["g"]["y_steps32"]).abs().max())
    print("full-shape fp32 32-step sample: final max |delta mel| = %.3e; per grid point max %.3e (at point %d)"
          % (err, float(per_point.max()), int(per_point.argmax())))
    assert float(per_point.max()) < 1e-3 and err < 1e-3


def test_full_sample_steps32_split_bf16_every_grid_point(full):
    """configs[1] in the bf16x3 mode (split-bf16 GEMMs on the bf16 MFMA path): the same 1e-3 gate over the whole grid."""
    f = full
    m = make_model(f["cfg"], f["P"], "bf16x3")
    traj = []
    got = _sample(m, f, 32, trajectory_out=traj)
    sub = torch.stack([t[0, ::8].cpu() for t in traj])
    per_point = (sub - f["g"]["traj32_sub"]).abs().amax(dim=(1, 2))
    err = float((got[0] - f["g"]["y_steps32"]).abs().max())
    print("full-shape bf16x3 32-step sample: final max |delta mel| = %.3e; per grid point max %.3e" % (err, float(per_point.max())))
    assert float(per_point.max()) < 1e-3 and err < 1e-3


def test_full_sample_steps32_bf16_report(full, mbf):
    """The benchmarked mode on the same grid: drift over 31 evaluations, reported (bf16 operands are narrower than the
    reference's fp32 arithmetic: this mode carries no 1e-3 claim)."""
    f = full
    traj = []
    got = _sample(mbf, f, 32, trajectory_out=traj)
    sub = torch.stack([t[0, ::8].cpu() for t in traj])
    d = (sub - f["g"]["traj32_sub"]).abs()
    err = (got[0] - f["g"]["y_steps32"]).abs()
    print("full-shape bf16 32-step sample: final max |delta mel| = %.4f mean %.5f; per grid point max: %s"
          % (float(err.max()), float(err.mean()), " ".join("%.3f" % v for v in d.amax(dim=(1, 2))[::4])))
    # ~1.5x what this mode measures today (max 0.050 - 0.055, mean 0.0087): an accuracy regression fails, the mode still carries no 1e-3 claim
    assert float(err.max()) < 0.085 and float(err.mean()) < 0.014 and bool(torch.isfinite(got).all())


def test_config2_eight_clips_per_gpu(full, m32, mbf):
    """configs[2] per-GPU shape: 8 clips in one sample() call.  Clip 3 carries the B=1 golden inputs and must reproduce
    the B=1 vector (fp32 < 1e-3); a clip's result does not depend on its batch (checked on clip 5 against a B=1 call)."""
    f = full
    B = 8
    y0, text, roll, ctx, cm = O.synthetic_inputs(f["cfg"], B, 750, nc=16, seed=21)
    y0[3], text[3], roll[3], ctx[3] = f["y0"][0], f["text"][0], f["roll"][0], f["ctx"][0]
    kw = dict(y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll)
    got = _sample(m32, f, 4, **kw)
    err = float((got[3] - f["g"]["y_steps4"]).abs().max())
    one = _sample(m32, f, 4, y0=y0[5:6], text_embed=text[5:6], context=ctx[5:6], context_mask=cm[5:6], frames_embed=roll[5:6])
    dep = float((got[5] - one[0]).abs().max())
    print(f"configs[2] fp32: clip 3 of 8 vs B=1 golden {err:.3e}; clip 5 in batch vs alone {dep:.3e}")
    assert err < 1e-3 and dep < 1e-4
    gb = _sample(mbf, f, 4, **kw)
    eb = (gb[3] - f["g"]["y_steps4"]).abs()
    print(f"configs[2] bf16: clip 3 of 8 vs B=1 golden max {float(eb.max()):.4f} mean {float(eb.mean()):.5f}")
    assert float(eb.max()) < 0.1 and float(eb.mean()) < 0.016 and bool(torch.isfinite(gb).all())


def test_config2_eight_clips_per_gpu_32_steps(full, mbf):
    """configs[2] at its real size: 8 clips per GPU on the full 32-point grid (31 CFG evaluations).  In the parity-grade
    bf16x3 mode clip 3 of the batch carries the B=1 golden inputs and must land on the committed 32-step vector within
    north_star's 1e-3; the same batch in the benchmarked bf16 mode is reported and bounded at ~1.5x today's distance."""
    f = full
    B = 8
    y0, text, roll, ctx, cm = O.synthetic_inputs(f["cfg"], B, 750, nc=16, seed=21)
    y0[3], text[3], roll[3], ctx[3] = f["y0"][0], f["text"][0], f["roll"][0], f["ctx"][0]
    kw = dict(y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll)
    m = make_model(f["cfg"], f["P"], "bf16x3")
    got = _sample(m, f, 32, **kw)
    del m
    err = float((got[3] - f["g"]["y_steps32"]).abs().max())
    print(f"configs[2] bf16x3, 8 clips x 32-point grid: clip 3 vs B=1 golden max |delta mel| = {err:.3e}")
    assert err < 1e-3 and bool(torch.isfinite(got).all())
    gb = _sample(mbf, f, 32, **kw)
    eb = (gb[3] - f["g"]["y_steps32"]).abs()
    dx = (gb - got).abs()
    print(f"configs[2] bf16, 8 clips x 32-point grid: clip 3 vs golden max {float(eb.max()):.4f} mean {float(eb.mean()):.5f}; "
          f"all 8 clips vs bf16x3 max {float(dx.max()):.4f} mean {float(dx.mean()):.5f}")
    assert float(eb.max()) < 0.085 and float(eb.mean()) < 0.014 and float(dx.mean()) < 0.014 and bool(torch.isfinite(gb).all())


def test_config3_v2p_roll_and_64_steps(full, m32, mbf):
    """configs[3]: non-zero piano roll (`piano=True`).  fp32 vs the golden vector at steps=4; then the CLI's 64-point grid
    (src/inference_v2p.py:183; 63 CFG evaluations) against the oracle's `y_steps64_piano`: fp32 and bf16x3 inside north_star's 1e-3,
    the benchmarked bf16 mode reported and bounded; the device step counter must reach 63."""
    f = full
    y0, text, roll, ctx, cm = O.synthetic_inputs(f["cfg"], 1, 750, nc=16, seed=0, piano=True)
    assert float(roll.abs().sum()) > 0
    kw = dict(y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll)
    got = _sample(m32, f, 4, **kw)
    err = float((got[0] - f["g"]["y_steps4_piano"]).abs().max())
    assert float((f["g"]["y_steps4_piano"] - f["g"]["y_steps4"]).abs().max()) > 1e-2      # the roll really conditions the result
    print(f"configs[3] fp32 4-step V2P sample: max |delta mel| = {err:.3e}")
    assert err < 1e-3
    want = f["g"]["y_steps64_piano"]
    e32 = float((_sample(m32, f, 64, **kw)[0] - want).abs().max())
    mx = make_model(f["cfg"], f["P"], "bf16x3")
    ex3 = float((_sample(mx, f, 64, **kw)[0] - want).abs().max())
    del mx
    out = _sample(mbf, f, 64, **kw)
    step = int(mbf.engine().plan["step"].item())
    eb = (out[0] - want).abs()
    print(f"configs[3] 64-point V2P sample vs oracle: fp32 {e32:.3e}, bf16x3 {ex3:.3e}, bf16 max {float(eb.max()):.4f} mean {float(eb.mean()):.5f}; "
          f"step counter {step}")
    assert e32 < 1e-3 and ex3 < 1e-3
    assert step == 63 and bool(torch.isfinite(out).all()) and float(eb.max()) < 0.12 and float(eb.mean()) < 0.02


def test_full_sample_rope_cross_reading(full):
    """The other reading of A7 (`rope_cross=True`: x-transformers applying rotary_pos_emb in cross-attention as the call site x3:1131
    asks; the default follows 1.37.4's `not has_context` guard): 4- and 32-point grids against the oracle vectors of THAT reading,
    every grid point, fp32 and bf16x3 inside 1e-3 -- whichever reading a diff against the real package confirms, it is fixtured."""
    f = full
    g = f["g"]
    assert float((g["y_steps32_ropecross"] - g["y_steps32"]).abs().max()) > 1e-2          # the switch really changes the result
    for mode in ("fp32", "bf16x3"):
        m = make_model(f["cfg"], f["P"], mode, rope_cross=True)
        e4 = float((_sample(m, f, 4)[0] - g["y_steps4_ropecross"]).abs().max())
        traj = []
        got = _sample(m, f, 32, trajectory_out=traj)
        sub = torch.stack([t[0, ::8].cpu() for t in traj])
        per_point = float((sub - g["traj32_sub_ropecross"]).abs().amax(dim=(1, 2)).max())
        err = float((got[0] - g["y_steps32_ropecross"]).abs().max())
        print(f"A7 rope_cross=True {mode}: 4-point {e4:.3e}, 32-point final {err:.3e}, per grid point max {per_point:.3e}")
        assert e4 < 1e-3 and err < 1e-3 and per_point < 1e-3
        del m


def test_config4_cascade_equals_independent_calls(full, mbf):
    """configs[4] (defined by this build, SURVEY 8d: three sequential sample(steps=32) passes): a cascade of three calls on
    one model gives bit-for-bit what three calls on freshly planned models give -- no state leaks from pass to pass (the
    device step counter, modulation tables, cross-attention K/V and the captured graph are re-armed by every call)."""
    f = full
    B = 4                 # configs[4]: batch 32 on 8 GPUs = 4 clips per GPU
    ins = [O.synthetic_inputs(f["cfg"], B, 750, nc=16, seed=40 + i) for i in range(3)]
    def run(m, i):
        y0, text, roll, ctx, cm = ins[i]
        return _sample(m, f, 32, y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll).clone()
    casc = [run(mbf, i) for i in range(3)]
    for i in range(3):
        fresh = make_model(f["cfg"], f["P"], "bf16")
        ind = run(fresh, i)
        assert torch.equal(ind, casc[i]), f"pass {i}: cascade differs from an independent call"
        del fresh
    assert not torch.equal(casc[0], casc[1])


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
@pytest.mark.parametrize("clips,ragged", [(1, False), (2, True)])
def test_bf16_one_launch_cross_attention_equals_two_launches(full, mbf, clips, ragged, mode):
    """bf16 mode, up to two clips: q-projection + cross-attention as one launch (`engine().fuse_xattn`, v2a_qproj_xattn) against the
    GEMM into the [q | gate] buffer followed by v2a_attention -- the same arithmetic, so the sampled latents are equal bit for bit
    (ragged clip lengths and context lengths included)."""
    f = full
    cfg = f["cfg"]
    y0, text, roll, ctx, cm = O.synthetic_inputs(cfg, clips, 750, nc=16, seed=3)
    kw = dict(y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll, return_raw_output=True, steps=4, cfg_strength=2.0,
              remove_parallel_component=False)
    if ragged:
        cm = cm.clone()
        cm[1, 9:] = False
        kw.update(context_mask=cm, lens=torch.tensor([750, 611]), duration=torch.tensor([750, 611]))
    m = mbf if mode == "bf16" else make_model(f["cfg"], f["P"], "bf16x3")
    outs = []
    for fuse in (True, False):
        m.engine().fuse_xattn = fuse
        outs.append(m.sample(torch.zeros(clips, 750, 128), **kw).float().cpu())
    m.engine().fuse_xattn = True
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
