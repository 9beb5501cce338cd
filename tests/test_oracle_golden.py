"""CPU: the oracle against its committed golden vectors, plus the algebraic properties the
accelerated path relies on.  No GPU, no HIP library calls."""
import numpy as np
import pytest
import torch

from oracle import e2_cfm_oracle as O


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_param_fingerprint_matches_fixture(small, golden):
    P = small["P"]
    keys = sorted(P)
    fp = np.array([[float(P[k].double().sum()), float(P[k].double().abs().sum())] for k in keys])
    np.testing.assert_allclose(fp, golden["forward_small"]["param_fingerprint"], rtol=1e-9, atol=1e-9)


def test_param_count_full_shape(golden):
    shapes = O.param_shapes(O.DiTConfig())
    n = sum(int(np.prod(s)) for s in shapes.values())
    assert n == golden["full_stats"]["n_params"] == 776583456       # ~776.6 M


@pytest.mark.parametrize("layout", ["interleaved", "half"])
def test_forward_matches_golden(small, golden, layout):
    cfg, P, i = small["cfg"], small["P"], small["inp"]
    g = golden["forward_small"]
    opts = O.OracleOptions(rope_layout=layout)
    with torch.no_grad():
        pc = O.transformer_with_pred_head(P, cfg, i["y0"], torch.tensor(0.37), None, i["text"], i["roll"], i["ctx"],
                                          i["ctx_mask"], drop_text_cond=False, drop_text_prompt=False, opts=opts)
        pn = O.transformer_with_pred_head(P, cfg, i["y0"], torch.tensor(0.37), None, i["text"], i["roll"], i["ctx"],
                                          i["ctx_mask"], drop_text_cond=True, drop_text_prompt=True, opts=opts)
    np.testing.assert_allclose(pc.numpy(), g[f"pred_cond_{layout}"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(pn.numpy(), g[f"pred_null_{layout}"], atol=2e-5, rtol=0)


def test_sample_matches_golden(small, golden):
    cfg, P, i = small["cfg"], small["P"], small["inp"]
    g = golden["sample_small"]
    kw = dict(steps=4, cfg_strength=2.0, sway_sampling=True, remove_parallel_component=False)
    y = O.sample(P, cfg, i["y0"], i["text"], i["roll"], i["ctx"], i["ctx_mask"], **kw)
    np.testing.assert_allclose(y.numpy(), g["y_full"], atol=5e-5, rtol=0)
    yr = O.sample(P, cfg, i["y0"], i["text"], i["roll"], i["ctx"], i["ctx_mask"], duration=[40, 29], **kw)
    np.testing.assert_allclose(yr.numpy(), g["y_ragged"], atol=5e-5, rtol=0)
    # clip 0 has full length in both runs: batching with a shorter clip must not change it
    np.testing.assert_allclose(yr[0].numpy(), g["y_full"][0], atol=5e-5, rtol=0)


def test_sway_grid(golden):
    g = golden["sample_small"]
    np.testing.assert_array_equal(O.sway_grid(4).numpy(), g["sway_grid_4"])
    t = O.sway_grid(32)
    np.testing.assert_array_equal(t.numpy(), g["sway_grid_32"])
    assert t[0] == 0 and abs(float(t[-1]) - 1.0) < 1e-6 and bool((t[1:] > t[:-1]).all())
    # x3:2252 with coefficient -1: t + -(cos(pi/2 t) - 1 + t) == 1 - cos(pi/2 t)
    lin = torch.linspace(0, 1, 32)
    np.testing.assert_allclose(t.numpy(), (1 - torch.cos(torch.pi / 2 * lin)).numpy(), atol=1e-6)


def test_blocks_match_golden(small, golden):
    P, cfg = small["P"], small["cfg"]
    b = golden["blocks_small"]
    x, c, mask = _t(b["x"]), _t(b["c"]), _t(b["mask"])
    L0 = "transformer.layers.0"
    with torch.no_grad():
        np.testing.assert_allclose(O.depthwise_conv(x, P[f"{L0}.0.1.dw_conv1d.0.weight"], P[f"{L0}.0.1.dw_conv1d.0.bias"], mask).numpy(),
                                   b["dwconv"], atol=1e-5)
        np.testing.assert_allclose(O.adaptive_rmsnorm(x, P[f"{L0}.0.2.to_gamma.weight"], c).numpy(), b["ada_rmsnorm"], atol=1e-5)
        np.testing.assert_allclose(O.feedforward(P, f"{L0}.0.9", x).numpy(), b["feedforward"], atol=1e-5)
        fr = O.rotary_freqs(44, 64, "interleaved")
        np.testing.assert_allclose(O.attention(P, f"{L0}.0.3", x, cfg.heads, 64, fr, mask, O.OracleOptions()).numpy(),
                                   b["self_attn"], atol=1e-5)


# ---- properties the restructured GPU path depends on ------------------------------------------

def test_null_pass_cross_attention_is_exactly_zero(small):
    """context == 0 with bias-free to_k/to_v/to_out => cross-attention adds exactly 0 (dit.py skips it)."""
    P, cfg = small["P"], small["cfg"]
    x = torch.randn(2, 44, cfg.dim)
    ctx = torch.zeros(2, 5, cfg.dim)
    cm = torch.ones(2, 5, dtype=torch.bool)
    fr = O.rotary_freqs(44, 64, "interleaved")
    out = O.attention(P, "transformer.layers.1.0.6", x, cfg.heads, 64, fr, None, O.OracleOptions(), context=ctx, context_mask=cm)
    assert float(out.abs().max()) == 0.0


def test_layer0_side_streams_do_not_depend_on_x_or_t(small):
    """The hoist of layer 0's text/frames blocks out of the Euler loop (dit.py prepare())."""
    cfg, P, i = small["cfg"], small["P"], small["inp"]
    taps_a, taps_b = {}, {}
    with torch.no_grad():
        O.transformer_with_pred_head(P, cfg, i["y0"], torch.tensor(0.1), None, i["text"], i["roll"], i["ctx"], i["ctx_mask"],
                                     drop_text_cond=False, drop_text_prompt=False, taps=taps_a)
        O.transformer_with_pred_head(P, cfg, i["y0"] * 3 + 1, torch.tensor(0.9), None, i["text"], i["roll"], i["ctx"], i["ctx_mask"],
                                     drop_text_cond=False, drop_text_prompt=False, taps=taps_b)
    # text_l0 / frames_l0 are post-cross-condition (depend on x); their x-independent part is checked by
    # recomputing the blocks directly from the stream inputs
    for k in ("text0", "frames0"):
        assert torch.equal(taps_a[k], taps_b[k])
    assert not torch.equal(taps_a["x_l0"], taps_b["x_l0"])


def test_rope_layouts_are_a_head_dim_permutation(small):
    """A6: the two pair layouts give identical attention logits when q/k head dims are permuted
    accordingly -- so the choice only matters for trained checkpoints, not for seeded weights."""
    q = torch.randn(1, 2, 11, 64)
    k = torch.randn(1, 2, 11, 64)
    perm = torch.cat([torch.arange(0, 64, 2), torch.arange(1, 64, 2)])     # interleaved -> half
    fi, fh = O.rotary_freqs(11, 64, "interleaved"), O.rotary_freqs(11, 64, "half")
    qi, ki = O.apply_rope(q, fi, "interleaved"), O.apply_rope(k, fi, "interleaved")
    qh, kh = O.apply_rope(q[..., perm], fh, "half"), O.apply_rope(k[..., perm], fh, "half")
    torch.testing.assert_close(qi @ ki.transpose(-1, -2), qh @ kh.transpose(-1, -2), atol=1e-4, rtol=1e-4)


def test_cfg_zero_strength_is_single_pass(small):
    cfg, P, i = small["cfg"], small["P"], small["inp"]
    a = (P, cfg, i["y0"], torch.tensor(0.5), None, i["text"], i["roll"], i["ctx"], i["ctx_mask"])
    with torch.no_grad():
        p0 = O.cfg_pred(*a, cfg_strength=0.0)
        pc = O.transformer_with_pred_head(*a, drop_text_cond=False, drop_text_prompt=False)
    assert torch.equal(p0, pc)


def test_apg_projection_decomposes(small):
    x, y = torch.randn(2, 7, 5), torch.randn(2, 7, 5)
    par, orth = O.apg_project(x, y)
    torch.testing.assert_close(par + orth, x, atol=1e-6, rtol=1e-6)
    assert float((orth.reshape(2, -1) * y.reshape(2, -1)).sum(-1).abs().max()) < 1e-4


def test_euler_is_linear_in_steps(small):
    """A12: one Euler step from t0 to t1 equals y + (t1 - t0) f(t0, y)."""
    cfg, P, i = small["cfg"], small["P"], small["inp"]
    y, traj = O.sample(P, cfg, i["y0"], i["text"], i["roll"], i["ctx"], i["ctx_mask"], steps=3, cfg_strength=2.0,
                       remove_parallel_component=False, return_trajectory=True)
    t = O.sway_grid(3)
    with torch.no_grad():
        f0 = O.cfg_pred(P, cfg, i["y0"], t[0], torch.ones(2, 40, dtype=torch.bool), i["text"], i["roll"], i["ctx"], i["ctx_mask"],
                        cfg_strength=2.0, remove_parallel_component=False)
    torch.testing.assert_close(traj[1], i["y0"] + (t[1] - t[0]) * f0, atol=1e-5, rtol=1e-5)
    assert len(traj) == 3 and torch.equal(traj[-1], y)


def test_oracle_audio_prompt_fixture_and_properties():
    """The audio-prompted branch (x3:2015-2035, 2196-2231, 2260-2261): the oracle reproduces its committed vectors; the prompt
    frames come back unchanged; with lens == duration `cond` is ignored; a dropped prompt changes the result."""
    import json, os
    from conftest import GOLDEN
    g = dict(np.load(os.path.join(GOLDEN, "sample_small_prompt.npz"), allow_pickle=False))
    cfg = O.DiTConfig(**json.loads(str(g["meta"]))["cfg"])
    P = O.init_params(cfg, 1234)
    t = lambda k: torch.from_numpy(g[k])
    kw = dict(steps=4, cfg_strength=2.0, sway_sampling=True, remove_parallel_component=False)
    y = O.sample(P, cfg, t("y0"), t("text"), t("roll"), t("ctx"), t("ctx_mask"), duration=t("duration"), cond=t("cond"), lens=t("lens"), **kw)
    np.testing.assert_allclose(y.numpy(), g["y_prompt"], atol=1e-5)
    for b in range(2):
        n = int(g["lens"][b])
        assert np.array_equal(y.numpy()[b, :n], g["cond"][b, :n])
    assert np.abs(g["y_prompt"] - g["y_prompt_audio_drop"])[1, 20:33].max() > 1e-3 and np.abs(g["y_prompt"] - g["y_prompt_audio_drop"])[0].max() < 1e-6
    same = O.sample(P, cfg, t("y0"), t("text"), t("roll"), t("ctx"), t("ctx_mask"), duration=t("duration"), cond=t("cond"), lens=t("duration"), **kw)
    none = O.sample(P, cfg, t("y0"), t("text"), t("roll"), t("ctx"), t("ctx_mask"), duration=t("duration"), **kw)
    assert torch.equal(same, none)


# ---- the oracle's generic arithmetic against torch's own independent implementations ---------------------------------------------
# (This does NOT pin the x-transformers-specific choices A1-A13 -- those stay unpinned, SURVEY 8c -- it rules out restatement slips in the
# parts any implementation must agree on: the attention core, the depthwise convolution, the GEGLU feed-forward, the RMS normalisation.)

def test_attention_core_matches_torch_sdpa(small):
    """softclamp off, head gates saturated open (weight 0, bias +40: sigmoid == 1 in fp32), no rotary: what is left is
    softmax(q k^T / sqrt(d) | key mask) v followed by to_out -- torch.nn.functional.scaled_dot_product_attention on the same projections."""
    import torch.nn.functional as F
    P, cfg = dict(small["P"]), small["cfg"]
    pre = "transformer.layers.1.0.3"
    P[f"{pre}.to_v_head_gate.weight"] = torch.zeros_like(P[f"{pre}.to_v_head_gate.weight"])
    P[f"{pre}.to_v_head_gate.bias"] = torch.full_like(P[f"{pre}.to_v_head_gate.bias"], 40.0)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 44, cfg.dim, generator=g)
    mask = O.lens_to_mask(torch.tensor([44, 30]), 44)
    with torch.no_grad():
        got = O.attention(P, pre, x, cfg.heads, 64, None, mask, O.OracleOptions(softclamp=0.0, zero_masked_queries=False))
        split = lambda t: t.reshape(2, 44, cfg.heads, 64).permute(0, 2, 1, 3)
        q, k, v = (split(F.linear(x, P[f"{pre}.to_{n}.weight"])) for n in "qkv")
        ref = F.scaled_dot_product_attention(q, k, v, attn_mask=mask[:, None, None, :])
        ref = F.linear(ref.permute(0, 2, 1, 3).reshape(2, 44, -1), P[f"{pre}.to_out.weight"])
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-5)


def test_depthwise_conv_matches_torch_conv1d(small):
    """DepthwiseConv (x3:495-528) on an unmasked batch: Conv1d(groups = dim, padding = k // 2) followed by SiLU."""
    import torch.nn.functional as F
    P = small["P"]
    w, b = P["transformer.layers.0.0.1.dw_conv1d.0.weight"], P["transformer.layers.0.0.1.dw_conv1d.0.bias"]
    x = torch.randn(2, 44, w.shape[0], generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        got = O.depthwise_conv(x, w, b, None)
        ref = F.silu(F.conv1d(x.transpose(1, 2), w, b, padding=w.shape[-1] // 2, groups=w.shape[0])).transpose(1, 2)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=1e-5)


def test_feedforward_and_rmsnorm_match_torch_modules(small):
    import torch.nn.functional as F
    P, cfg = small["P"], small["cfg"]
    pre = "transformer.layers.0.0.9"
    x = torch.randn(3, 7, cfg.dim, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        a, gt = F.linear(x, P[f"{pre}.ff.0.proj.weight"], P[f"{pre}.ff.0.proj.bias"]).chunk(2, dim=-1)
        ref = F.linear(a * F.gelu(gt), P[f"{pre}.ff.2.weight"], P[f"{pre}.ff.2.bias"])
        np.testing.assert_allclose(O.feedforward(P, pre, x).numpy(), ref.numpy(), atol=1e-5)
        g = 1 + 0.1 * torch.randn(cfg.dim, generator=torch.Generator().manual_seed(6))
        # x / rms(x) * g: torch's rms_norm with eps -> 0 (x-transformers' F.normalize form clamps the NORM at 1e-12 instead)
        np.testing.assert_allclose(O.rmsnorm(x, g).numpy(), F.rms_norm(x, (cfg.dim,), g, eps=1e-30).numpy(), atol=1e-5)
