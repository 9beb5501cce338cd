"""GPU parity of the Video2Roll frame encoder through the C ABI.

Checked against (a) vectors produced by the REFERENCE module (tests/golden/video2roll_*.npz, made by
oracle/make_golden_video2roll.py running src/audeo/Video2RollNet.py), (b) the CPU restatement
oracle/video2roll_oracle.py on seeded inputs, (c) plain torch ops for the individual kernels.
Tolerances: fp32 mode -- logits (|x| up to ~17) 2e-3 abs, probabilities 1e-4 abs, feature maps 1e-4 rel
(fp32 summation-order and BatchNorm-folding noise only); bf16 mode -- stated in each test."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import video2roll_oracle as VO

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
PARAM_SEED, INPUT_SEED = 4321, 77


@pytest.fixture(scope="module")
def L():
    from v2a_amd import _lib
    _lib.lib()
    return _lib


@pytest.fixture(scope="module")
def params():
    from v2a_amd.synth import random_video2roll_state_dict
    return random_video2roll_state_dict(PARAM_SEED)


@pytest.fixture(scope="module")
def engines(params):
    from v2a_amd.video2roll import Video2RollEngine
    return {c: Video2RollEngine(params, DEV, compute=c, chunk=3) for c in ("fp32", "bf16")}


def _g(seed=0):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------------------- im2col
@pytest.mark.parametrize("geom", [(3, 3, 1, 1), (3, 3, 2, 1), (1, 1, 1, 1), (1, 1, 2, 0), (1, 1, 1, 0)])
@pytest.mark.parametrize("odt", [torch.float32, torch.bfloat16])
def test_im2col_nhwc_matches_unfold(L, geom, odt):
    kh, kw, stride, pad = geom
    B, H, W, C = 2, 7, 13, 64
    x = torch.randn(B, C, H, W, generator=_g(kh * 10 + stride))
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    K = kh * kw * C
    kpad = (K + 63) // 64 * 64 + 64                              # one extra zero block
    col = torch.full((B * Ho * Wo, kpad), 7.0, dtype=odt, device=DEV)
    L.im2col(x.permute(0, 2, 3, 1).contiguous().to(DEV), col, B=B, H=H, W=W, C_=C, kh=kh, kw=kw, stride=stride, pad=pad,
             Ho=Ho, Wo=Wo, ldo=kpad)
    # F.unfold orders k as (c, ky, kx); ours is (ky, kx, c)
    u = F.unfold(x, (kh, kw), padding=pad, stride=stride).view(B, C, kh * kw, Ho * Wo).permute(0, 3, 2, 1).reshape(B * Ho * Wo, K)
    ref = u if odt == torch.float32 else u.to(torch.bfloat16).float()
    got = col.float().cpu()
    assert torch.equal(got[:, :K], ref)                          # a gather: exact (bf16: same rounding)
    assert torch.all(got[:, K:] == 0)


@pytest.mark.parametrize("odt", [torch.float32, torch.bfloat16])
def test_im2col_window_gathers_clamped_frames(L, odt):
    clips, T, H, W = 2, 4, 12, 21
    frames = torch.rand(clips, T, H, W, generator=_g(2))
    kh = kw = 11
    stride, pad = 2, 4
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    K = 5 * kh * kw
    kpad = (K + 63) // 64 * 64
    win = VO.frame_windows(frames[:, None])                       # (clips*T, 5, H, W), x3:1531-1539 restated
    u = F.unfold(win, (kh, kw), padding=pad, stride=stride).permute(0, 2, 1).reshape(clips * T * Ho * Wo, K)
    ref = u if odt == torch.float32 else u.to(torch.bfloat16).float()
    for first, n in ((0, clips * T), (3, 4)):                     # all windows; a chunk straddling the clip boundary
        col = torch.full((n * Ho * Wo, kpad), 7.0, dtype=odt, device=DEV)
        L.im2col(frames.to(DEV), col, B=n, H=H, W=W, C_=5, kh=kh, kw=kw, stride=stride, pad=pad, Ho=Ho, Wo=Wo, ldo=kpad,
                 window_t=T, window_first=first)
        got = col.float().cpu()
        assert torch.equal(got[:, :K], ref[first * Ho * Wo:(first + n) * Ho * Wo])
        assert torch.all(got[:, K:] == 0)


def test_im2col_rejects_bad_geometry(L):
    x = torch.zeros(1, 4, 4, 8, device=DEV)
    col = torch.zeros(16, 128, device=DEV)
    with pytest.raises(L.V2AError, match="Ho/Wo"):
        L.im2col(x, col, B=1, H=4, W=4, C_=8, kh=3, kw=3, stride=1, pad=1, Ho=3, Wo=4, ldo=128)
    with pytest.raises(L.V2AError, match="C %% 4|C % 4"):
        L.im2col(x, col, B=1, H=4, W=4, C_=6, kh=1, kw=1, stride=1, pad=0, Ho=4, Wo=4, ldo=128)


# ------------------------------------------------------------------------------- pooling
@pytest.mark.parametrize("cfg", [(3, 2, 1, 0), (2, 2, 0, 1), (3, 1, 0, 1)])
def test_pool2d_matches_torch(L, cfg):
    k, stride, pad, mode = cfg
    B, H, W, C = 2, 9, 15, 64
    x = torch.randn(B, C, H, W, generator=_g(k))
    ref = F.max_pool2d(x, k, stride, pad) if mode == 0 else F.avg_pool2d(x, k, stride)
    Ho, Wo = ref.shape[2:]
    out = torch.empty(B, Ho, Wo, C, device=DEV)
    L.pool2d(x.permute(0, 2, 3, 1).contiguous().to(DEV), out, B=B, H=H, W=W, C_=C, k=k, stride=stride, pad=pad, mode=mode, Ho=Ho, Wo=Wo)
    torch.testing.assert_close(out.cpu().permute(0, 3, 1, 2), ref, atol=1e-6, rtol=1e-6)


# ------------------------------------------------------------------------------- GEMM relu flag
@pytest.mark.parametrize("compute", ["fp32", "bf16"])
@pytest.mark.parametrize("N", [64, 128, 512])
def test_gemm_relu_and_residual_relu(L, compute, N):
    M, K = 700, 576
    cd = torch.float32 if compute == "fp32" else torch.bfloat16
    code = L.F32 if compute == "fp32" else L.BF16
    a = torch.randn(M, K, generator=_g(1)).to(cd)
    w = (torch.randn(N, K, generator=_g(2)) / K ** 0.5).to(cd)
    b = torch.randn(N, generator=_g(3))
    r = torch.randn(M, N, generator=_g(4))
    acc = a.float() @ w.float().t() + b
    tol = 2e-5 if compute == "fp32" else 2e-3
    out = torch.empty(M, N, device=DEV)
    L.gemm([(a.to(DEV), K, K)], w.to(DEV), out, M=M, N=N, compute=code, bias=b.to(DEV), relu=True)
    torch.testing.assert_close(out.cpu(), acc.clamp_min(0), atol=tol, rtol=tol)
    L.gemm([(a.to(DEV), K, K)], w.to(DEV), out, M=M, N=N, compute=code, epilogue=L.EPI_RESID, bias=b.to(DEV), resid=r.to(DEV), relu=True)
    torch.testing.assert_close(out.cpu(), (acc + r).clamp_min(0), atol=tol, rtol=tol)
    assert (out >= 0).all()


# ------------------------------------------------------------------------------- first layer as an implicit GEMM
def test_frames_pack_and_offset_tables_give_the_first_layer(L):
    """v2a_frames_pack + v2a_gemm offset tables == Conv2d(5, 64, 11, stride 2, pad 4) on the 5-frame clamped windows
    (v2r:138, x3:1531-1539), bf16 operands."""
    T, H, W, kh, stride, pad, co = 6, 20, 37, 11, 2, 4, 64
    g = _g(21)
    frames = torch.rand(T, H, W, generator=g)
    w = (torch.randn(co, 5, kh, kh, generator=g) / (5 * kh * kh) ** 0.5).to(torch.bfloat16)
    bias = torch.randn(co, generator=g)
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kh) // stride + 1
    Hp = H + 2 * pad
    packed = torch.full(((T + 4) * Wo * Hp * 16,), 3.0, dtype=torch.bfloat16, device=DEV)
    L.frames_pack(frames.to(DEV), packed, T=T, H=H, W=W, kw=kh, stride=stride, pad=pad, Wo=Wo)
    pk = packed.float().cpu().view(T + 4, Wo, Hp, 16)
    fp = F.pad(frames.to(torch.bfloat16).float(), (pad, pad + 16, pad, pad))                    # zero border (+ slack on the right)
    idx = (torch.arange(T + 4) - 2).clamp(0, T - 1)
    for xo in (0, 1, Wo // 2, Wo - 1):
        ref = fp[idx][:, :, stride * xo: stride * xo + 16].clone()
        ref[:, :, kh:] = 0
        assert torch.equal(pk[:, xo], ref)
    win = VO.frame_windows(frames[None, None])                                                   # (T, 5, H, W)
    ref = F.relu(F.conv2d(win.to(torch.bfloat16).float(), w.float(), bias, stride, pad))         # (T, co, Ho, Wo)
    first, n, gk = 1, 4, 3
    wq = torch.zeros(co, 5, gk * 4, 16, dtype=torch.bfloat16)
    wq[:, :, :kh, :kh] = w
    ni, yo, xo = torch.arange(first, first + n)[:, None, None], torch.arange(Ho)[None, :, None], torch.arange(Wo)[None, None, :]
    a_row = (((ni * Wo + xo) * Hp + yo * stride) * 16).reshape(-1).int()
    kt = torch.arange(5 * gk)
    a_k = ((kt // gk) * (Wo * Hp * 16) + (kt % gk) * 64).int()
    out = torch.empty(n * Ho * Wo, co, device=DEV)
    K = 5 * gk * 64
    L.gemm([(packed, K, K)], wq.reshape(co, K).contiguous().to(DEV), out, M=n * Ho * Wo, N=co, compute=L.BF16, bias=bias.to(DEV), relu=True,
           ldo=co, a_row_offset=a_row.to(DEV), a_ktile_offset=a_k.to(DEV))
    got = out.cpu().view(n, Ho, Wo, co).permute(0, 3, 1, 2)
    torch.testing.assert_close(got, ref[first:first + n], atol=2e-3, rtol=2e-3)


# ------------------------------------------------------------------------------- implicit-GEMM convolution (offset tables)
@pytest.mark.parametrize("geom", [(3, 1, 1), (3, 2, 1), (1, 2, 0), (1, 1, 1)])
@pytest.mark.parametrize("cout", [64, 128])
def test_gemm_offset_tables_are_a_convolution(L, geom, cout):
    """v2a_gemm with a_row_offset / a_ktile_offset / out_row_offset on zero-bordered NHWC bf16 maps == nn.Conv2d
    (+bias, +residual, ReLU), including the 1x1 / padding=1 conv of FTB (v2r:18) and the stride-2 downsample (v2r:176)."""
    k, stride, pad = geom
    n, H, W, C, b = 3, 9, 14, 64, 1
    g = _g(k * 7 + stride)
    x = torch.randn(n, C, H, W, generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, C, k, k, generator=g) / (C * k * k) ** 0.5).to(torch.bfloat16)
    bias = torch.randn(cout, generator=g)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(n, cout, Ho, Wo, generator=g)
    ref = F.relu(F.conv2d(x.float(), w.float(), bias, stride, pad) + res)
    Hp, Wp = H + 2 * b, W + 2 * b
    xp = torch.zeros(n, Hp, Wp, C, dtype=torch.bfloat16)
    xp[:, b:-b, b:-b] = x.permute(0, 2, 3, 1)
    ni, yo, xo = torch.arange(n)[:, None, None], torch.arange(Ho)[None, :, None], torch.arange(Wo)[None, None, :]
    a_row = (((ni * Hp + yo * stride - pad + b) * Wp + xo * stride - pad + b) * C).reshape(-1).int()
    k0 = torch.arange(0, k * k * C, 64)
    a_k = (((k0 // C) // k * Wp + (k0 // C) % k) * C + k0 % C).int()
    ob = 1
    o_row = ((((ni * (Ho + 2 * ob) + yo + ob) * (Wo + 2 * ob) + xo + ob) * cout).reshape(-1)).int()
    rp = torch.zeros(n, Ho + 2, Wo + 2, cout)
    rp[:, 1:-1, 1:-1] = res.permute(0, 2, 3, 1)
    out = torch.full((n, Ho + 2, Wo + 2, cout), 5.0, device=DEV)
    sh = torch.full((n, Ho + 2, Wo + 2, cout), 5.0, device=DEV, dtype=torch.bfloat16)
    wk = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    K = k * k * C
    L.gemm([(xp.to(DEV), K, K)], wk.to(DEV), out, M=n * Ho * Wo, N=cout, compute=L.BF16, epilogue=L.EPI_RESID, bias=bias.to(DEV),
           resid=rp.to(DEV), relu=True, ldo=cout, ldr=cout, out_bf16=sh, ld_out_bf16=cout,
           a_row_offset=a_row.to(DEV), a_ktile_offset=a_k.to(DEV), out_row_offset=o_row.to(DEV))
    got = out.cpu()
    torch.testing.assert_close(got[:, 1:-1, 1:-1].permute(0, 3, 1, 2), ref, atol=2e-3, rtol=2e-3)
    assert torch.all(got[:, 0] == 5) and torch.all(got[:, :, 0] == 5) and torch.all(got[:, -1] == 5) and torch.all(got[:, :, -1] == 5)
    assert torch.equal(sh.cpu()[:, 1:-1, 1:-1], got[:, 1:-1, 1:-1].to(torch.bfloat16))      # border untouched, shadow = rounded out


def test_gemm_offset_tables_need_the_bf16_path(L):
    a = torch.zeros(64, 64, device=DEV)
    w = torch.zeros(64, 64, device=DEV)
    out = torch.zeros(64, 64, device=DEV)
    t = torch.zeros(64, dtype=torch.int32, device=DEV)
    with pytest.raises(L.V2AError, match="offset tables"):
        L.gemm([(a, 64, 64)], w, out, M=64, N=64, compute=L.F32, a_row_offset=t, a_ktile_offset=t)


def test_pool2d_bordered_maps_and_bf16_shadow(L):
    B, H, W, C = 2, 9, 15, 64
    x = torch.randn(B, C, H, W, generator=_g(9))
    xp = torch.full((B, H + 2, W + 2, C), 99.0)          # a non-zero border must never be read (max pool pads with -inf)
    xp[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1)
    ref = F.max_pool2d(x, 3, 2, 1)
    Ho, Wo = ref.shape[2:]
    out = torch.zeros(B, Ho + 2, Wo + 2, C, device=DEV)
    sh = torch.zeros(B, Ho + 2, Wo + 2, C, device=DEV, dtype=torch.bfloat16)
    L.pool2d(xp.to(DEV), out, B=B, H=H, W=W, C_=C, k=3, stride=2, pad=1, mode=0, Ho=Ho, Wo=Wo, out_bf16=sh, in_border=1, out_border=1)
    got = out.cpu()
    assert torch.equal(got[:, 1:-1, 1:-1].permute(0, 3, 1, 2), ref)
    assert torch.all(got[:, 0] == 0) and torch.all(got[:, :, -1] == 0)
    assert torch.equal(sh.cpu()[:, 1:-1, 1:-1], got[:, 1:-1, 1:-1].to(torch.bfloat16))


# ------------------------------------------------------------------------------- fused head
def test_roll_head_matches_reference_tail(L, params, engines):
    """v2r:224-249 on random pyramid maps: FRB gates, spatial softmax, conv2 + pool + fc (pool / 1x1-conv commuted)."""
    n, Hh, Ww = 3, 4, 29
    g = _g(11)
    x2_, x3_, x4_ = (torch.randn(n, 128, Hh, Ww, generator=g) for _ in range(3))
    x5 = torch.randn(n, 64, Hh, Ww, generator=g).clamp_min(0)
    P = params
    p4 = VO.frb(P, "FRB4", x4_, x5)
    p3 = VO.frb(P, "FRB3", x3_, p4)
    p2 = VO.frb(P, "FRB2", x2_, p3)
    out1 = p2 * p3
    a = F.softmax(out1.flatten(2), dim=2).view_as(out1)
    o = F.conv2d(a * p4, P["conv2.weight"], P["conv2.bias"]) + p4
    ref = F.linear(o.mean((2, 3)), P["fc.weight"], P["fc.bias"])
    eng = engines["fp32"]
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)
    bufs = [nhwc(t) for t in (x2_, x3_, x4_, x5)]
    for sig in (0, 1):
        args = L.RollHeadArgs()
        args.x2, args.x3, args.x4, args.x5 = (b.data_ptr() for b in bufs)
        args.B, args.P = n, Hh * Ww
        for k, v in eng.head_w.items():
            setattr(args, k, v.data_ptr())
        out = torch.empty(n, 51, device=DEV)
        args.notes, args.apply_sigmoid, args.out = 51, sig, out.data_ptr()
        L.roll_head(args)
        torch.testing.assert_close(out.cpu(), torch.sigmoid(ref) if sig else ref, atol=2e-5, rtol=2e-5)


# ------------------------------------------------------------------------------- whole network vs the reference vectors
def _windows_0_3_6():
    from v2a_amd.synth import synthetic_piano_frames
    return VO.frame_windows(synthetic_piano_frames(1, 7, seed=INPUT_SEED))[[0, 3, 6]]


def test_forward_fp32_matches_reference_vectors(engines):
    g = np.load(os.path.join(GOLD, "video2roll_forward.npz"))
    taps = {}
    logits = engines["fp32"].forward_windows(_windows_0_3_6(), taps).cpu().numpy()
    err = np.abs(logits - g["logits"]).max()
    print(f"\nvideo2roll fp32 vs reference logits: max |d| = {err:.3e} (|logit| max {np.abs(g['logits']).max():.1f})")
    assert err < 2e-3
    for k in ("x1", "x2", "x3", "x4", "x5", "x2_", "x3_", "x4_"):
        a = torch.cat(taps[k], 0).cpu().numpy()
        assert tuple(g[f"{k}_shape"]) == a.shape
        np.testing.assert_allclose(a[tuple(g[f"{k}_idx"].T)], g[f"{k}_val"], rtol=1e-4, atol=1e-4)
        assert np.abs(a).mean(dtype=np.float64) == pytest.approx(g[f"{k}_stats"][1], rel=1e-4)


def test_forward_bf16_close_to_reference_vectors(engines):
    """bf16 operands, fp32 accumulation through 21 conv layers: logits of magnitude ~17 within 0.35 abs / 3 % of range."""
    g = np.load(os.path.join(GOLD, "video2roll_forward.npz"))
    logits = engines["bf16"].forward_windows(_windows_0_3_6()).cpu().numpy()
    err = np.abs(logits - g["logits"])
    print(f"\nvideo2roll bf16 vs reference logits: max |d| = {err.max():.3e}, mean {err.mean():.3e}")
    assert err.max() < 0.35 and err.mean() < 0.08


@pytest.mark.parametrize("l", [10, 14])
def test_encode_frames_fp32_matches_reference_lines(engines, l):
    from v2a_amd.synth import synthetic_piano_frames
    g = np.load(os.path.join(GOLD, "video2roll_encode.npz"))
    x = synthetic_piano_frames(2, 4, seed=INPUT_SEED + 1)
    roll = engines["fp32"].encode_frames(x, l)
    assert roll.shape == (2, l, 51) and roll.dtype == torch.float32 and roll.is_cuda
    np.testing.assert_allclose(roll.cpu().numpy(), g[f"roll_l{l}"], rtol=0, atol=1e-4)
    if l == 14:
        assert torch.all(roll[:, 12:] == 0)


def test_encode_frames_bf16_and_chunking(params):
    """bf16 probabilities within 0.03 of the fp32 restatement; the chunk size never changes a result bit."""
    from v2a_amd.synth import synthetic_piano_frames
    from v2a_amd.video2roll import Video2RollEngine
    x = synthetic_piano_frames(1, 9, seed=5)
    with torch.no_grad():
        ref = VO.encode_frames(params, x, 30)
    a = Video2RollEngine(params, DEV, compute="bf16", chunk=4).encode_frames(x, 30)
    b = Video2RollEngine(params, DEV, compute="bf16", chunk=9).encode_frames(x, 30)
    assert torch.equal(a, b)
    err = (a.cpu() - ref).abs()
    print(f"\nvideo2roll bf16 encode_frames vs CPU restatement: max {err.max():.3e} mean {err.mean():.3e}")
    assert err.max() < 0.03 and err.mean() < 0.004
    assert torch.all(a[:, 27:] == 0)


def test_engine_rejects_incomplete_state_dict(params):
    from v2a_amd.video2roll import Video2RollEngine
    sd = dict(params)
    sd.pop("FTB3.conv1.weight")
    with pytest.raises(KeyError, match="FTB3.conv1.weight"):
        Video2RollEngine(sd, DEV)
    sd = dict(params)
    sd["fc.weight"] = torch.zeros(51, 64)
    with pytest.raises(ValueError, match="fc.weight"):
        Video2RollEngine(sd, DEV)


# ------------------------------------------------------------------------------- E2TTS.sample(frames=...) end to end (V2P)
def test_sample_takes_frames_through_the_hip_encoder(small, params):
    """V2P path of x3:2164-2176: `frames` -> encode_frames (HIP Video2Roll) -> roll conditioning of the sampler,
    against the CPU restatements of both stages chained."""
    from conftest import make_model
    from oracle import e2_cfm_oracle as O
    from v2a_amd.synth import synthetic_piano_frames
    cfg, P, i = small["cfg"], small["P"], small["inp"]
    m = make_model(cfg, P, "fp32")
    x = synthetic_piano_frames(2, 14, seed=3)                          # floor(40 / 3) + 1 frames, x3:1913
    with pytest.raises(NotImplementedError, match="video2roll_net"):
        m.sample(torch.zeros(2, 40, 16), y0=i["y0"], frames=x, text_embed=i["text"], context=i["ctx"], context_mask=i["ctx_mask"], steps=4)
    res = m.load_state_dict({**P, **{"video2roll_net." + k: v for k, v in params.items()}}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert "video2roll_net.fc.weight" in m.state_dict()
    kw = dict(steps=4, cfg_strength=2.0, remove_parallel_component=False, sway_sampling=True)
    y = m.sample(torch.zeros(2, 40, 16), y0=i["y0"], frames=x, text_embed=i["text"], context=i["ctx"], context_mask=i["ctx_mask"], **kw)
    with torch.no_grad():
        roll = VO.encode_frames(params, x, 40)
        ref = O.sample(P, cfg, i["y0"], i["text"], roll, i["ctx"], i["ctx_mask"], **kw)
    err = float((y.cpu() - ref).abs().max())
    print(f"\nV2P sample(frames=) fp32 vs CPU restatement: max |d| = {err:.3e}")
    assert err < 1e-3
    np.testing.assert_allclose(m.encode_frames(x, 40).cpu().numpy(), roll.numpy(), atol=1e-4)


def test_inconsistent_pyramid_is_refused(engines):
    """A frame size whose FTB maps do not line up with layer4's (the reference would fail inside FRB's torch.cat): raised
    loudly before the head runs."""
    from v2a_amd import _lib
    x = torch.rand(1, 5, 90, 300, generator=_g(1))
    with pytest.raises(_lib.V2AError, match="pyramid maps disagree"):
        engines["bf16"].forward_windows(x)


def test_other_consistent_frame_size(params, engines):
    """The network is fully convolutional: a smaller frame whose pyramid lines up (40 x 132) matches the restatement too."""
    x = torch.rand(2, 5, 40, 132, generator=_g(2))
    with torch.no_grad():
        ref = VO.resnet_forward(params, x)
    got = engines["fp32"].forward_windows(x).cpu()
    assert float((got - ref).abs().max()) < 2e-3
    gb = engines["bf16"].forward_windows(x).cpu()
    assert float((gb - ref).abs().max()) < 0.5


def test_cli_piano_to_waveform(tmp_path, params):
    """The whole V2P chain from the command line (N4 + N2 + hot path + N1): scp list, cached CLIP / T5 / piano-frame files,
    a reference-layout checkpoint that holds `video2roll_net.*`, Encodec weights -> latents AND 24 kHz wav files, equal to
    composing the three stages by hand."""
    import json
    from scipy.io import wavfile
    import v2a_amd
    from v2a_amd import cli
    from v2a_amd.synth import random_encodec_decoder_state_dict, random_state_dict, synthetic_piano_frames
    mc = dict(dim=128, dim_text=192, dim_frames=64, depth=2, heads=2, dim_head=64, frames_heads=1, num_registers=4, max_seq_len=256,
              num_channels=128)
    cfg = v2a_amd.DiTConfig(**mc)
    sd = random_state_dict(cfg, seed=3)
    ck = tmp_path / "v2p.pt"
    torch.save({"model_state_dict": {**sd, **{"video2roll_net." + k: v for k, v in params.items()}}}, ck)
    esd = random_encodec_decoder_state_dict(5)
    torch.save({"decoder." + k: v for k, v in esd.items()}, tmp_path / "encodec.pt")
    vids = [str(tmp_path / f"piano{i}.mp4") for i in range(2)]
    (tmp_path / "list.scp").write_text("".join(f"{v}\tpiano {i}\n" for i, v in enumerate(vids)))
    g = torch.Generator().manual_seed(8)
    for i, v in enumerate(vids):
        dur = 0.5 + 0.02 * i
        v2a_amd.save_clip_cache(v2a_amd.feature_cache_path(v), torch.randn(12 + i, cfg.dim_text, generator=g), dur)
        np.savez(v.replace(".mp4", ".t5.npz"), (0.2 * torch.randn(4, cfg.dim, generator=g)).numpy())
        v2a_amd.save_piano_frames_cache(v2a_amd.piano_frames_cache_path(v), synthetic_piano_frames(1, 12 + i, seed=i)[0, 0][..., None], dur)
    out = tmp_path / "out"
    written = cli.main([str(ck), "0", str(tmp_path / "list.scp"), "0", "2", str(out), "--batch", "2", "--steps", "3", "--frames", "40",
                        "--dtype", "fp32", "--model-config", json.dumps(mc), "--piano", "--encodec", str(tmp_path / "encodec.pt")])
    assert len(written) == 2
    for i, v in enumerate(vids):
        n = int((0.5 + 0.02 * i) * 24000) // 320
        lat = np.load(out / f"piano{i}.latent.npy")
        rate, wav = wavfile.read(out / f"piano{i}.wav")
        assert rate == 24000 and wav.dtype == np.float32 and wav.shape == (n * 320,) and np.isfinite(wav).all()
        ref = v2a_amd.EncodecDecoder(esd, DEV).decode(torch.from_numpy(lat[:n]).t()[None])[0].cpu().numpy()
        np.testing.assert_allclose(wav, ref, atol=1e-5)
    # the roll really conditioned the sample: without --piano the latents differ
    plain = cli.main([str(ck), "0", str(tmp_path / "list.scp"), "0", "2", str(tmp_path / "out2"), "--batch", "2", "--steps", "3",
                      "--frames", "40", "--dtype", "fp32", "--model-config", json.dumps(mc)])
    assert np.abs(np.load(plain[0]) - np.load(written[0])).max() > 1e-3
