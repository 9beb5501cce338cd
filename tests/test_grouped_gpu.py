"""Grouped launches (ABI 7: v2a_gemm_grouped, v2a_attention_grouped, v2a_dwconv_grouped): the same op of the audio, text and frames blocks of
a layer (x3:1081-1137: A_i, T_i+1, F_i+1 are independent) behind ONE kernel launch.  Every problem of a group must come out exactly as its
own single launch does -- same K order per output element, same epilogue expressions -- so each test runs the problems separately and
grouped on the sampler's shapes and compares bit for bit; the separate launches are the ones the oracle-parity tests of
tests/test_kernels_gpu.py cover."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def L():
    from v2a_amd import _lib
    _lib.lib()
    return _lib


def _g(seed=0):
    return torch.Generator().manual_seed(seed)


def _bf(*shape, seed, scale=1.0):
    return (torch.randn(*shape, generator=_g(seed)) * scale).to(torch.bfloat16).to(DEV)


def _f32(*shape, seed, scale=1.0):
    return (torch.randn(*shape, generator=_g(seed)) * scale).to(DEV)


# stream widths of the shipped model: (d, heads), K of the feed-forward-out GEMM = 4 d
STREAMS = [(1024, 16), (1280, 16), (512, 8)]


def _resid_family(L, M, Ns, segs, seed):
    """problem 0: GATE_RESID with a step-indexed gate, a switched folded-norm producer and a bf16 shadow; problem 1: RESID with a plain
    producer; problem 2: fp32 STORE (no residual, no shadow).  Every problem has its own number of K segments."""
    step = torch.tensor([1], dtype=torch.int32, device=DEV)
    probs = []
    for j, (N, ks) in enumerate(zip(Ns, segs)):
        a = [(_bf(M, k, seed=seed + 10 * j + i, scale=0.5), k, k) for i, k in enumerate(ks)]
        K = sum(ks)
        w = _bf(N, K, seed=seed + 10 * j + 5, scale=1 / math.sqrt(K))
        bias = _f32(N, seed=seed + 10 * j + 6)
        resid = _f32(M, N, seed=seed + 10 * j + 7)
        kw = dict(M=M, N=N, compute=L.BF16, bias=bias)
        if j == 0:
            gate = torch.rand(3, 2, N, generator=_g(seed + 8)).to(DEV)
            gam = (1 + 0.1 * torch.randn(3, 2 * N, generator=_g(seed + 9))).to(DEV)
            kw.update(epilogue=L.EPI_GATE_RESID, resid=resid, gate=gate[0, 0], step=step, gate_step_stride=2 * N, rows_per_batch=(M + 1) // 2,
                      norm_gamma=gam[0], norm_step_stride=2 * N, norm_switch_row=M // 2, norm_switch_offset=N)
        elif j == 1:
            gam = (1 + 0.1 * torch.randn(N, generator=_g(seed + 19))).to(DEV)
            kw.update(epilogue=L.EPI_RESID, resid=resid, norm_gamma=gam)
        probs.append((a, w, kw, j < 2))
    return probs


def _run(L, probs, out_dtype, grouped, tile_hint, geglu=False):
    outs, built = [], []
    for a, w, kw, shadow in probs:
        M, N = kw["M"], kw["N"]
        out = torch.full((M, N // 2 if geglu else N), float("nan"), dtype=out_dtype, device=DEV)
        if kw.get("resid") is not None:              # the residual epilogues run in place in the sampler
            out.copy_(kw["resid"])
            kw = dict(kw, resid=out)
        extra = {}
        sh = ssq = None
        if shadow:
            sh = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            ssq = torch.zeros(M, (N // 32 + 3) // 4 * 4, device=DEV)
            extra = dict(out_bf16=sh, norm_ssq=ssq)
        outs.append((out, sh, ssq))
        if grouped:
            built.append(L.gemm_args(a, w, out, **kw, **extra))
        else:
            L.gemm(a, w, out, tile_hint=tile_hint, **kw, **extra)
    if grouped:
        L.gemm_grouped(built, tile_hint=tile_hint)
    torch.cuda.synchronize()
    return outs


def _same(x, y):
    for (o1, s1, q1), (o2, s2, q2) in zip(x, y):
        assert torch.isfinite(o1.float()).all()
        assert torch.equal(o1, o2), float((o1.float() - o2.float()).abs().max())
        if s1 is not None:
            assert torch.equal(s1, s2) and torch.equal(q1, q2)


@pytest.mark.parametrize("tile_hint", [13, 15, 2, 4, 1, 16, 7])
@pytest.mark.parametrize("M", [1564, 300])
def test_gemm_grouped_residual_family(L, M, tile_hint):
    """out-projections / feed-forward-out / cross-condition groups: GATE_RESID + RESID + fp32 STORE in one launch, on every tile shape the
    grouped entry takes (13 = 128x128 / 8 waves ... 7 = the 256x256 8-phase kernel)."""
    probs = _resid_family(L, M, [1024, 1280, 512], [(1024,), (1280, 1024), (512, 1024, 512)], seed=100 + M)
    _same(_run(L, probs, torch.float32, False, tile_hint), _run(L, probs, torch.float32, True, tile_hint))


@pytest.mark.parametrize("tile_hint", [7, 13, 1, 2])
def test_gemm_grouped_qkv_store_rope_and_folded_norm_consumer(L, tile_hint):
    """[q | k | v | gate] projections of the three streams: bf16 STORE with RoPE in the epilogue and the folded RMSNorm's row scale."""
    M, rpb = 1564, 782
    ang = torch.arange(rpb + 4).float()[:, None] * (1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64)))[None]
    tab = torch.stack((ang.cos(), ang.sin()), -1).contiguous().to(DEV)
    probs = []
    for j, (d, H) in enumerate(STREAMS):
        N = 3 * H * 64 + 16
        a = [(_bf(M, d, seed=300 + j, scale=0.5), d, d)]
        w = _bf(N, d, seed=310 + j, scale=1 / math.sqrt(d))
        ssq = torch.zeros(M, 40, device=DEV)
        ssq[:, :d // 32] = (torch.rand(M, d // 32, generator=_g(320 + j)) * 30 + 5).to(DEV)
        probs.append((a, w, dict(M=M, N=N, compute=L.BF16, bias=_f32(N, seed=330 + j), rope_table=tab, rope_cols=2 * H * 64, rope_pos_offset=2,
                                 rows_per_batch=rpb, row_ssq=ssq, row_norm_dim=d), False))
    _same(_run(L, probs, torch.bfloat16, False, tile_hint), _run(L, probs, torch.bfloat16, True, tile_hint))


@pytest.mark.parametrize("tile_hint", [7, 13, 1, 0])
def test_gemm_grouped_geglu(L, tile_hint):
    """feed-forward-in of the three streams: GEGLU epilogue, N = 8 d (value / gate rows packed), three different K."""
    M = 1564 if tile_hint in (7, 0) else 400
    probs = []
    for j, (d, _) in enumerate(STREAMS):
        N = 8 * d
        a = [(_bf(M, d, seed=400 + j, scale=0.5), d, d)]
        probs.append((a, _bf(N, d, seed=410 + j, scale=1 / math.sqrt(d)), dict(M=M, N=N, compute=L.BF16, epilogue=L.EPI_GEGLU, bias=_f32(N, seed=420 + j)), False))
    _same(_run(L, probs, torch.bfloat16, False, tile_hint, geglu=True), _run(L, probs, torch.bfloat16, True, tile_hint, geglu=True))


def test_gemm_grouped_two_problems_and_rejects(L):
    probs = _resid_family(L, 200, [256, 128], [(128,), (64, 64)], seed=7)
    _same(_run(L, probs, torch.float32, False, 13), _run(L, probs, torch.float32, True, 13))
    a = [(_bf(64, 64, seed=1), 64, 64)]
    w = _bf(64, 64, seed=2)
    o32, o16 = torch.zeros(64, 64, device=DEV), torch.zeros(64, 64, dtype=torch.bfloat16, device=DEV)
    mixed = [L.gemm_args(a, w, o32, M=64, N=64, compute=L.BF16), L.gemm_args(a, w, o16, M=64, N=64, compute=L.BF16)]
    with pytest.raises(L.V2AError):
        L.gemm_grouped(mixed)                       # an fp32 and a bf16 result do not share an epilogue instantiation
    f32a = [(torch.zeros(64, 64, device=DEV), 64, 64)]
    with pytest.raises(L.V2AError):
        L.gemm_grouped([L.gemm_args(f32a, w, o32, M=64, N=64, compute=L.BF16)] * 2)      # fp32 A operands
    with pytest.raises(L.V2AError):
        L.gemm_grouped([L.gemm_args(a, w, o16, M=64, N=64, compute=L.BF16)] * 2, tile_hint=9)


@pytest.mark.parametrize("B,N,kv_len", [(2, 782, None), (2, 782, [782, 611]), (3, 100, [100, 64, 1])])
@pytest.mark.parametrize("clamp", [50.0, 0.0])
def test_attention_grouped_equals_separate(L, B, N, kv_len, clamp):
    """self-attention of the audio (16 heads), text (16) and frames (8) blocks inside their fused [q | k | v | gate] buffers."""
    lens = None if kv_len is None else torch.tensor(kv_len, dtype=torch.int32, device=DEV)
    bufs = []
    for j, (_, H) in enumerate(STREAMS):
        inner = H * 64
        npad = 3 * inner + 16
        qkv = _bf(B * N, npad, seed=500 + j)
        bufs.append((qkv, inner, npad, H))

    def call(grouped):
        outs, built = [], []
        for qkv, inner, npad, H in bufs:
            ao = torch.full((B * N, inner), float("nan"), dtype=torch.bfloat16, device=DEV)
            base = qkv.data_ptr()
            kw = dict(strides=(npad, npad, npad, npad, inner, N * npad, N * npad, N * npad, N * npad, N * inner), B=B, H=H, Nq=N, Nk=N, kv_len=lens,
                      q_len=lens, scale=0.125, softclamp=clamp, dtype=L.BF16)
            args = (base, base + inner * 2, base + 2 * inner * 2, base + 3 * inner * 2, ao.data_ptr())
            if grouped:
                built.append(L.attention_args(*args, **kw))
            else:
                L.attention(*args, **kw)
            outs.append(ao)
        if grouped:
            L.attention_grouped(built)
        torch.cuda.synchronize()
        return outs
    for a, b in zip(call(False), call(True)):
        assert torch.isfinite(a.float()).all() and torch.equal(a, b)


@pytest.mark.parametrize("norm", [True, False])
@pytest.mark.parametrize("B,N,lens", [(2, 782, None), (2, 782, [782, 500]), (4, 97, [97, 1, 50, 96])])
def test_dwconv_grouped_equals_separate(L, B, N, lens, norm):
    ld = None if lens is None else torch.tensor(lens, dtype=torch.int32, device=DEV)
    step = torch.tensor([1], dtype=torch.int32, device=DEV)
    probs = []
    for j, (d, _) in enumerate(STREAMS):
        q = dict(x=_f32(B, N, d, seed=600 + j), wt=_f32(31, d, seed=610 + j, scale=1 / math.sqrt(31)), bias=_f32(d, seed=620 + j, scale=0.1), d=d)
        if norm:
            gam = (1 + 0.1 * torch.randn(2, d, generator=_g(630 + j))).to(DEV)
            q["gamma"], q["stepped"] = gam, j == 0
        probs.append(q)

    def call(grouped):
        outs, items = [], []
        for q in probs:
            d = q["d"]
            out = torch.full((B, N, d), float("nan"), device=DEV)
            nd = None
            sh = ssq = None
            if norm:
                sh = torch.zeros(B * N, d, dtype=torch.bfloat16, device=DEV)
                ssq = torch.zeros(B * N, (d // 32 + 3) // 4 * 4, device=DEV)
                nd = dict(out_bf16=sh, ld_out_bf16=d, gamma=q["gamma"][0], ssq=ssq)
                if q["stepped"]:
                    nd.update(step=step, step_stride=d)
            if grouped:
                items.append(dict(x=q["x"], out=out, wt=q["wt"], bias=q["bias"], d=d, norm=nd))
            else:
                L.dwconv(q["x"], out, q["wt"], q["bias"], B=B, N=N, d=d, ksize=31, lens=ld, norm=nd)
            outs.append((out, sh, ssq))
        if grouped:
            L.dwconv_grouped(items, B=B, N=N, ksize=31, lens=ld)
        torch.cuda.synchronize()
        return outs
    _same(call(False), call(True))


# ------------------------------------------------------------------------------------------- bf16x3 mode (split operands)
def _planes(x32):
    hi = x32.to(torch.bfloat16)
    lo = (x32 - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo], -1).contiguous()


@pytest.mark.parametrize("tile_hint", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("family", ["resid", "store_rope", "geglu"])
def test_gemm_grouped_split_operands(L, family, tile_hint):
    """The bf16x3 mode's groups: A rows and W rows as hi | lo planes, three MFMA products per fp32 product; STORE to fp32 with RoPE (the
    [q | k | v | gate] rows stay fp32 there), GEGLU to hi | lo planes, the fp32 residual family with split shadows.  The tile shape is
    named (1..4 = the split ring tiles, 5 = the 8-phase kernel on three K segments): the ring sums hi*lo + lo*hi + hi*hi per K step, the
    8-phase form sums the three products over all of K one after another -- same value up to fp32 rounding, not bit for bit, so a group
    is compared with single launches of the SAME form."""
    if family == "resid" and tile_hint == 5:
        pytest.skip("the 8-phase form takes one logical K segment per problem")
    M, rpb = 782, 391
    ang = torch.arange(rpb + 4).float()[:, None] * (1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64)))[None]
    tab = torch.stack((ang.cos(), ang.sin()), -1).contiguous().to(DEV)
    step = torch.tensor([1], dtype=torch.int32, device=DEV)
    probs = []
    for j, (d, H) in enumerate(STREAMS):
        g = _g(700 + j)
        if family == "resid":
            ks = [(d,), (d, 512), (256, d)][j]
            N = d
        elif family == "store_rope":
            ks, N = (d,), 3 * H * 64 + 16
        else:
            ks, N = (d,), 8 * d if tile_hint in (5, 3) else 2 * d
        a = [(_planes(torch.randn(M, k, generator=g) * 0.5).to(DEV), 2 * k, k) for k in ks]
        K = sum(ks)
        w = _planes(torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
        kw = dict(M=M, N=N, compute=L.BF16, a_split=True, bias=_f32(N, seed=720 + j))
        if family == "resid":
            resid = _f32(M, N, seed=730 + j)
            if j == 0:
                gate = torch.rand(3, N, generator=g).to(DEV)
                kw.update(epilogue=L.EPI_GATE_RESID, resid=resid, gate=gate[0], step=step, gate_step_stride=N, rows_per_batch=rpb)
            elif j == 1:
                kw.update(epilogue=L.EPI_RESID, resid=resid)
        elif family == "store_rope":
            kw.update(rope_table=tab, rope_cols=2 * H * 64, rope_pos_offset=1, rows_per_batch=rpb)
        else:
            kw.update(epilogue=L.EPI_GEGLU, out_split=True)
        probs.append((a, w, kw, family == "resid" and j < 2))

    def call(grouped):
        outs, built = [], []
        for a, w, kw, shadow in probs:
            N = kw["N"]
            if family == "geglu":
                out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)            # hi | lo planes of the N / 2 hidden values
                extra = dict(ldo=N)
            else:
                out = torch.full((M, N), float("nan"), device=DEV)
                extra = {}
            if kw.get("resid") is not None:
                out.copy_(kw["resid"])
                kw = dict(kw, resid=out)
            sh = None
            if shadow:
                sh = torch.zeros(M, 2 * N, dtype=torch.bfloat16, device=DEV)
                extra.update(out_bf16=sh, ld_out_bf16=2 * N, out_bf16_split=True)
            outs.append((out, sh, None))
            if grouped:
                built.append(L.gemm_args(a, w, out, **kw, **extra))
            else:
                L.gemm(a, w, out, tile_hint=tile_hint, **kw, **extra)
        if grouped:
            L.gemm_grouped(built, tile_hint=tile_hint)
        torch.cuda.synchronize()
        return outs
    x, y = call(False), call(True)
    for (o1, s1, _), (o2, s2, _) in zip(x, y):
        assert torch.isfinite(o1.float()).all() and torch.equal(o1, o2), float((o1.float() - o2.float()).abs().max())
        if s1 is not None:
            assert torch.equal(s1, s2)


@pytest.mark.parametrize("B,N,kv_len", [(2, 782, [782, 611]), (3, 100, [100, 64, 1])])
def test_attention_grouped_split_equals_separate(L, B, N, kv_len):
    """bf16x3 mode: fp32 q | k | v | gate buffers, split-bf16 MFMA products, hi | lo output planes."""
    lens = torch.tensor(kv_len, dtype=torch.int32, device=DEV)
    bufs = [(_f32(B * N, 3 * H * 64 + 16, seed=800 + j), H * 64, 3 * H * 64 + 16, H) for j, (_, H) in enumerate(STREAMS)]

    def call(grouped):
        outs, built = [], []
        for qkv, inner, npad, H in bufs:
            ao = torch.zeros(B * N, 2 * inner, dtype=torch.bfloat16, device=DEV)
            base = qkv.data_ptr()
            kw = dict(strides=(npad, npad, npad, npad, 2 * inner, N * npad, N * npad, N * npad, N * npad, N * 2 * inner), B=B, H=H, Nq=N, Nk=N,
                      kv_len=lens, q_len=lens, scale=0.125, softclamp=50.0, dtype=L.BF16_SPLIT, out_split=True)
            args = (base, base + inner * 4, base + 2 * inner * 4, base + 3 * inner * 4, ao.data_ptr())
            if grouped:
                built.append(L.attention_args(*args, **kw))
            else:
                L.attention(*args, **kw)
            outs.append(ao)
        if grouped:
            L.attention_grouped(built)
        torch.cuda.synchronize()
        return outs
    for a, b in zip(call(False), call(True)):
        assert torch.isfinite(a.float()).all() and float(a.float().abs().max()) > 0 and torch.equal(a, b)
