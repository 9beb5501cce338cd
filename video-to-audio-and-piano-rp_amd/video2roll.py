"""Video2Roll frame encoder on the MI355X kernels.

Mirrors `E2TTS.encode_frames` (src/e2_tts_pytorch/e2_tts_crossatt3.py:1525-1553) and the network it runs,
`Video2RollNet.resnet18(num_classes=51)` (src/audeo/Video2RollNet.py:127-251, `v2r` below): 5-frame windows of
100x900 grey frames -> 51 key probabilities per frame, repeated x3 in time and cropped / padded to the latent length.

Design (not a translation of the reference's NCHW module tree):
  * activations NHWC fp32; every convolution is `v2a_im2col` (patch matrix in the compute dtype, the 5-frame window
    of the first layer gathered on the fly) + `v2a_gemm` -- the same MFMA GEMM the sampler uses -- with the eval-mode
    BatchNorm folded into weight / bias at load time and ReLU / residual add in the GEMM epilogue;
  * FRB gates, spatial softmax, the last 1x1 conv, global pooling, fc and sigmoid are one fused kernel per window
    (`v2a_roll_head`; pooling commutes with the 1x1 conv, so only per-channel position sums are formed);
  * windows are processed in chunks so the patch matrix of the first layer (28 MB per window in bf16) stays bounded.
There is no CPU fallback: without libv2a_cfm.so every call raises.
"""
from __future__ import annotations

import math

import torch

from . import _lib as L

NOTES = 51
BN_EPS = 1e-5
_LAYERS = (("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2), ("layer4", 512, 2))


def expected_state_dict_shapes(num_classes: int = NOTES) -> dict[str, tuple]:
    """Key layout of `Video2RollNet.resnet18(num_classes=51).state_dict()` (v2r:127-168, 254-258)."""
    s: dict[str, tuple] = {}

    def bn(p, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            s[f"{p}.{k}"] = (c,)
        s[f"{p}.num_batches_tracked"] = ()

    s["conv1.weight"] = (64, 5, 11, 11)
    bn("bn1", 64)
    inpl = 64
    for name, planes, stride in _LAYERS:
        for b in range(2):
            p = f"{name}.{b}"
            s[f"{p}.conv1.weight"] = (planes, inpl, 3, 3)
            bn(f"{p}.bn1", planes)
            s[f"{p}.conv2.weight"] = (planes, planes, 3, 3)
            bn(f"{p}.bn2", planes)
            if b == 0 and (stride != 1 or inpl != planes):
                s[f"{p}.downsample.0.weight"] = (planes, inpl, 1, 1)
                bn(f"{p}.downsample.1", planes)
            inpl = planes
    for name, cin in (("FTB2_1", 128), ("FTB2_2", 128), ("FTB3", 256), ("FTB4", 512)):
        s[f"{name}.conv0.weight"] = (128, cin, 1, 1)
        s[f"{name}.conv1.weight"] = (128, 128, 3, 3)
        bn(f"{name}.bn1", 128)
        s[f"{name}.conv2.weight"] = (128, 128, 3, 3)
    for name, c1 in (("FRB2", 128), ("FRB3", 128), ("FRB4", 64)):
        s[f"{name}.fc1.weight"] = (128, c1 + 128)
        s[f"{name}.fc1.bias"] = (128,)
        s[f"{name}.fc2.weight"] = (128, 128)
        s[f"{name}.fc2.bias"] = (128,)
    s["toplayer.weight"] = (64, 512, 1, 1)
    s["toplayer.bias"] = (64,)
    bn("toplayer_bn", 64)
    s["conv2.weight"] = (128, 128, 1, 1)
    s["conv2.bias"] = (128,)
    s["fc.weight"] = (num_classes, 128)
    s["fc.bias"] = (num_classes,)
    return s


class _Conv:
    """One convolution as a GEMM operand: weight [Cout][Kpad] in the compute dtype (BatchNorm scale folded in),
    fp32 bias (folded BatchNorm shift, plus the conv's own bias where it has one)."""

    def __init__(self, sd, wkey, bnkey, dev, cd, *, stride, pad, window=False, conv_bias=None):
        w = sd[wkey].float()
        co, ci, kh, kw = w.shape
        bias = sd[conv_bias].float() if conv_bias else None
        if bnkey is not None:
            s = sd[f"{bnkey}.weight"].float() / torch.sqrt(sd[f"{bnkey}.running_var"].float() + BN_EPS)
            shift = sd[f"{bnkey}.bias"].float() - sd[f"{bnkey}.running_mean"].float() * s
            w = w * s[:, None, None, None]
            bias = shift if bias is None else bias * s + shift
        # K order: (c, ky, kx) for the windowed first layer (weight's own order), (ky, kx, c) for NHWC sources
        wk = w.reshape(co, -1) if window else w.permute(0, 2, 3, 1).reshape(co, -1)
        K = wk.shape[1]
        kq = 64 if cd == torch.bfloat16 else 16
        self.Kpad = (K + kq - 1) // kq * kq
        wp = torch.zeros(co, self.Kpad)
        wp[:, :K] = wk
        self.w = wp.to(dev, cd).contiguous()
        self.w_pack = None
        if window and cd == torch.bfloat16:
            # implicit first layer (v2a_frames_pack): per input channel kh rows of 16 columns, padded to whole 64-element K tiles
            g = (kh + 3) // 4
            wq = torch.zeros(co, ci, g * 4, 16)
            wq[:, :, :kh, :kw] = w
            self.w_pack = wq.reshape(co, ci * g * 64).to(dev, cd).contiguous()
            self.g = g
        self.bias = None if bias is None else bias.to(dev).contiguous()
        self.co, self.ci, self.kh, self.kw, self.stride, self.pad, self.window = co, ci, kh, kw, stride, pad, window

    def out_hw(self, H, W):
        return (H + 2 * self.pad - self.kh) // self.stride + 1, (W + 2 * self.pad - self.kw) // self.stride + 1


class Video2RollEngine:
    """`video2roll_net` + `encode_frames` of the reference E2TTS on HIP kernels.

    sd: state dict of the reference module (`video2roll_net.` prefix already stripped, or pass prefix=).
    compute: "bf16" (bf16 MFMA operands, fp32 accumulate / activations) or "fp32" (parity mode, exact-fp32 MFMA).
    chunk: windows per pass.  Default 256 in bf16 mode (implicit GEMM: no patch matrix; ~22 MB of activation maps per window,
    and one pass per clip measured fastest: 8.8 ms per 251-frame clip vs 13.3 ms at 25) and 16 in fp32 mode (the explicit
    fp32 patch matrix of the first layer is 53 MB per window)."""

    def __init__(self, sd, device="cuda:0", compute="bf16", prefix="", chunk=None):
        L.lib()                                                   # fail loudly without the HIP library
        self.dev = torch.device(device)
        if compute not in ("bf16", "fp32"):
            raise ValueError(f"compute must be 'bf16' or 'fp32', got {compute!r}")
        self.cd = torch.bfloat16 if compute == "bf16" else torch.float32
        self.code = L.BF16 if compute == "bf16" else L.F32
        self.chunk = int(chunk) if chunk else (256 if compute == "bf16" else 16)
        sd = {k[len(prefix):]: v.detach().cpu() for k, v in sd.items() if k.startswith(prefix)}
        want = expected_state_dict_shapes(sd["fc.weight"].shape[0] if "fc.weight" in sd else NOTES)
        missing = [k for k in want if k not in sd and not k.endswith("num_batches_tracked")]
        if missing:
            raise KeyError(f"Video2RollEngine: state dict lacks {len(missing)} keys, e.g. {missing[:4]}")
        bad = [k for k, shp in want.items() if k in sd and not k.endswith("num_batches_tracked") and tuple(sd[k].shape) != tuple(shp)]
        if bad:
            raise ValueError(f"Video2RollEngine: shape mismatch for {bad[:4]}")
        self.notes = sd["fc.weight"].shape[0]
        dev, cd = self.dev, self.cd
        c = {}
        c["conv1"] = _Conv(sd, "conv1.weight", "bn1", dev, cd, stride=2, pad=4, window=True)
        for name, planes, stride in _LAYERS:
            for b in range(2):
                p = f"{name}.{b}"
                st = stride if b == 0 else 1
                c[f"{p}.conv1"] = _Conv(sd, f"{p}.conv1.weight", f"{p}.bn1", dev, cd, stride=st, pad=1)
                c[f"{p}.conv2"] = _Conv(sd, f"{p}.conv2.weight", f"{p}.bn2", dev, cd, stride=1, pad=1)
                if f"{p}.downsample.0.weight" in sd:
                    c[f"{p}.down"] = _Conv(sd, f"{p}.downsample.0.weight", f"{p}.downsample.1", dev, cd, stride=st, pad=0)
        for name in ("FTB2_1", "FTB2_2", "FTB3", "FTB4"):
            c[f"{name}.conv0"] = _Conv(sd, f"{name}.conv0.weight", None, dev, cd, stride=1, pad=1)
            c[f"{name}.conv1"] = _Conv(sd, f"{name}.conv1.weight", f"{name}.bn1", dev, cd, stride=1, pad=1)
            c[f"{name}.conv2"] = _Conv(sd, f"{name}.conv2.weight", None, dev, cd, stride=1, pad=1)
        c["toplayer"] = _Conv(sd, "toplayer.weight", "toplayer_bn", dev, cd, stride=1, pad=0, conv_bias="toplayer.bias")
        self.convs = c
        t32 = lambda k: sd[k].float().t().contiguous().to(dev)    # [in][out]
        f32 = lambda k: sd[k].float().contiguous().to(dev)
        self.head_w = {}
        for i in (4, 3, 2):
            self.head_w[f"frb{i}_w1t"] = t32(f"FRB{i}.fc1.weight")
            self.head_w[f"frb{i}_b1"] = f32(f"FRB{i}.fc1.bias")
            self.head_w[f"frb{i}_w2t"] = t32(f"FRB{i}.fc2.weight")
            self.head_w[f"frb{i}_b2"] = f32(f"FRB{i}.fc2.bias")
        self.head_w["conv2_wt"] = sd["conv2.weight"].float().reshape(128, 128).t().contiguous().to(dev)
        self.head_w["conv2_b"] = f32("conv2.bias")
        self.head_w["fc_wt"] = t32("fc.weight")
        self.head_w["fc_b"] = f32("fc.bias")
        self._maps: dict = {}
        self._tabs: dict = {}
        self._col = None
        # bf16: implicit GEMM (the MFMA kernel gathers patch rows from zero-bordered NHWC bf16 maps through offset tables);
        # fp32 parity mode: explicit patch matrix + the exact-fp32 GEMM.  The windowed first layer uses im2col in both.
        self.implicit = compute == "bf16"

    # ---- activation maps --------------------------------------------------------------------------
    class _Map:
        """NHWC fp32 activation (n, H + 2b, W + 2b, C) with a zero border of b pixels, plus its bf16 copy when a
        convolution reads it (the DMA GEMM moves raw bf16 bytes)."""
        __slots__ = ("f32", "b16", "n", "H", "W", "C", "border")

        def __init__(self, f32, b16, n, H, W, C, border):
            self.f32, self.b16, self.n, self.H, self.W, self.C, self.border = f32, b16, n, H, W, C, border

        def interior(self):
            b = self.border
            return self.f32 if b == 0 else self.f32[:, b:-b, b:-b, :]

    def _map(self, name, n, H, W, C, border=0, shadow=False):
        """Cached by full geometry: a bordered map relies on its border staying zero, so two geometries never share memory."""
        key = (name, n, H, W, C, border, shadow)
        m = self._maps.get(key)
        if m is None:
            shp = (n, H + 2 * border, W + 2 * border, C)
            f32 = torch.zeros(shp, device=self.dev, dtype=torch.float32)
            b16 = torch.zeros(shp, device=self.dev, dtype=torch.bfloat16) if shadow else None
            m = self._maps[key] = self._Map(f32, b16, n, H, W, C, border)
        return m

    def _colbuf(self, rows, kpad):
        n = rows * kpad
        if self._col is None or self._col.numel() < n:
            self._col = torch.empty(n, device=self.dev, dtype=self.cd)
        return self._col[:n].view(rows, kpad)

    def _tables(self, cv, src, Ho, Wo, out_border):
        """Offset tables of one convolution geometry for v2a_gemm's implicit-GEMM mode (elements, int32)."""
        key = (cv.kh, cv.kw, cv.stride, cv.pad, cv.ci, cv.co, src.n, src.H, src.W, src.border, out_border)
        t = self._tabs.get(key)
        if t is None:
            b, C = src.border, cv.ci
            assert b >= cv.pad, "the source map's zero border must cover the convolution's padding"
            Hp, Wp = src.H + 2 * b, src.W + 2 * b
            dev = self.dev
            ni = torch.arange(src.n, device=dev, dtype=torch.int64)[:, None, None]
            yo = torch.arange(Ho, device=dev, dtype=torch.int64)[None, :, None]
            xo = torch.arange(Wo, device=dev, dtype=torch.int64)[None, None, :]
            a_row = ((ni * Hp + yo * cv.stride - cv.pad + b) * Wp + xo * cv.stride - cv.pad + b) * C
            k0 = torch.arange(0, cv.kh * cv.kw * C, 64, device=dev, dtype=torch.int64)
            tap, c0 = k0 // C, k0 % C
            a_k = ((tap // cv.kw) * Wp + tap % cv.kw) * C + c0
            o_row = None
            if out_border:
                o_row = (((ni * (Ho + 2 * out_border) + yo + out_border) * (Wo + 2 * out_border) + xo + out_border) * cv.co)
                o_row = o_row.reshape(-1).to(torch.int32).contiguous()
            assert int(a_row.max()) + int(a_k.max()) + 64 <= src.n * Hp * Wp * C < 2 ** 31
            t = self._tabs[key] = (a_row.reshape(-1).to(torch.int32).contiguous(), a_k.to(torch.int32).contiguous(), o_row)
        return t

    # ---- one convolution (+bias, +residual, +ReLU in the GEMM epilogue) ---------------------------------
    def _conv(self, name, src, dst_name, *, relu=False, resid=None, border=None, shadow=None, window=None, dst=None):
        cv = self.convs[name]
        if cv.window:
            frames, patches, T, first, n, H, W = window
        else:
            n, H, W = src.n, src.H, src.W
        Ho, Wo = cv.out_hw(H, W)
        rows = n * Ho * Wo
        if border is None:
            border = 1 if self.implicit else 0
        if shadow is None:
            shadow = self.implicit
        if dst is None:
            dst = self._map(dst_name, n, Ho, Wo, cv.co, border, shadow)
        assert (dst.n, dst.H, dst.W, dst.C) == (n, Ho, Wo, cv.co)
        if resid is not None:
            assert (resid.n, resid.H, resid.W, resid.C, resid.border) == (dst.n, dst.H, dst.W, dst.C, dst.border)
        epi = L.EPI_RESID if resid is not None else L.EPI_STORE
        rs = None if resid is None else resid.f32
        if cv.window and patches is not None:
            # first layer over the packed column patches of this clip (v2a_frames_pack): windows [first, first + n) of the clip
            key = ("c1", T, first, n, H, W)
            t = self._tabs.get(key)
            if t is None:
                Hp = H + 2 * cv.pad
                dev = self.dev
                ni = torch.arange(first, first + n, device=dev, dtype=torch.int64)[:, None, None]
                yo = torch.arange(Ho, device=dev, dtype=torch.int64)[None, :, None]
                xo = torch.arange(Wo, device=dev, dtype=torch.int64)[None, None, :]
                a_row = ((ni * Wo + xo) * Hp + yo * cv.stride) * 16
                kt = torch.arange(cv.ci * cv.g, device=dev, dtype=torch.int64)
                a_k = (kt // cv.g) * (Wo * Hp * 16) + (kt % cv.g) * 64
                assert int(a_row.max()) + int(a_k.max()) + 64 <= patches.numel() < 2 ** 31
                t = self._tabs[key] = (a_row.reshape(-1).to(torch.int32).contiguous(), a_k.to(torch.int32).contiguous())
            K = cv.ci * cv.g * 64
            L.gemm([(patches, K, K)], cv.w_pack, dst.f32, M=rows, N=cv.co, compute=self.code, epilogue=epi, bias=cv.bias, resid=rs,
                   relu=relu, ldo=cv.co, ldr=cv.co, a_row_offset=t[0], a_ktile_offset=t[1])
            return dst
        if self.implicit and not cv.window:
            a_row, a_k, o_row = self._tables(cv, src, Ho, Wo, dst.border)
            K = cv.kh * cv.kw * cv.ci
            L.gemm([(src.b16, K, K)], cv.w, dst.f32, M=rows, N=cv.co, compute=self.code, epilogue=epi, bias=cv.bias, resid=rs,
                   relu=relu, ldo=cv.co, ldr=cv.co, out_bf16=dst.b16, ld_out_bf16=cv.co,
                   a_row_offset=a_row, a_ktile_offset=a_k, out_row_offset=o_row)
            return dst
        # explicit patch matrix
        assert dst.border == 0 or cv.window is False
        col = self._colbuf(rows, cv.Kpad)
        if cv.window:
            L.im2col(frames, col, B=n, H=H, W=W, C_=5, kh=cv.kh, kw=cv.kw, stride=cv.stride, pad=cv.pad, Ho=Ho, Wo=Wo,
                     ldo=cv.Kpad, window_t=T, window_first=first)
        else:
            assert src.border == 0
            L.im2col(src.f32, col, B=n, H=H, W=W, C_=cv.ci, kh=cv.kh, kw=cv.kw, stride=cv.stride, pad=cv.pad, Ho=Ho, Wo=Wo,
                     ldo=cv.Kpad)
        L.gemm([(col, cv.Kpad, cv.Kpad)], cv.w, dst.f32, M=rows, N=cv.co, compute=self.code, epilogue=epi, bias=cv.bias, resid=rs,
               relu=relu, ldo=cv.co, ldr=cv.co)
        return dst

    def _pool(self, src, dst_name, k, stride, pad, mode, *, border=0, shadow=False, dst=None):
        Ho, Wo = (src.H + 2 * pad - k) // stride + 1, (src.W + 2 * pad - k) // stride + 1
        if dst is None:
            dst = self._map(dst_name, src.n, Ho, Wo, src.C, border, shadow)
        L.pool2d(src.f32, dst.f32, B=src.n, H=src.H, W=src.W, C_=src.C, k=k, stride=stride, pad=pad, mode=mode, Ho=Ho, Wo=Wo,
                 out_bf16=dst.b16, in_border=src.border, out_border=dst.border)
        return dst

    def _block(self, p, x):
        """BasicBlock (v2r:70-88)."""
        h = self._conv(f"{p}.conv1", x, f"{p}.h", relu=True)
        res = x
        if f"{p}.down" in self.convs:
            res = self._conv(f"{p}.down", x, f"{p}.down", shadow=False)
        return self._conv(f"{p}.conv2", h, f"{p}.out", relu=True, resid=res)

    def _ftb(self, p, x, avg=True, *, to_conv=False, dst=None):
        """FTB (v2r:24-36): its 1x1 conv0 has padding=1 -- over a zero-bordered map that is a plain 1x1 conv of the border too."""
        x1 = self._conv(f"{p}.conv0", x, f"{p}.x1")
        h = self._conv(f"{p}.conv1", x1, f"{p}.h", relu=True)
        o = self._conv(f"{p}.conv2", h, f"{p}.o", resid=x1, shadow=False)
        k, st = (2, 2) if avg else (3, 1)
        b = 1 if (to_conv and self.implicit) else 0
        return self._pool(o, f"{p}.pool", k, st, 0, 1, border=b, shadow=b == 1, dst=dst)

    def _windows(self, frames, patches, T, first, n, H, W, head, row, taps=None):
        """ResNet.forward up to the pyramid maps (v2r:195-222) for windows [first, first + n) of ONE clip: frames (T, H, W) fp32
        (explicit patch matrix) or its packed column patches (implicit GEMM); the four maps the head needs land in rows
        [row, row + n) of the `head` buffers."""
        c1 = self._conv("conv1", None, "c1", relu=True, border=0, shadow=False, window=(frames, patches, T, first, n, H, W))
        b = 1 if self.implicit else 0
        h = self._pool(c1, "mp", 3, 2, 1, 0, border=b, shadow=self.implicit)
        feats = []
        for name, _, _ in _LAYERS:
            h = self._block(f"{name}.0", h)
            h = self._block(f"{name}.1", h)
            feats.append(h)
        x1, x2, x3, x4 = feats
        P = x4.H * x4.W
        view = lambda k, C: self._Map(head[k][row:row + n].view(n, x4.H, x4.W, C), None, n, x4.H, x4.W, C, 0)
        x5 = self._conv("toplayer", x4, "x5", relu=True, dst=view("x5", 64))
        t = self._ftb("FTB2_1", x2, to_conv=True)
        shapes = []
        for nm, srcm, avg, key in (("FTB2_2", t, True, "x2"), ("FTB3", x3, True, "x3"), ("FTB4", x4, False, "x4")):
            k, st = (2, 2) if avg else (3, 1)
            shapes.append(((srcm.H + 2 - k) // st + 1, (srcm.W + 2 - k) // st + 1))
        if any(sh != (x4.H, x4.W) for sh in shapes):
            raise L.V2AError(f"Video2Roll: pyramid maps disagree for a {H}x{W} input: {shapes} vs {(x4.H, x4.W)}")
        x2_ = self._ftb("FTB2_2", t, dst=view("x2", 128))
        x3_ = self._ftb("FTB3", x3, dst=view("x3", 128))
        x4_ = self._ftb("FTB4", x4, avg=False, dst=view("x4", 128))
        if taps is not None:
            nchw = lambda m: m.interior().permute(0, 3, 1, 2).clone()
            for k, v in dict(c1=c1, x1=x1, x2=x2, x3=x3, x4=x4, x5=x5, x2_=x2_, x3_=x3_, x4_=x4_).items():
                taps.setdefault(k, []).append(nchw(v))
        return P

    def _head_buffers(self, total, H, W):
        cv = self.convs
        h1, w1 = cv["conv1"].out_hw(H, W)
        hc, wc = (h1 + 2 - 3) // 2 + 1, (w1 + 2 - 3) // 2 + 1
        for _ in range(3):                                   # layer2..4 halve the map (3x3, stride 2, pad 1)
            hc, wc = (hc + 2 - 3) // 2 + 1, (wc + 2 - 3) // 2 + 1
        P = hc * wc
        key = ("head", total, P)
        hb = self._maps.get(key)
        if hb is None:
            e = lambda C: torch.empty(total, P, C, device=self.dev, dtype=torch.float32)
            hb = self._maps[key] = dict(x2=e(128), x3=e(128), x4=e(128), x5=e(64))
        return hb, P

    def _pack(self, clip_frames, T, H, W):
        """bf16 column patches of one clip for the implicit first layer (v2a_frames_pack)."""
        cv = self.convs["conv1"]
        _, Wo = cv.out_hw(H, W)
        key = ("patches", T, H, W)
        buf = self._maps.get(key)
        if buf is None:
            buf = self._maps[key] = torch.empty((T + 4) * Wo * (H + 2 * cv.pad) * 16, device=self.dev, dtype=torch.bfloat16)
        L.frames_pack(clip_frames, buf, T=T, H=H, W=W, kw=cv.kw, stride=cv.stride, pad=cv.pad, Wo=Wo)
        return buf

    def _run(self, frames, T, sigmoid, windows=None, taps=None):
        """frames (clips, T, H, W).  windows: explicit (clip, index) pairs (tests); default every window, clip by clip in chunks."""
        clips, _, H, W = frames.shape
        if windows is None:
            spans = [(c, f, min(self.chunk, T - f)) for c in range(clips) for f in range(0, T, self.chunk)]
        else:
            spans = [(c, w, 1) for c, w in windows]
        total = sum(n for _, _, n in spans)
        head, P = self._head_buffers(total, H, W)
        row, packed = 0, None
        for clip, first, n in spans:
            patches = None
            if self.implicit:
                if packed != clip:
                    pbuf, packed = self._pack(frames[clip], T, H, W), clip
                patches = pbuf
            self._windows(frames[clip], patches, T, first, n, H, W, head, row, taps)
            row += n
        out = torch.empty(total, self.notes, device=self.dev, dtype=torch.float32)
        a = L.RollHeadArgs()
        a.x2, a.x3, a.x4, a.x5 = head["x2"].data_ptr(), head["x3"].data_ptr(), head["x4"].data_ptr(), head["x5"].data_ptr()
        a.B, a.P = total, P
        for k, v in self.head_w.items():
            setattr(a, k, v.data_ptr())
        a.notes, a.apply_sigmoid, a.out = self.notes, 1 if sigmoid else 0, out.data_ptr()
        L.roll_head(a)
        return out

    # ---- reference-shaped entry points ----------------------------------------------------------------
    @torch.no_grad()
    def forward_windows(self, x, taps=None):
        """`ResNet.forward` on explicit windows (n, 5, H, W) -> logits (n, notes) (v2r:195-251).  Each window is loaded as
        its own 5-frame clip; the clip's centre window (index 2) is exactly that un-clamped stack (parity / test entry)."""
        n, c, H, W = x.shape
        assert c == 5
        frames = x.to(self.dev, torch.float32).contiguous()          # (n clips, T = 5 frames, H, W)
        return self._run(frames, 5, False, windows=[(i, 2) for i in range(n)], taps=taps)

    @torch.no_grad()
    def encode_frames(self, x, l: int):
        """`E2TTS.encode_frames(x, l)` (x3:1525-1553): x (b, 1, t, H, W) -> roll probabilities (b, l, notes)."""
        b, c, t, H, W = x.shape
        assert c == 1
        frames = x[:, 0].to(self.dev, torch.float32).contiguous()
        roll = self._run(frames, t, True)
        out = torch.empty(b, int(l), self.notes, device=self.dev, dtype=torch.float32)
        L.roll_expand(roll, out, B=b, t=t, notes=self.notes, rep=3, l=int(l))
        return out
