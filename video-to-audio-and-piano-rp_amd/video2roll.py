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
        self.bias = None if bias is None else bias.to(dev).contiguous()
        self.co, self.ci, self.kh, self.kw, self.stride, self.pad, self.window = co, ci, kh, kw, stride, pad, window

    def out_hw(self, H, W):
        return (H + 2 * self.pad - self.kh) // self.stride + 1, (W + 2 * self.pad - self.kw) // self.stride + 1


class Video2RollEngine:
    """`video2roll_net` + `encode_frames` of the reference E2TTS on HIP kernels.

    sd: state dict of the reference module (`video2roll_net.` prefix already stripped, or pass prefix=).
    compute: "bf16" (bf16 MFMA operands, fp32 accumulate / activations) or "fp32" (parity mode, exact-fp32 MFMA).
    chunk: windows per pass (bounds the patch-matrix scratch: 28 MB per window in bf16 for the first layer)."""

    def __init__(self, sd, device="cuda:0", compute="bf16", prefix="", chunk=25):
        L.lib()                                                   # fail loudly without the HIP library
        self.dev = torch.device(device)
        if compute not in ("bf16", "fp32"):
            raise ValueError(f"compute must be 'bf16' or 'fp32', got {compute!r}")
        self.cd = torch.bfloat16 if compute == "bf16" else torch.float32
        self.code = L.BF16 if compute == "bf16" else L.F32
        self.chunk = int(chunk)
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
        self._bufs: dict = {}
        self._col = None

    # ---- scratch ------------------------------------------------------------------------------
    def _buf(self, name, *shape):
        t = self._bufs.get(name)
        n = math.prod(shape)
        if t is None or t.numel() < n:
            t = torch.empty(n, device=self.dev, dtype=torch.float32)
            self._bufs[name] = t
        return t[:n].view(*shape)

    def _colbuf(self, rows, kpad):
        n = rows * kpad
        if self._col is None or self._col.numel() < n:
            self._col = torch.empty(n, device=self.dev, dtype=self.cd)
        return self._col[:n].view(rows, kpad)

    # ---- one convolution: im2col + GEMM (+bias, +residual, +ReLU in the epilogue) ---------------------
    def _conv(self, name, x, n, H, W, out_name, *, relu=False, resid=None, window=None):
        cv = self.convs[name]
        Ho, Wo = cv.out_hw(H, W)
        rows = n * Ho * Wo
        col = self._colbuf(rows, cv.Kpad)
        if cv.window:
            frames, T, first = window
            L.im2col(frames, col, B=n, H=H, W=W, C_=5, kh=cv.kh, kw=cv.kw, stride=cv.stride, pad=cv.pad, Ho=Ho, Wo=Wo,
                     ldo=cv.Kpad, window_t=T, window_first=first)
        else:
            L.im2col(x, col, B=n, H=H, W=W, C_=cv.ci, kh=cv.kh, kw=cv.kw, stride=cv.stride, pad=cv.pad, Ho=Ho, Wo=Wo,
                     ldo=cv.Kpad)
        out = self._buf(out_name, n, Ho, Wo, cv.co)
        o2 = out.view(rows, cv.co)
        L.gemm([(col, cv.Kpad, cv.Kpad)], cv.w, o2, M=rows, N=cv.co, compute=self.code,
               epilogue=L.EPI_RESID if resid is not None else L.EPI_STORE, bias=cv.bias,
               resid=None if resid is None else resid.view(rows, cv.co), relu=relu)
        return out, Ho, Wo

    def _block(self, p, x, n, H, W, tag):
        """BasicBlock (v2r:70-88)."""
        h, Ho, Wo = self._conv(f"{p}.conv1", x, n, H, W, f"{tag}.h", relu=True)
        res = x
        if f"{p}.down" in self.convs:
            res, _, _ = self._conv(f"{p}.down", x, n, H, W, f"{tag}.down")
        out, _, _ = self._conv(f"{p}.conv2", h, n, Ho, Wo, f"{tag}.out", relu=True, resid=res)
        return out, Ho, Wo

    def _ftb(self, p, x, n, H, W, avg=True):
        """FTB (v2r:24-36)."""
        x1, H1, W1 = self._conv(f"{p}.conv0", x, n, H, W, f"{p}.x1")
        h, _, _ = self._conv(f"{p}.conv1", x1, n, H1, W1, f"{p}.h", relu=True)
        o, _, _ = self._conv(f"{p}.conv2", h, n, H1, W1, f"{p}.o", resid=x1)
        k, st = (2, 2) if avg else (3, 1)
        Ho, Wo = (H1 - k) // st + 1, (W1 - k) // st + 1
        out = self._buf(f"{p}.pool", n, Ho, Wo, 128)
        L.pool2d(o, out, B=n, H=H1, W=W1, C_=128, k=k, stride=st, pad=0, mode=1, Ho=Ho, Wo=Wo)
        return out, Ho, Wo

    def _windows(self, frames, T, first, n, H, W, out, sigmoid, taps=None):
        """ResNet.forward (v2r:195-251) for windows [first, first + n) of frames (clips, T, H, W); writes out (n, notes)."""
        c1, H1, W1 = self._conv("conv1", None, n, H, W, "c1", relu=True, window=(frames, T, first))
        Hp, Wp = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        h = self._buf("mp", n, Hp, Wp, 64)
        L.pool2d(c1, h, B=n, H=H1, W=W1, C_=64, k=3, stride=2, pad=1, mode=0, Ho=Hp, Wo=Wp)
        feats = []
        Hc, Wc = Hp, Wp
        for name, _, _ in _LAYERS:
            h, Hc, Wc = self._block(f"{name}.0", h, n, Hc, Wc, f"{name}.0")
            h, Hc, Wc = self._block(f"{name}.1", h, n, Hc, Wc, f"{name}.1")
            feats.append((h, Hc, Wc))
        (x1, _, _), (x2, H2, W2), (x3, H3, W3), (x4, H4, W4) = feats
        x5, _, _ = self._conv("toplayer", x4, n, H4, W4, "x5", relu=True)
        t, Ht, Wt = self._ftb("FTB2_1", x2, n, H2, W2)
        x2_, Ha, Wa = self._ftb("FTB2_2", t, n, Ht, Wt)
        x3_, Hb, Wb = self._ftb("FTB3", x3, n, H3, W3)
        x4_, Hc4, Wc4 = self._ftb("FTB4", x4, n, H4, W4, avg=False)
        if not ((Ha, Wa) == (Hb, Wb) == (Hc4, Wc4) == (H4, W4)):
            raise L.V2AError(f"Video2Roll: pyramid maps disagree for a {H}x{W} input: {(Ha, Wa)}, {(Hb, Wb)}, {(Hc4, Wc4)}, {(H4, W4)}")
        a = L.RollHeadArgs()
        a.x2, a.x3, a.x4, a.x5 = x2_.data_ptr(), x3_.data_ptr(), x4_.data_ptr(), x5.data_ptr()
        a.B, a.P = n, H4 * W4
        for k, v in self.head_w.items():
            setattr(a, k, v.data_ptr())
        a.notes, a.apply_sigmoid, a.out = self.notes, 1 if sigmoid else 0, out.data_ptr()
        L.roll_head(a)
        if taps is not None:
            nchw = lambda v: v.permute(0, 3, 1, 2).clone()
            for k, v in dict(c1=c1, x1=x1, x2=x2, x3=x3, x4=x4, x5=x5, x2_=x2_, x3_=x3_, x4_=x4_).items():
                taps.setdefault(k, []).append(nchw(v))

    def _run(self, frames, T, sigmoid, taps=None):
        clips, _, H, W = frames.shape
        total = clips * T
        out = torch.empty(total, self.notes, device=self.dev, dtype=torch.float32)
        for first in range(0, total, self.chunk):
            n = min(self.chunk, total - first)
            self._windows(frames, T, first, n, H, W, out[first:first + n], sigmoid, taps)
        return out

    # ---- reference-shaped entry points ----------------------------------------------------------------
    @torch.no_grad()
    def forward_windows(self, x, taps=None):
        """`ResNet.forward` on explicit windows (n, 5, H, W) -> logits (n, notes) (v2r:195-251).  Each window is treated
        as its own 5-frame clip whose centre frame sees exactly channels 0..4 (parity / test entry point)."""
        n, c, H, W = x.shape
        assert c == 5
        frames = x.to(self.dev, torch.float32).contiguous()          # (n clips, T = 5 frames, H, W)
        out = torch.empty(n, self.notes, device=self.dev, dtype=torch.float32)
        # window index 2 of every 5-frame clip is the un-clamped stack; run them one clip-centre at a time per chunk
        for i in range(n):
            self._windows(frames, 5, 5 * i + 2, 1, H, W, out[i:i + 1], False, taps)
        return out

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
