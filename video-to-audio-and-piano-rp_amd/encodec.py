"""Encodec 24 kHz decoder (the vocoder) on the MI355X kernels.

Mirrors `EncodecWrapper.decode` (src/e2_tts_pytorch/e2_tts_crossatt3.py:434-437): `self.model.decoder(emb)` then
`output[0]`, called on the sampler's latents at predict.py:277-278.  The network is the SEANet decoder of
`transformers.models.encodec` (reference pin transformers==4.46.0): Conv1d(128->512, k7), 2-layer LSTM + skip, four
[ELU, ConvTranspose1d(k = 2r, stride r), residual block] stages for r = 8, 5, 4, 2, ELU, Conv1d(32->1, k7); all
convolutions causal and weight-normalised.  750 latent frames -> 240 000 samples.

Design: activations are time-major [T][C] fp32.  A causal Conv1d(k) is then ONE `v2a_gemm` whose A rows overlap
(lda = C, K = k*C) over a buffer with k-1 reflected rows in front; a ConvTranspose1d(k = 2r, stride r) is ONE `v2a_gemm`
with A row q = [x[q-1], x[q]] (K = 2C) and N = r*Cout columns ordered (phase, channel), whose [L][r*Cout] output is the
up-sampled [L*r][Cout] signal in place -- no col2im, no scatter.  `v2a_elu_pad` applies ELU and writes the pad rows in one
pass; `v2a_lstm2` runs both LSTM layers' recurrences in one persistent kernel (weights in registers, T + 1 exchange steps).  Weight norm is
resolved at load time.  Everything is fp32 (exact-fp32 MFMA): the whole decoder is ~30 GFLOP, memory- and latency-bound.
There is no CPU fallback: without libv2a_cfm.so every call raises.
"""
from __future__ import annotations

import torch

from . import _lib as L

RATIOS = (8, 5, 4, 2)
HIDDEN, FILTERS = 128, 32


def expected_state_dict_shapes() -> dict[str, tuple]:
    """Key layout of `EncodecModel(EncodecConfig()).decoder.state_dict()` (weight_norm parametrisation)."""
    s: dict[str, tuple] = {}

    def conv(p, co, ci, k, transpose=False):
        s[f"{p}.conv.bias"] = (co,)
        s[f"{p}.conv.parametrizations.weight.original0"] = ((ci if transpose else co), 1, 1)
        s[f"{p}.conv.parametrizations.weight.original1"] = (ci, co, k) if transpose else (co, ci, k)

    c = FILTERS * 2 ** len(RATIOS)
    conv("layers.0", c, HIDDEN, 7)
    for l in range(2):
        s[f"layers.1.lstm.weight_ih_l{l}"] = (4 * c, c)
        s[f"layers.1.lstm.weight_hh_l{l}"] = (4 * c, c)
        s[f"layers.1.lstm.bias_ih_l{l}"] = (4 * c,)
        s[f"layers.1.lstm.bias_hh_l{l}"] = (4 * c,)
    idx = 3
    for r in RATIOS:
        conv(f"layers.{idx}", c // 2, c, 2 * r, transpose=True)
        c //= 2
        conv(f"layers.{idx + 1}.block.1", c // 2, c, 3)
        conv(f"layers.{idx + 1}.block.3", c, c // 2, 1)
        conv(f"layers.{idx + 1}.shortcut", c, c, 1)
        idx += 3
    conv(f"layers.{idx}", 1, FILTERS, 7)
    return s


def _resolve_weight(sd, p):
    """Plain `conv.weight`, the parametrised (original0 = g, original1 = v) pair, or the legacy weight_g / weight_v pair
    of the hub checkpoint: w = g * v / ||v|| with the norm over all dims but the first (torch weight_norm, dim 0)."""
    if f"{p}.conv.weight" in sd:
        return sd[f"{p}.conv.weight"].float()
    for gk, vk in ((f"{p}.conv.parametrizations.weight.original0", f"{p}.conv.parametrizations.weight.original1"),
                   (f"{p}.conv.weight_g", f"{p}.conv.weight_v")):
        if gk in sd and vk in sd:
            g, v = sd[gk].float(), sd[vk].float()
            return g * v / v.norm(dim=(1, 2), keepdim=True)
    raise KeyError(f"EncodecDecoder: no weight for {p}.conv (looked for .weight, parametrizations.weight.original0/1, weight_g/v)")


class EncodecDecoder:
    """HIP mirror of `EncodecWrapper.decode` / `EncodecModel.decoder`.

    state_dict: an `EncodecModel` state dict (keys `decoder.layers...`) or the decoder's own (`layers...`)."""

    def __init__(self, state_dict, device="cuda:0"):
        L.lib()                                                   # fail loudly without the HIP library
        self.dev = dev = torch.device(device)
        sd = {k: v.detach().cpu() for k, v in state_dict.items()}
        if any(k.startswith("decoder.layers.") for k in sd):
            sd = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
        f32 = lambda t: t.float().contiguous().to(dev)

        def conv(p):
            w = _resolve_weight(sd, p)                            # (co, ci, k)
            co, ci, k = w.shape
            return dict(w=f32(w.permute(0, 2, 1).reshape(co, k * ci)), b=f32(sd[f"{p}.conv.bias"]), co=co, ci=ci, k=k)

        def convt(p):
            w = _resolve_weight(sd, p)                            # (ci, co, k = 2r)
            ci, co, k = w.shape
            r = k // 2
            assert k == 2 * r
            # row (phase, co); columns [x[q-1] part: tap phase + r | x[q] part: tap phase]
            wp = torch.cat([w[:, :, r:].permute(2, 1, 0), w[:, :, :r].permute(2, 1, 0)], 2).reshape(r * co, 2 * ci)
            return dict(w=f32(wp), b=f32(sd[f"{p}.conv.bias"].float().repeat(r)), co=co, ci=ci, r=r)

        self.c0 = conv("layers.0")
        self.H = self.c0["co"]
        self.lstm = []
        for l in range(2):
            self.lstm.append(dict(wih=f32(sd[f"layers.1.lstm.weight_ih_l{l}"]), whh=f32(sd[f"layers.1.lstm.weight_hh_l{l}"]),
                                  b=f32(sd[f"layers.1.lstm.bias_ih_l{l}"].float() + sd[f"layers.1.lstm.bias_hh_l{l}"].float())))
        self.stages = []
        idx = 3
        for r in RATIOS:
            self.stages.append(dict(up=convt(f"layers.{idx}"), b1=conv(f"layers.{idx + 1}.block.1"), b3=conv(f"layers.{idx + 1}.block.3"),
                                    sc=conv(f"layers.{idx + 1}.shortcut")))
            assert self.stages[-1]["up"]["r"] == r
            idx += 3
        self.cf = conv(f"layers.{idx}")
        self.hop = 1
        for r in RATIOS:
            self.hop *= r
        self._bufs: dict = {}
        self._ws = torch.zeros(8 * self.H + 2, dtype=torch.int32, device=dev)      # exchange tables of both LSTM layers + error flag

    def _buf(self, name, rows, C):
        key = (name, rows, C)
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.empty(rows, C, device=self.dev, dtype=torch.float32)
        return t

    def _conv(self, cv, src, T, name, *, act, resid=None):
        """Causal Conv1d on time-major src (T, ci): ELU (optional) + reflect pad, then one GEMM over overlapping rows."""
        k, ci, co = cv["k"], cv["ci"], cv["co"]
        a = src
        if act or k > 1:
            a = self._buf(name + ".in", T + k - 1, ci)
            L.elu_pad(src, a, T=T, C_=ci, pad=k - 1, reflect=True, act=act)
        out = self._buf(name, T, co)
        L.gemm([(a, ci, k * ci)], cv["w"], out, M=T, N=co, compute=L.F32, bias=cv["b"], ldo=co,
               epilogue=L.EPI_RESID if resid is not None else L.EPI_STORE, resid=resid, ldr=co)
        return out

    def _decode_one(self, emb, taps=None):
        """emb (128, T) on the device -> waveform (T * 320,)."""
        T = emb.shape[1]
        H = self.H
        x0 = emb.t().contiguous()                                                  # time-major (T, 128)
        h0 = self._conv(self.c0, x0, T, "c0", act=False)
        # 2-layer LSTM + skip: one GEMM for layer 0's input projection, then both recurrences in one persistent kernel
        l0, l1 = self.lstm
        gx = self._buf("gx0", T, 4 * H)
        L.gemm([(h0, H, H)], l0["wih"], gx, M=T, N=4 * H, compute=L.F32, bias=l0["b"], ldo=4 * H)
        x = self._buf("lstm_out", T, H)
        L.lstm2(gx, l0["whh"], l1["wih"], l1["b"], l1["whh"], x, self._ws, T=T, H=H, resid=h0)
        if taps is not None:
            taps["lstm"] = x.t().clone()
        Lc, C = T, H
        for si, st in enumerate(self.stages):
            up = st["up"]
            r, co = up["r"], up["co"]
            a = self._buf(f"s{si}.upin", Lc + 1, C)
            L.elu_pad(x, a, T=Lc, C_=C, pad=1, reflect=False, act=True)
            u = self._buf(f"s{si}.up", Lc, r * co)                                  # == (Lc * r, co) time-major
            L.gemm([(a, C, 2 * C)], up["w"], u, M=Lc, N=r * co, compute=L.F32, bias=up["b"], ldo=r * co)
            Lc, C = Lc * r, co
            u = u.view(Lc, C)
            h1 = self._conv(st["b1"], u, Lc, f"s{si}.b1", act=True)
            h2 = self._conv(st["b3"], h1, Lc, f"s{si}.b3", act=True)
            x = self._conv(st["sc"], u, Lc, f"s{si}.out", act=False, resid=h2)
            if taps is not None:
                taps[f"stage{r}"] = x.t().clone()
        wav = self._conv(self.cf, x, Lc, "final", act=True)                         # (Lc, 1)
        return wav.view(Lc)

    @torch.no_grad()
    def decoder(self, emb, taps=None):
        """`EncodecModel.decoder(emb)`: emb (b, 128, T) -> (b, 1, 320*T) fp32 on the device."""
        assert emb.ndim == 3 and emb.shape[1] == HIDDEN, f"emb must be (b, {HIDDEN}, T), got {tuple(emb.shape)}"
        if emb.shape[2] < 7:
            raise ValueError("EncodecDecoder: need at least 7 latent frames (reflect padding of the k=7 convolutions)")
        emb = emb.to(self.dev, torch.float32)
        out = torch.empty(emb.shape[0], 1, emb.shape[2] * self.hop, device=self.dev, dtype=torch.float32)
        for i in range(emb.shape[0]):
            out[i, 0].copy_(self._decode_one(emb[i], taps if i == 0 else None))
            if int(self._ws[8 * self.H].item()):                                   # one host sync per clip
                raise L.V2AError("v2a_lstm2: a workgroup timed out at the step barrier (GPU oversubscribed?); result discarded")
        return out

    def decode(self, emb):
        """`EncodecWrapper.decode(emb)` (x3:434-437): the first clip's waveform, shape (1, samples)."""
        return self.decoder(emb)[0]
