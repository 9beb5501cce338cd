#!/usr/bin/env python3
"""Which kernel gives different results while another process keeps the GPU busy?  (debug aid)  Each op: reference on a quiet device, then
`reps` launches beside a child process that runs matmuls / fills on the same GPU; prints how many launches differ and by how much."""
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import v2a_amd  # noqa: E402,F401
from v2a_amd import _lib as L  # noqa: E402

DEV = torch.device("cuda:0")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(0)
R = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)

ops = {}


def op(name):
    def deco(f):
        ops[name] = f
        return f
    return deco


B, N = 10, 128
for d in (256, 320, 128):
    x, wt, bias = R(B, N, d), R(31, d, sc=0.2), R(d, sc=0.1)
    gam = R(d)

    @op(f"dwconv d={d}")
    def _(x=x, wt=wt, bias=bias, d=d):
        out = torch.empty_like(x)
        L.dwconv(x, out, wt, bias, B=B, N=N, d=d, ksize=31)
        return out

    @op(f"dwconv+norm d={d}")
    def _(x=x, wt=wt, bias=bias, d=d, gam=gam):
        out = torch.empty_like(x)
        sh = torch.zeros(B * N, d, dtype=torch.bfloat16, device=DEV)
        ssq = torch.zeros(B * N, (d // 32 + 3) // 4 * 4, device=DEV)
        L.dwconv(x, out, wt, bias, B=B, N=N, d=d, ksize=31, norm=dict(out_bf16=sh, gamma=gam, ssq=ssq))
        return torch.cat([out.reshape(B * N, d), sh.float(), ssq], 1)

    @op(f"rmsnorm d={d}")
    def _(x=x, d=d, gam=gam):
        y = torch.empty(B * N, d, device=DEV)
        L.rmsnorm(x.reshape(B * N, d), y, rows=B * N, d=d, gamma=gam)
        return y

M = B * N
for (Nn, K) in ((256, 256), (784, 256), (2048, 256), (256, 1024), (320, 576)):
    a32, w32, res = R(M, K, sc=0.5), R(Nn, K, sc=1 / math.sqrt(K)), R(M, Nn)

    @op(f"gemm fp32 {M}x{Nn}x{K} resid")
    def _(a32=a32, w32=w32, res=res, Nn=Nn, K=K):
        out = torch.empty(M, Nn, device=DEV)
        L.gemm([(a32, K, K)], w32, out, M=M, N=Nn, compute=L.F32, epilogue=L.EPI_RESID, resid=res)
        return out

    ab, wb = a32.bfloat16(), w32.bfloat16()

    @op(f"gemm bf16 {M}x{Nn}x{K} resid")
    def _(ab=ab, wb=wb, res=res, Nn=Nn, K=K):
        out = torch.empty(M, Nn, device=DEV)
        L.gemm([(ab, K, K)], wb, out, M=M, N=Nn, compute=L.BF16, epilogue=L.EPI_RESID, resid=res)
        return out

H = 4
for dt, code in ((torch.float32, L.F32), (torch.bfloat16, L.BF16), (torch.float32, L.BF16_SPLIT)):
    inner = H * 64
    qkv = R(B, N, 3 * inner + 16).to(dt)

    @op(f"attention {code}")
    def _(qkv=qkv, dt=dt, code=code, inner=inner):
        out = torch.empty(B, N, inner, dtype=dt, device=DEV)
        es = qkv.element_size()
        ld = 3 * inner + 16
        L.attention(qkv.data_ptr(), qkv.data_ptr() + inner * es, qkv.data_ptr() + 2 * inner * es, qkv.data_ptr() + 3 * inner * es, out.data_ptr(),
                    strides=(ld, ld, ld, ld, inner, N * ld, N * ld, N * ld, N * ld, N * inner), B=B, H=H, Nq=N, Nk=N, scale=0.125, softclamp=50.0, dtype=code)
        return out.float()

# epilogues whose compiled code has packed-fp32 instructions with op_sel modifiers: STORE + RoPE, GEGLU, the one-launch cross-attention
inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
ang = torch.arange(N + 8).float()[:, None] * inv[None, :]
TAB = torch.stack((ang.cos(), ang.sin()), -1).contiguous().to(DEV)
for (Nn, K) in ((784, 256), (2048, 256)):
    ab, wb, bb = R(M, K, sc=0.5).bfloat16(), R(Nn, K, sc=1 / math.sqrt(K)).bfloat16(), R(Nn)

    @op(f"gemm bf16 {M}x{Nn}x{K} store+rope")
    def _(ab=ab, wb=wb, bb=bb, Nn=Nn, K=K):
        out = torch.empty(M, Nn, device=DEV, dtype=torch.bfloat16)
        L.gemm([(ab, K, K)], wb, out, M=M, N=Nn, compute=L.BF16, bias=bb, rope_table=TAB, rope_cols=512, rope_pos_offset=0, rows_per_batch=N)
        return out.float()

    @op(f"gemm bf16 {M}x{Nn}x{K} store+rope 8-phase")
    def _(ab=ab, wb=wb, bb=bb, Nn=Nn, K=K):
        out = torch.empty(M, Nn, device=DEV, dtype=torch.bfloat16)
        L.gemm([(ab, K, K)], wb, out, M=M, N=Nn, compute=L.BF16, bias=bb, rope_table=TAB, rope_cols=512, rope_pos_offset=0, rows_per_batch=N, tile_hint=7)
        return out.float()

    if Nn % 32 == 0:
        @op(f"gemm bf16 {M}x{Nn}x{K} geglu")
        def _(ab=ab, wb=wb, bb=bb, Nn=Nn, K=K):
            out = torch.empty(M, Nn // 2, device=DEV, dtype=torch.bfloat16)
            L.gemm([(ab, K, K)], wb, out, M=M, N=Nn, compute=L.BF16, bias=bb, epilogue=L.EPI_GEGLU, ldo=Nn // 2)
            return out.float()

        @op(f"gemm bf16 {M}x{Nn}x{K} geglu 8-phase")
        def _(ab=ab, wb=wb, bb=bb, Nn=Nn, K=K):
            out = torch.empty(M, Nn // 2, device=DEV, dtype=torch.bfloat16)
            L.gemm([(ab, K, K)], wb, out, M=M, N=Nn, compute=L.BF16, bias=bb, epilogue=L.EPI_GEGLU, ldo=Nn // 2, tile_hint=7)
            return out.float()

Hx, Kx, Nkx = 4, 512, 16
ax, wx, bx = R(M, Kx, sc=0.5).bfloat16(), R(Hx * 64 + 16, Kx, sc=1 / math.sqrt(Kx)).bfloat16(), R(Hx * 64 + 16)
kvx = R(B, Nkx, 2 * Hx * 64).bfloat16()


@op("qproj_xattn bf16")
def _():
    out = torch.empty(B, N, Hx * 64, device=DEV, dtype=torch.bfloat16)
    inner = Hx * 64
    L.qproj_xattn(ax, Kx, Kx, wx, bias=bx, M=M, N=inner + 16, rows_per_batch=N, k=kvx.data_ptr(), v=kvx.data_ptr() + inner * 2, out=out.data_ptr(),
                  kv_strides=(2 * inner, 2 * inner, Nkx * 2 * inner, Nkx * 2 * inner), out_strides=(inner, N * inner), B=B, H=Hx, Nk=Nkx,
                  scale=0.125, softclamp=50.0, rope_table=TAB, rope_cols=inner, rope_pos_offset=0)
    return out.float()


y = R(5, 120, 32)
wt_in, b_in, pos, regs = R(32, 256), R(256), R(120, 256), R(8, 256)


@op("linear_small (embed)")
def _():
    out = torch.zeros(10, 128, 256, device=DEV)
    L.linear_small(y, wt_in, b_in, pos, out, M=5 * 120, K=32, T=120, out_batch_stride=128 * 256, row_off=8, d=256, dup=5, regs=regs)
    return out


def main():
    ref = {}
    for k, f in ops.items():
        ref[k] = f().float().cpu()
        again = f().float().cpu()
        assert torch.equal(ref[k], again), k
    child = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "concurrency_probe.py"), "--load", "120", "matmul"])
    time.sleep(5.0)
    for k, f in ops.items():
        worst, bad = 0.0, 0
        for _ in range(REPS):
            d = float((f().float().cpu() - ref[k]).abs().nan_to_num(9e9).max())
            worst = max(worst, d)
            bad += d != 0.0
        print(f"{k:36s} {bad:3d} of {REPS} launches differ, worst {worst:.3e}", flush=True)
    child.kill()
    child.wait()


if __name__ == "__main__":
    main()
