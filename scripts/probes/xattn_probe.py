#!/usr/bin/env python3
"""Stand-alone timing of the audio stream's cross-attention at one clip (tuning aid): v2a_gemm (q-projection, RoPE, folded norm) +
v2a_attention against v2a_qproj_xattn, each as 20 back-to-back launches in a hipGraph.
usage: python scripts/probes/xattn_probe.py [clips]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import v2a_amd  # noqa: E402,F401
from v2a_amd import _lib as L  # noqa: E402

DEV = torch.device("cuda:0")
REPS = 20


def time_graph(fn):
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st):
            for _ in range(REPS):
                fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
    return best


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    Nq, Nk, H, K = 782, 32, 16, 1024
    M, inner, N = B * Nq, H * 64, H * 64 + 16
    g = torch.Generator().manual_seed(0)
    a = (torch.randn(M, K, generator=g) * 0.7).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    kv = torch.randn(B, Nk, 2 * inner, generator=g).bfloat16().to(DEV)
    inv = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(Nq + 8).float()[:, None] * inv[None, :]
    tab = torch.stack((ang.cos(), ang.sin()), -1).contiguous().to(DEV)
    ssq = torch.zeros(M, 40, device=DEV)
    ssq[:, :32] = 20.0
    kvl = torch.full((B,), 20, dtype=torch.int32, device=DEV)
    qb = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(B, Nq, inner, dtype=torch.bfloat16, device=DEV)
    rk = dict(rope_table=tab, rope_cols=inner, rope_pos_offset=0)
    nk = dict(row_ssq=ssq, row_norm_dim=K)

    def gemm(tile):
        L.gemm([(a, K, K)], w, qb, M=M, N=N, compute=L.BF16, bias=bias, rows_per_batch=Nq, tile_hint=tile + 1, **rk, **nk)

    def attn():
        L.attention(qb.data_ptr(), kv.data_ptr(), kv.data_ptr() + inner * 2, qb.data_ptr() + inner * 2, out.data_ptr(),
                    strides=(N, 2 * inner, 2 * inner, N, inner, Nq * N, Nk * 2 * inner, Nk * 2 * inner, Nq * N, Nq * inner),
                    B=B, H=H, Nq=Nq, Nk=Nk, kv_len=kvl, q_len=None, scale=0.125, softclamp=50.0, dtype=L.BF16)

    def fused():
        L.qproj_xattn(a, K, K, w, bias=bias, M=M, N=N, rows_per_batch=Nq, k=kv.data_ptr(), v=kv.data_ptr() + inner * 2, out=out.data_ptr(),
                      kv_strides=(2 * inner, 2 * inner, Nk * 2 * inner, Nk * 2 * inner), out_strides=(inner, Nq * inner), B=B, H=H, Nk=Nk,
                      kv_len=kvl, q_len=None, scale=0.125, softclamp=50.0, **rk, **nk)

    for tile in (3, 14, -1, 6):
        tg = time_graph(lambda: gemm(tile))
        tb = time_graph(lambda: (gemm(tile), attn()))
        print(f"{B} clip(s): q-projection on tile {tile:2d} {tg:6.2f} us; + attention {tb:6.2f} us", flush=True)
    print(f"{B} clip(s): attention alone {time_graph(attn):6.2f} us; one launch {time_graph(fused):6.2f} us", flush=True)


if __name__ == "__main__":
    main()
