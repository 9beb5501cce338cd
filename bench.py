"""Headline benchmark: mel-frames/sec of the flow-matching V2A sampler (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full sample() of this rank's clip shard: VGGSound 10 s shape (750 latent frames
x 128 channels), 32-point sway grid = 31 CFG evaluations = 62 forwards of the 776.6 M-parameter
three-stream DiT, bf16 operands / fp32 accumulate, CFG 2.0 -- BASELINE.json configs[1].  Weights are
random-init and conditioning is synthetic (no checkpoints or videos exist offline) but of the real
shapes; they are resident in HBM before the timed region, which covers everything sample() does
(modulation tables, cross-attention K/V, the Euler loop) plus, for N > 1, the single all-gather
of latents.  Ranks are clip-sharded (weak scaling, no per-step communication).

Rank 0 prints ONE JSON line.  At N = 1 it also carries
  roofline     : the dominant kernel class (the bf16 MFMA GEMM instantiation with the most time per evaluation) of the
                 configuration the sampler RUNS -- three streams, the per-(stream, op) tile table -- measured live with HIP
                 event pairs on the stream each kernel is launched on (`scope: production`); `standalone` holds the same class
                 with every kernel alone on the chip (one stream, the library's tile choices), `traffic` the recorded PMC bytes
                 per launch next to the algorithmic bytes, and `hbm` the memory-bound kernels at 1 and 8 clips per GPU;
  parity_mode  : throughput AND max |delta mel| over the whole 32-point grid against the committed oracle vector
                 (tests/golden/sample_full.npz) for the fp32 parity mode, the split-bf16 mode (bf16x3) and the bf16
                 headline mode;
  v2p, cascade : BASELINE configs[3] / [4] as supplementary measurements;
  cpu_baseline : the CPU restatement of the reference (oracle/, "port") timed on this box's host
                 cores on a bounded sample of the same workload (steps=4: 3 CFG evaluations; one warm-up run,
                 median of three).
With N > 1 the default is 8 clips per GPU (BASELINE configs[2]: 64 clips on 8 GPUs).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

def log(*a):
    print("[bench %.1fs]" % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


def host_cores() -> int:
    """CPU threads this process may really use: affinity, cgroup quota and torch's own default,
    capped at 16 (the GPU box's CPU share per GPU); os.cpu_count() reports the whole host."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


_T0 = time.perf_counter()
PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
# committed rocprofv3 summaries of THIS build in the headline mode (bf16x3), produced by scripts/gpu_ci.sh (profiles/README.md):
PMC_FILES = ("r05_pmc1_summary.csv", "r05_pmc2_summary.csv")    # FETCH_SIZE pass, WRITE_SIZE pass of the rocprofv3 --pmc runs (one clip, three streams)
PMC_FILES_8CLIPS = ("r05_pmc1_8clips_summary.csv", "r05_pmc2_8clips_summary.csv")   # ... of `--clips-per-gpu 8`
ROCPROF_STATS = "r05_kernel_stats_bf16x3_singlestream.csv"       # rocprofv3 --kernel-trace --stats of `bench.py --single-stream`
ROCPROF_STATS_MULTI = "r05_kernel_stats_bf16x3_multistream.csv"  # ... of the production configuration (three streams)
ROCPROF_STATS_8CLIPS = "r05_kernel_stats_bf16x3_8clips.csv"      # ... of `bench.py --clips-per-gpu 8`
ROCPROF_STATS_8CLIPS_ALONE = "r05_kernel_stats_bf16x3_8clips_singlestream.csv"   # ... of `bench.py --clips-per-gpu 8 --single-stream`: every kernel alone on the chip

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5, help="timed sample() calls")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips-per-gpu", type=int, default=0, help="0 = 1 clip at --gpus 1 (configs[1]), 8 clips per GPU at --gpus N > 1 (configs[2])")
    ap.add_argument("--cfm-steps", type=int, default=32)
    ap.add_argument("--cfg-strength", type=float, default=2.0)
    ap.add_argument("--frames", type=int, default=750)
    ap.add_argument("--dtype", default="bf16x3", choices=["bf16", "fp32", "bf16x3"],
                    help="compute mode of the headline: bf16x3 (default) is the fastest mode whose 32-point sample stays inside north_star's "
                         "|delta mel| < 1e-3 against the oracle; bf16 is ~2x faster and ~50x outside it (reported as roofline.fast_bf16_*)")
    ap.add_argument("--v2p", action="store_true", help="configs[3]: non-zero piano roll, 64 steps")
    ap.add_argument("--cascade", type=int, default=1, help="configs[4]: this many sequential sample() passes per step (the "
                    "reference has no CoT-guidance code, SURVEY 8d: defined here as cascaded 32-step passes, each pass "
                    "starting from fresh noise with the same conditioning)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--single-stream", action="store_true", help="no side streams: kernels run one at a time (profiling aid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--shapes", action="store_true", help="print a per-GEMM-shape timing table to stderr")
    ap.add_argument("--no-batched", action="store_true", help="skip the supplementary 8-clips-per-GPU measurement (N=1 only)")
    ap.add_argument("--cpu-baseline-steps", type=int, default=4, help="grid points of the bounded CPU sample (SURVEY 8d: steps=4)")
    ap.add_argument("--gemm-8phase", type=int, default=-1, help="v2a_set_tuning: 256x256 phase-interleaved GEMM kernel (-1 library default, 0 off, 1 staggered, 2 lock-step)")
    ap.add_argument("--gemm-8phase-min-tiles", type=int, default=0, help="v2a_set_tuning: minimum 256x256 tile count for that kernel (0 = library default)")
    ap.add_argument("--gemm-force-tile", type=int, default=-1, help="v2a_set_tuning: one tile configuration for every bf16 GEMM (experiment)")
    ap.add_argument("--tuning-reserved", type=int, default=0, help="v2a_tuning.reserved[0] (A/B bits: 128 = GEGLU epilogue with 8-byte stores)")
    ap.add_argument("--xcd-1x8", action="store_true", help="A/B: round-1 XCD tile order (column strips) instead of the per-shape rectangle grid")
    ap.add_argument("--interleave-capture", type=int, default=-1, help="A/B: 1 = interleave the capture order of audio and side blocks")
    ap.add_argument("--cross-on-main", action="store_true", help="A/B: all three cross-condition GEMMs on the main stream")
    ap.add_argument("--no-fuse-skip", action="store_true", help="A/B: cross-condition and skip projection as two GEMMs")
    ap.add_argument("--no-fuse-xattn", action="store_true", help="A/B: q-projection and cross-attention of the audio stream as two launches")
    ap.add_argument("--fold-gemm-all", action="store_true", help="A/B: RMSNorms folded into GEMM epilogues at every batch size (default: up to two clips)")
    ap.add_argument("--no-fold-norm", action="store_true", help="A/B: separate RMSNorm launches instead of folding them into the neighbouring kernels")
    ap.add_argument("--big-tiles", default="", help="A/B: the same table for launches of more than two clips, e.g. a.qkv=0,t.ff2=6")
    ap.add_argument("--side-tiles", default="", help="A/B: per-(stream, op) tile configurations of the side-stream GEMMs, e.g. t.qkv=1,f.ff2=2 (ops: cross qkv out ff1 ff2; -1 = library choice)")
    ap.add_argument("--attn-one-group-from", type=int, default=0, help="A/B: workgroup count from which bf16 attention runs one wave group per workgroup (0 = library default)")
    ap.add_argument("--stream-priority", default="", help="experiment: 'A,S' = HIP stream priorities of the audio (capture) stream and of "
                    "the two side streams (lower number = higher priority)")
    ap.add_argument("--cu-masks", default="", help="experiment: 'A,T,F' = CU counts of the audio / text / frames streams (hipExtStreamCreateWithCUMask, "
                    "disjoint bit ranges); needs --no-graph (a replayed multi-stream hipGraph does not keep stream masks)")
    ap.add_argument("--main-tile", type=int, default=-1, help="A/B: GEMM tile configuration of the audio stream's narrow-output GEMMs (-1 library choice)")
    ap.add_argument("--side-tile", type=int, default=-2, help="A/B: GEMM tile configuration of the text / frames blocks (-1 library choice, default = engine's)")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the fp32 / bf16 32-step parity + throughput leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the supplementary configs[3] (V2P) and configs[4] (cascade) legs")
    ap.add_argument("--graph-roofline", action="store_true", help="try to time kernels with events between graph nodes (not available on ROCm 7.0 torch)")
    ap.add_argument("--no-video2roll", action="store_true", help="skip the supplementary Video2Roll frame-encoder measurement (SURVEY 8f N2)")
    ap.add_argument("--no-vocoder", action="store_true", help="skip the supplementary Encodec-decoder measurement (SURVEY 8f N1)")
    ap.add_argument("--video2roll-frames", type=int, default=251, help="video frames per clip: floor(750 / 3) + 1 (x3:1913)")
    ap.add_argument("--stand-in-sampler", action="store_true",
                    help="CPU rehearsal of the N-rank control path (tests/test_dist_gloo.py): no GPU and no HIP library are touched, sample() is a "
                         "stand-in that returns each clip's global index, the collectives run over gloo.  Everything around it -- sharding, warm-up, "
                         "barriers, the timed loop, the MAX over ranks, the all-gather, `scaling_reference` -- is this script's real code.  The line "
                         "says so in `data`; its numbers measure nothing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    standin = args.stand_in_sampler
    import torch.distributed as dist
    if standin:
        dev = torch.device("cpu")
        backend = "gloo"
        torch.cuda.synchronize = lambda *a, **k: None          # (this process never touches a GPU)
    else:
        ndev = torch.cuda.device_count()
        local = local % max(ndev, 1)             # rehearsal of N ranks on fewer devices (V2A_BENCH_BACKEND=gloo)
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        backend = os.environ.get("V2A_BENCH_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import v2a_amd
    from v2a_amd import _lib as L
    from v2a_amd.synth import random_state_dict, synthetic_conditioning

    if standin:
        return standin_main(args, v2a_amd, dist, rank, world)
    if args.gemm_8phase >= 0 or args.gemm_force_tile >= 0 or args.xcd_1x8 or args.gemm_8phase_min_tiles > 0 or args.attn_one_group_from > 0 or args.tuning_reserved:
        L.set_tuning(force_tile=args.gemm_force_tile, eight_phase=(args.gemm_8phase if args.gemm_8phase >= 0 else None),
                     eight_phase_min_tiles=args.gemm_8phase_min_tiles, xcd_order_1x8=args.xcd_1x8,
                     attn_one_group_from=args.attn_one_group_from, reserved=args.tuning_reserved)
    cfg = v2a_amd.DiTConfig()
    if args.clips_per_gpu <= 0:
        args.clips_per_gpu = 1 if world == 1 else 8
    B, T, NC = args.clips_per_gpu, args.frames, 16
    cfm_steps = 64 if args.v2p and args.cfm_steps == 32 else args.cfm_steps
    sd = random_state_dict(cfg, seed=0, device=dev)
    model = v2a_amd.E2TTS(transformer=dict(depth=cfg.depth, dim=cfg.dim, dim_text=cfg.dim_text, heads=cfg.heads, dim_head=cfg.dim_head,
                                           if_text_modules=True, if_cross_attn=True, if_audio_conv=True, if_text_conv=True),
                          num_channels=cfg.num_channels, sampling_rate=24000, if_cond_proj_in=False, tokenizer="phoneme_zh",
                          compute_dtype=args.dtype, device=dev, use_graph=not args.no_graph)
    log("weights generated")
    model.load_state_dict(sd, strict=False)
    del sd
    model.engine().multi_stream = not args.single_stream
    if args.single_stream:
        model.engine().side_tile = -1
        model.engine().main_tile = -1
    elif args.side_tile >= -1:
        model.engine().side_tile = args.side_tile
    if args.main_tile >= 0:
        model.engine().main_tile = args.main_tile
    # (bf16x3 mode: --side-tiles addresses the split-operand tile table: 1 = 64x64, 2 = 128x64, 3 = 128x128 / 8 waves, 4 = 64x128 / 8 waves, 5 = 8-phase)
    for spec, table in ((args.side_tiles, model.engine().split_tiles if args.dtype == "bf16x3" else model.engine().side_tiles), (args.big_tiles, model.engine().split_big_tiles if args.dtype == "bf16x3" else model.engine().big_tiles)):
        for item in filter(None, spec.split(",")):
            key, val = item.split("=")
            st_, op_ = key.split(".")
            table[(st_, op_)] = int(val)
    if args.no_fold_norm:
        model.engine().fold_norm = False
    if args.fold_gemm_all:
        model.engine().fold_gemm_all = True
    if args.no_fuse_skip:
        model.engine().fuse_skip = False
    if args.no_fuse_xattn:
        model.engine().fuse_xattn = False
    model.engine().cross_on_main = args.cross_on_main
    if args.interleave_capture >= 0:
        model.engine().interleave_capture = bool(args.interleave_capture)
    log("weights packed")
    y0, text, roll, ctx, cm = synthetic_conditioning(cfg, B, T, NC, seed=1000 + rank, piano=args.v2p, device=dev)
    cm = cm.cpu()
    cond = torch.empty(B, T, cfg.num_channels, device=dev)      # placeholder, as predict.py:261
    lens = torch.full((B,), T, dtype=torch.long)
    n_clips = B * world

    trace = bool(os.environ.get("V2A_BENCH_TRACE"))      # per-phase wall times on stderr (debug aid; adds host syncs)

    def one_step(steps=cfm_steps):
        ta = time.perf_counter()
        for _ in range(args.cascade):
            out = model.sample(cond, y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll, lens=lens,
                               duration=lens, steps=steps, cfg_strength=args.cfg_strength, remove_parallel_component=False,
                               sway_sampling=True, return_raw_output=True)
        if trace:
            torch.cuda.synchronize()
            tb = time.perf_counter()
        res = v2a_amd.gather_latents(out, n_clips, B)
        if trace:
            torch.cuda.synchronize()
            log("rank %d: sample %.1f ms, gather %.1f ms" % (rank, (tb - ta) * 1e3, (time.perf_counter() - tb) * 1e3))
        return res

    main_ctx = None
    if args.stream_priority:
        # experiment: the audio chain (the graph-capture stream) on a high-priority stream, the side chains on normal / low ones
        import ctypes
        from v2a_amd import dit as _dit
        hip = ctypes.CDLL("libamdhip64.so")
        lo_p, hi_p = ctypes.c_int(), ctypes.c_int()
        assert hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo_p), ctypes.byref(hi_p)) == 0
        pa, ps = (int(v) for v in args.stream_priority.split(","))
        log("stream priority range: least %d, greatest %d; audio %d, sides %d" % (lo_p.value, hi_p.value, pa, ps))
        def mk(prio):
            h = ctypes.c_void_p()
            assert hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, prio) == 0      # 1 = hipStreamNonBlocking
            return torch.cuda.ExternalStream(h.value, device=dev)
        key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
        _dit._STREAMS[key] = (mk(ps), mk(ps), mk(pa))
    if args.cu_masks:
        import ctypes
        from v2a_amd import dit as _dit
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        counts = [int(v) for v in args.cu_masks.split(",")]
        made, lo = [], 0
        for n in counts:
            words = (ctypes.c_uint32 * 8)()
            for b in range(lo, lo + n):
                words[b // 32] |= 1 << (b % 32)
            h = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
            assert rc == 0, rc
            made.append(torch.cuda.ExternalStream(h.value, device=dev))
            lo += n
        key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
        cap = _dit.process_streams(dev)[2]
        _dit._STREAMS[key] = (made[1], made[2], cap)
        main_ctx = made[0]
        log("CU masks: audio %d, text %d, frames %d CUs" % tuple(counts))
        _orig_step = one_step
        def one_step(steps=cfm_steps):            # the audio stream (sample()'s current stream) is the masked one
            main_ctx.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(main_ctx):
                r = _orig_step(steps)
            torch.cuda.current_stream().wait_stream(main_ctx)
            return r
    log("model ready: %s, B=%d/GPU, %d-point grid" % (args.dtype, B, cfm_steps))
    for i in range(args.warmup):
        one_step()
        torch.cuda.synchronize()
        log("warmup %d done" % i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    assert out.shape == (n_clips, T, cfg.num_channels) and bool(torch.isfinite(out).all())
    log("timed %d steps: %.1f ms/step" % (args.steps, el / args.steps * 1e3))

    ms_per_step = el / args.steps * 1e3
    ph = model.phase_ms()
    frames_per_s = n_clips * T * args.cascade / (el / args.steps)
    evals = cfm_steps - 1
    # algorithmic work per forward per clip at this shape (1040.7 GFLOP GEMM + 75.8 attention)
    res = {
        "metric": "mel-frames/sec (10s VGGSound shape, %d-step CFM, CFG %.1f)" % (cfm_steps, args.cfg_strength),
        "value": round(frames_per_s, 2), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[%d]: %d clip(s)/GPU x %d latent frames x 128 ch, %d-point sway grid = %d CFG "
                               "evaluations (%d DiT forwards), T5 context %d tokens, %s"
                               % (3 if args.v2p else (4 if args.cascade > 1 else (1 if B == 1 else 2)), B, T, cfm_steps, evals, 2 * evals, NC,
                                  "V2P roll" if args.v2p else "V2A zero roll"),
                   "clips_per_gpu": B, "global_clips": n_clips, "parallelism": "clip-sharded x%d, 1 all-gather" % world,
                   "hipgraph": not args.no_graph, "schedule": "single stream" if args.single_stream else "three streams",
                   "side_streams": not args.single_stream, "cascade_passes": args.cascade},
        "per_gpu_mel_frames_per_s": round(frames_per_s / world, 2),
        "clips_per_s": round(n_clips / (el / args.steps), 4),
        "ms_per_cfg_evaluation": round(ms_per_step / evals / args.cascade, 4),
        "prepare_ms": round(ph[0], 3), "euler_loop_ms": round(ph[1], 3),     # device-side split of the LAST sample() call of the timed region
    }

    if world > 1:
        # Like-for-like comparator of the scaling curve: the N = 1 default of this script is ONE clip (configs[1]), an N > 1 run
        # samples `clips_per_gpu` clips per GPU (configs[2]) -- a ratio of the two `value`s would mix a shape change into the
        # scaling.  Rank 0 therefore times the SAME per-GPU shape alone (its own shard, no all-gather, the other ranks idle at a
        # barrier) with the line's --steps / --warmup: what `--gpus 1 --clips-per-gpu B` prints as `value`.
        same = None
        if rank == 0:
            def shard_only():
                return model.sample(cond, y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll, lens=lens, duration=lens,
                                    steps=cfm_steps, cfg_strength=args.cfg_strength, remove_parallel_component=False, sway_sampling=True,
                                    return_raw_output=True)
            for _ in range(max(1, args.warmup)):
                shard_only()
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(args.steps):
                for _ in range(args.cascade):
                    shard_only()
            torch.cuda.synchronize()
            same = B * T * args.cascade / ((time.perf_counter() - ts) / args.steps)
        dist.barrier()
        if rank == 0:
            res["scaling_reference"] = scaling_reference(B, same, frames_per_s, world)
    if rank == 0 and world == 1:
        hbm = {}
        if not args.no_batched and B == 1:
            res["batched"] = batched_leg(model, cfg, cfm_steps, args, T, NC, dev)
            if not args.no_roofline:
                r8 = roofline_leg(model, L, args)
                hbm["clips_8"] = r8.pop("hbm")
                res["batched"]["roofline"] = {k: r8[k] for k in ("kernel", "scope", "achieved", "frac", "avg_launch_us", "launches_per_eval", "all_gemm_tflops",
                                                                   "all_gemm_frac", "eval_kernel_ms", "kernels", "frac_mfma_issued", "all_gemm_frac_mfma_issued", "traffic") if k in r8}
            log("batched leg done")
        if not args.no_roofline:
            one_step()                                     # restore this run's plan (and its graph) after the batched leg
            res["roofline"] = roofline_leg(model, L, args, production=not args.single_stream)
            hbm["clips_%d" % B] = res["roofline"].pop("hbm")
            res["roofline"]["hbm"] = hbm
            log("roofline leg done")
        if not args.no_parity_mode and T == 750:
            res["parity_mode"] = parity_mode_leg(v2a_amd, cfg, args, dev)
            log("parity-mode leg done")
        if not args.no_configs and B == 1 and not args.v2p and args.cascade == 1:
            res.update(configs_leg(model, cfg, args, T, NC, dev))
            log("configs[3] / configs[4] legs done")
        if not args.no_video2roll:
            res["video2roll"] = video2roll_leg(L, args, dev, cpu=not args.no_cpu_baseline)
            log("video2roll leg done")
        if not args.no_vocoder:
            res["vocoder"] = vocoder_leg(L, args, dev, T, cpu=not args.no_cpu_baseline)
            log("vocoder leg done")
        if not args.no_cpu_baseline:
            res.update(cpu_baseline_leg(model, cfg, one_step, y0, text, roll, ctx, cm, args, T))
        res.update(summary_fields(res))
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def standin_main(args, v2a_amd, dist, rank, world):
    """--stand-in-sampler: the N-rank control path of main() on CPU tensors over gloo (no GPU, no HIP library).  The sampler is replaced by
    a function that returns, for every clip of the rank's shard, a (T, C) tensor filled with the clip's GLOBAL index; the all-gather must
    then hand every rank the tensor whose clip i is filled with i, whatever the partition (a batch that does not divide by the world size
    is padded and the padding dropped).  Sharding follows v2a_amd.shard_range exactly as cli.py does for a real batch."""
    T, C = 6, 4
    B = args.clips_per_gpu if args.clips_per_gpu > 0 else (1 if world == 1 else 8)
    n_clips = int(os.environ.get("V2A_STANDIN_CLIPS", B * world))        # a batch size that need not divide by the world size
    lo, hi, per = v2a_amd.shard_range(n_clips, rank, world)
    def sample():
        time.sleep(0.002 * (1 + rank % 3))                                # ranks finish at different times: the MAX over ranks matters
        return torch.stack([torch.full((T, C), float(i)) for i in range(lo, hi)]) if hi > lo else torch.zeros(0, T, C)
    def one_step():
        return v2a_amd.gather_latents(sample(), n_clips, per)
    for _ in range(args.warmup):
        one_step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    ok = out.shape == (n_clips, T, C) and all(bool((out[i] == float(i)).all()) for i in range(n_clips))
    frames_per_s = n_clips * T / (el / args.steps)
    res = {"metric": "stand-in (control-path rehearsal, measures nothing)", "value": round(frames_per_s, 2), "unit": "mel-frames/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "none", "data": "stand-in sampler on CPU tensors over gloo: no GPU work",
           "config": {"workload": "stand-in", "clips_per_gpu": per, "global_clips": n_clips, "parallelism": "clip-sharded x%d, 1 all-gather" % world},
           "standin_gather_ok": bool(ok), "standin_shard": [lo, hi, per]}
    if world > 1:
        same = None
        if rank == 0:
            ts = time.perf_counter()
            for _ in range(args.steps):
                sample()
            same = max(hi - lo, 1) * T / ((time.perf_counter() - ts) / args.steps)
        dist.barrier()
        if rank == 0:
            res["scaling_reference"] = scaling_reference(per, same, frames_per_s, world)
    oks = [None] * world
    if world > 1:
        dist.all_gather_object(oks, bool(ok))
        res["standin_gather_ok_all_ranks"] = all(oks)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def summary_fields(res):
    """Short copies of the numbers a reader needs next to `value`: scalars inside `roofline` (the driver's parsed view keeps
    first-level scalars of that object), `parity_qualified` / `n1_8clips_mel_frames_per_s` at top level, and a compact `summary`
    as the LAST key of the line (it survives a truncated tail of stdout)."""
    out, summ = {}, {}
    pm = res.get("parity_mode")
    roof = res.get("roofline")
    head = res["dtype"]
    if pm:
        ok = [(m, pm[m]) for m in ("fp32", "bf16x3", "bf16") if m in pm and pm[m]["meets_1e-3"]]
        hp = pm.get(head, {})
        if ok:
            mode, best = max(ok, key=lambda kv: kv[1]["mel_frames_per_s"])
            out["parity_qualified"] = {"mode": mode, "mel_frames_per_s": best["mel_frames_per_s"], "ms_per_step": best["ms_per_step"],
                                       "max_abs_delta_mel_over_grid": max(best["max_abs_delta_mel_over_grid"], best["max_abs_delta_mel"]),
                                       "steps": best["steps"], "warmup": best["warmup"],
                                       "note": "fastest compute mode whose 32-point sample stays inside north_star's |delta mel| < 1e-3 against the "
                                               "oracle vector; `value` (dtype %s) is %s that tolerance" % (head, "inside" if hp.get("meets_1e-3") else "OUTSIDE")}
            summ["parity_qualified"] = {k: out["parity_qualified"][k] for k in ("mode", "mel_frames_per_s", "ms_per_step", "max_abs_delta_mel_over_grid")}
            summ["headline_mode_max_abs_delta_mel"] = hp.get("max_abs_delta_mel")
        if roof is not None and hp:
            # scalars of the HEADLINE mode's parity (the same compute mode, weights and inputs of the committed oracle vector) and of the
            # fast bf16 mode (outside north_star's tolerance: reported, never `value`), where the driver's parser keeps them
            roof["parity_max_abs_delta_mel_over_grid"] = max(hp["max_abs_delta_mel_over_grid"], hp["max_abs_delta_mel"])
            roof["parity_meets_1e-3"] = bool(hp["meets_1e-3"])
            roof["parity_against"] = ("CPU restatement oracle/e2_cfm_oracle.py (in-tree blocks pinned by the reference's own code; x-transformers / torchdiffeq arithmetic "
                                      "restated from the call sites, unpinned: DESIGN 0; A7 reading: no rotary in cross-attention)")
            roof["parity_fixture_mel_frames_per_s"] = hp["mel_frames_per_s"]
            if "clips8_max_abs_delta_mel" in hp:
                roof["clips8_parity_max_abs_delta_mel"] = hp["clips8_max_abs_delta_mel"]
            fb = pm.get("bf16")
            if fb:
                roof["fast_bf16_mel_frames_per_s"] = fb["mel_frames_per_s"]
                roof["fast_bf16_max_abs_delta_mel"] = max(fb["max_abs_delta_mel_over_grid"], fb["max_abs_delta_mel"])
                if "clips8_mel_frames_per_s" in fb:
                    roof["clips8_fast_bf16_mel_frames_per_s"] = fb["clips8_mel_frames_per_s"]
                    roof["clips8_fast_bf16_max_abs_delta_mel"] = fb["clips8_max_abs_delta_mel"]
    b = res.get("batched")
    if b:
        out["n1_8clips_mel_frames_per_s"] = b["mel_frames_per_s"]      # the N = 1 point of the scaling curve at configs[2]'s per-GPU shape
        summ["batched_8clips"] = {"mode": head, "mel_frames_per_s": b["mel_frames_per_s"], "ms_per_step": b["ms_per_step"]}
        r8 = b.get("roofline")
        if r8:
            fr = {}
            for k, row in r8["kernels"].items():
                if k.startswith("gemm<bf16") and "tflops" in row:
                    cls = k.split(",")[2]
                    key = "qkv_store" if cls == "store" else cls          # (the STORE class with the largest share is the fused QKV projection)
                    best = fr.get(key)
                    if best is None or row["share"] > best[1]:
                        fr[key] = (round(row["tflops"] / PEAK_BF16_TFLOPS, 4), row["share"], round(row.get("tflops_mfma_issued", row["tflops"]) / PEAK_BF16_TFLOPS, 4))
            summ["batched_8clips"].update({"all_gemm_frac": r8["all_gemm_frac"], **{"frac_" + k: v[0] for k, v in fr.items()}})
            if "all_gemm_frac_mfma_issued" in r8:
                summ["batched_8clips"].update({"all_gemm_frac_mfma_issued": r8["all_gemm_frac_mfma_issued"], **{"frac_" + k + "_mfma_issued": v[2] for k, v in fr.items()}})
            # the same classes from the committed kernel-only summary of `--clips-per-gpu 8 --single-stream` (no event packets)
            for k, row in r8["kernels"].items():
                if k.startswith("gemm<bf16") and "tflops" in row and "avg_us" in row:
                    rp = rocprof_avg(k, row["tflops"] * 1e12 * row["avg_us"] * 1e-6, PEAK_BF16_TFLOPS, ROCPROF_STATS_8CLIPS_ALONE)
                    if rp:
                        row["rocprof"] = {"avg_us": rp["avg_us"], "frac": rp["frac"]}
            # HBM / fabric traffic of the 8-clip GEGLU and QKV classes from the committed PMC passes of `--clips-per-gpu 8` (FETCH_SIZE x 2 + WRITE_SIZE per
            # launch) against the algorithmic bytes of the launch: co-running chip-filling kernels do not re-fetch more
            tr8 = {}
            for k, row in r8["kernels"].items():
                if k.startswith("gemm<bf16") and "alg_MB_per_launch" in row and k.split(",")[2] in ("geglu", "store"):
                    t = pmc_traffic(k, row["alg_MB_per_launch"] * 1e6, PMC_FILES_8CLIPS)
                    if t:
                        row["traffic"] = {"MB_per_launch": round(t["bytes_per_launch"] / 1e6, 2), "ratio": t["ratio"]}
                        cls = "qkv_store" if k.split(",")[2] == "store" else "geglu"
                        if cls not in tr8 or row["share"] > tr8[cls][1]:
                            tr8[cls] = (t["ratio"], row["share"])
            if roof is not None:
                roof.update({"clips8_traffic_ratio_" + c: v[0] for c, v in tr8.items()})
                roof.update({"clips8_all_gemm_frac": r8["all_gemm_frac"], "clips8_mel_frames_per_s": b["mel_frames_per_s"], "clips8_mode": head,
                             **{"clips8_frac_" + k: v[0] for k, v in fr.items()}})
                if "all_gemm_frac_mfma_issued" in r8:
                    roof.update({"clips8_all_gemm_frac_mfma_issued": r8["all_gemm_frac_mfma_issued"],
                                 **{"clips8_frac_" + k + "_mfma_issued": v[2] for k, v in fr.items()}})
                if pm and pm.get(head, {}).get("meets_1e-3"):
                    roof["clips8_parity_qualified_mel_frames_per_s"] = b["mel_frames_per_s"]
    h8 = (roof or {}).get("hbm", {}).get("clips_8")
    if h8:
        summ["hbm_clips_8"] = {k: {"live": v["frac"], "rocprof": (v.get("rocprof") or {}).get("frac")} for k, v in h8.items()}
    if summ:
        out["summary"] = summ
    return out


def scaling_reference(clips_per_gpu, n1_same_shape, value, world):
    """The N = 1 comparator of an N-GPU line at the SAME per-GPU shape, and the weak-scaling efficiency against it."""
    return {"clips_per_gpu": clips_per_gpu, "n1_same_shape_mel_frames_per_s": round(n1_same_shape, 2),
            "efficiency": round(value / (world * n1_same_shape), 4),
            "note": "n1_same_shape = one GPU sampling clips_per_gpu clips alone (rank 0, no all-gather, same --steps / --warmup); "
                    "efficiency = value / (n_gpus x n1_same_shape).  The --gpus 1 default line is ONE clip (configs[1]): do not divide by it"}


def vocoder_leg(L, args, dev, T, cpu=True):
    """Supplementary: latents (1, 128, T) -> 24 kHz waveform through the Encodec decoder (SURVEY 8f N1, x3:434-437,
    predict.py:277-278) -- downstream of the timed region of the headline metric, reported beside it.  fp32 throughout."""
    from v2a_amd.encodec import EncodecDecoder
    from v2a_amd.synth import random_encodec_decoder_state_dict
    sd = random_encodec_decoder_state_dict(0)
    dec = EncodecDecoder(sd, dev)
    g = torch.Generator(device="cpu").manual_seed(5)
    emb = torch.randn(1, 128, T, generator=g).to(dev)
    dec.decoder(emb)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        wav = dec.decoder(emb)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    assert wav.shape == (1, 1, 320 * T) and bool(torch.isfinite(wav).all())
    prof = L.KernelProfiler()
    L.set_profiler(prof)
    dec.decoder(emb)
    L.set_profiler(None)
    agg = prof.summary()
    tot_ms = sum(a["ms"] for a in agg.values())
    kern = {}
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        e = {"launches": a["launches"], "ms": round(a["ms"], 3), "share": round(a["ms"] / tot_ms, 4)}
        if a["flops"] > 0:
            e["tflops"] = round(a["flops"] / a["ms"] / 1e9, 3)
        else:
            e["gbs"] = round(a["bytes"] / a["ms"] / 1e6, 1)
        kern[k] = e
    res = {"latent_frames": T, "samples": 320 * T, "ms_per_clip": round(el * 1e3, 2), "audio_seconds_per_s": round(320 * T / 24000 / el, 1),
           "dtype": "fp32", "eager_kernel_ms": round(tot_ms, 2), "kernels": kern,
           "note": "EncodecDecoder.decoder on random latents, seeded weights; outside the timed region of `value`"}
    if cpu:
        from oracle import encodec_oracle as EO
        torch.set_num_threads(host_cores())
        with torch.no_grad():
            t0 = time.perf_counter()
            ref = EO.decoder_forward(sd, emb.cpu())
            cel = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(320 * T / 24000 / cel, 2), "unit": "audio-seconds/s", "cores": host_cores(), "kind": "port",
                               "sample": "the same clip through oracle/encodec_oracle.py (torch fp32)"}
        res["parity_vs_cpu"] = {"max_abs": float((wav.cpu() - ref).abs().max())}
    return res


def video2roll_leg(L, args, dev, cpu=True):
    """Supplementary: the V2P frame encoder (SURVEY 8f N2, x3:1525-1553) on one clip's 251 grey 100x900 frames --
    outside the timed region of the headline metric, reported beside it.  Algorithmic work: 10.36 GFLOP per
    5-frame window (2*MAC over the 21 convolutions of Video2RollNet.resnet18)."""
    from v2a_amd.synth import random_video2roll_state_dict, synthetic_piano_frames
    from v2a_amd.video2roll import Video2RollEngine
    t = args.video2roll_frames
    sd = random_video2roll_state_dict(0)
    eng = Video2RollEngine(sd, dev, compute="fp32" if args.dtype == "bf16x3" else args.dtype)   # the roll encoder has no bf16x3 mode
    x = synthetic_piano_frames(1, t, seed=0).to(dev)
    l = 3 * (t - 1)
    eng.encode_frames(x, l)
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        out = eng.encode_frames(x, l)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    assert out.shape == (1, l, 51) and bool(torch.isfinite(out).all())
    prof = L.KernelProfiler()
    L.set_profiler(prof)
    eng.encode_frames(x, l)
    L.set_profiler(None)
    agg = prof.summary()
    kern = {}
    tot_ms = sum(a["ms"] for a in agg.values())
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        e = {"launches": a["launches"], "ms": round(a["ms"], 3), "share": round(a["ms"] / tot_ms, 4)}
        if a["flops"] > 0:
            e["tflops"] = round(a["flops"] / a["ms"] / 1e9, 2)
        else:
            e["gbs"] = round(a["bytes"] / a["ms"] / 1e6, 1)
        kern[k] = e
    gflop_per_window = 10.36
    res = {"frames": t, "ms_per_clip": round(el * 1e3, 2), "video_frames_per_s": round(t / el, 1),
           "tflops": round(gflop_per_window * t / el / 1e3, 2), "dtype": args.dtype, "eager_kernel_ms": round(tot_ms, 2), "kernels": kern,
           "note": "E2TTS.encode_frames on synthetic frames, seeded weights; outside the timed region of `value`"}
    if cpu:
        from oracle import video2roll_oracle as VO
        torch.set_num_threads(host_cores())
        n = 8
        xc = x[:, :, :n].cpu()
        with torch.no_grad():
            VO.encode_frames(sd, xc[:, :, :2], 6)
            t0 = time.perf_counter()
            ref = VO.encode_frames(sd, xc, 3 * n)
            cel = time.perf_counter() - t0
        got = eng.encode_frames(x[:, :, :n], 3 * n).cpu()
        res["cpu_baseline"] = {"value": round(n / cel, 2), "unit": "video-frames/s", "cores": host_cores(), "kind": "port",
                               "sample": "%d windows of the same clip through oracle/video2roll_oracle.py (torch fp32)" % n}
        res["parity_vs_cpu"] = {"max_abs": float((got - ref).abs().max()), "mean_abs": float((got - ref).abs().mean())}
    return res


def batched_leg(model, cfg, cfm_steps, args, T, NC, dev):
    """Supplementary: the same sampler with 8 clips per GPU (the per-GPU shape of BASELINE.json configs[2]); the GEMMs
    then see M = 12512 rows instead of 1564.  Not the headline `value`."""
    from v2a_amd.synth import synthetic_conditioning
    Bb = 8
    y0, text, roll, ctx, cm = synthetic_conditioning(cfg, Bb, T, NC, seed=77, piano=args.v2p, device=dev)
    cond = torch.empty(Bb, T, cfg.num_channels, device=dev)
    lens = torch.full((Bb,), T, dtype=torch.long)
    def run():
        return model.sample(cond, y0=y0, text_embed=text, context=ctx, context_mask=cm.cpu(), frames_embed=roll, lens=lens, duration=lens,
                            steps=cfm_steps, cfg_strength=args.cfg_strength, remove_parallel_component=False, sway_sampling=True,
                            return_raw_output=True)
    for _ in range(max(1, args.warmup)):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = args.steps
    for _ in range(n):
        out = run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n
    assert bool(torch.isfinite(out).all())
    return {"clips_per_gpu": Bb, "mel_frames_per_s": round(Bb * T / el, 2), "ms_per_step": round(el * 1e3, 2), "clips_per_s": round(Bb / el, 3),
            "steps": n, "warmup": max(1, args.warmup)}


def _timed_evaluation(model, L, args, reps, shapes=False, production=False):
    """Per-kernel time of ONE Euler evaluation of the model's current plan, HIP events recorded on the stream each kernel is
    launched on.  production=False: kernels one at a time on one stream with the library's stand-alone tile choices (isolated
    launch durations).  production=True: the configuration sample() runs -- side streams and the per-(stream, op) tile table --
    so every duration is that of the kernel BESIDE the other two streams' kernels."""
    from v2a_amd.dit import process_streams
    eng = model.engine()
    p = eng.plan
    y = p["y"]
    side = (None, None) if production else (p.pop("st", None), p.pop("sf", None))     # no side streams: isolated launch durations
    side_tile, main_tile = eng.side_tile, eng.main_tile
    if not production:
        eng.side_tile, eng.main_tile = -1, -1         # ... of the library's stand-alone tile choices (as `--single-stream` under rocprofv3)
    keep = y.clone()
    p["step"].zero_()
    eng.euler_step(y, args.cfg_strength, False)       # warm
    torch.cuda.synchronize()
    how = "hip events between the nodes of a single-stream hipGraph (kernel + launch boundary)"
    prof = None
    if args.graph_roofline and not production:          # needs external event nodes in captured graphs: refused by torch on ROCm 7 ("External events
        try:                         # are disallowed in rocm"), and hipEventRecordWithFlags inside a capture invalidates it
            prof = L.KernelProfiler(shapes=shapes, external=True)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=process_streams(model.device)[2], capture_error_mode="thread_local"):
                L.set_profiler(prof)
                try:
                    eng.euler_step(y, args.cfg_strength, False)
                    prof.close()
                finally:
                    L.set_profiler(None)
            for _ in range(reps):
                p["step"].zero_()
                g.replay()
                torch.cuda.synchronize()
                prof.collect()
            del g
        except Exception as e:                          # external event nodes unsupported: eager brackets instead
            log("graph-timed evaluation unavailable (%s): falling back to eager event pairs" % str(e)[:120])
            L.set_profiler(None)
            torch.cuda.synchronize()
            prof = None
    if prof is None:
        # Eager launches, every one between its own pair of HIP events on the launch stream.  The GPU is first parked behind a
        # spin kernel (torch.cuda._sleep) while the host enqueues the whole evaluation, so no interval contains host launch
        # latency: what is measured is kernel + the two event packets around it.
        how = "eager HIP event pairs around every launch, queue pre-filled behind a spin kernel (kernel + event packets)"
        prof = L.KernelProfiler(shapes=shapes, inner=1)
        t0 = time.perf_counter()
        torch.cuda._sleep(10_000_000)
        torch.cuda.synchronize()
        cyc_per_ms = 10_000_000 / ((time.perf_counter() - t0) * 1e3)
        for _ in range(reps):
            p["step"].zero_()
            torch.cuda.synchronize()
            torch.cuda._sleep(int(cyc_per_ms * 12))          # ~12 ms: longer than the host needs to enqueue one evaluation
            L.set_profiler(prof)
            try:
                eng.euler_step(y, args.cfg_strength, False)
            finally:
                L.set_profiler(None)
    agg = prof.summary()
    y.copy_(keep)
    p["step"].zero_()
    if side[0] is not None:
        p["st"], p["sf"] = side
    eng.side_tile, eng.main_tile = side_tile, main_tile
    return agg, how


def _dominant(agg, reps, how, stats_file):
    """The GEMM instantiation with the most time per evaluation in a timed evaluation, as a roofline record.

    `achieved` / `frac` count ALGORITHMIC flops, 2*M*N*K per launch.  A split-operand launch of the bf16x3 mode issues three
    bf16 MFMA products per fp32 product: the profiler counts those 6*M*N*K, reported beside it as `achieved_mfma_issued` /
    `frac_mfma_issued` (what the matrix cores are asked to do, against the same dense bf16 peak)."""
    gem = [(k, a) for k, a in agg.items() if k.startswith("gemm")]
    dom_k, dom = max(gem, key=lambda kv: kv[1]["ms"])
    peak = PEAK_BF16_TFLOPS if "bf16" in dom_k.split(",")[0] else PEAK_F32_TFLOPS
    issue = 3.0 if "a_split" in dom_k else 1.0          # MFMA products per algorithmic product
    issued = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    ach = issued / issue
    alg_of = lambda k, a: a["flops"] / (3.0 if "a_split" in k else 1.0)
    all_alg = sum(alg_of(k, a) for k, a in gem)
    all_iss = sum(a["flops"] for _, a in gem)
    all_ms = sum(a["ms"] for _, a in gem)
    # the committed rocprofv3 summary has one row per kernel INSTANTIATION: every bench class that runs on the dominant class's instantiation
    # (e.g. the frames stream's feed-forward GEMM, keyed `...,tile4>`, on the same 8-phase GEGLU kernel) is in that row's average duration,
    # so the flops set against it are the mean over those classes' launches
    def best_row(k):
        rows = _kernel_rows(stats_file, k, "Name") if stats_file else []
        return max(rows, key=lambda r: float(r["TotalDurationNs"]))["Name"] if rows else None
    try:
        dom_row = best_row(dom_k)
        same = [(k, a) for k, a in gem if k.split(",")[:4] == dom_k.rstrip(">").split(",")[:4] or k == dom_k]
        same = [(k, a) for k, a in same if best_row(k) == dom_row] if dom_row else [(dom_k, dom)]
    except Exception:
        same = [(dom_k, dom)]
    rp_flops = sum(a["flops"] for _, a in same) / issue / max(1, sum(a["launches"] for _, a in same))
    rp = rocprof_avg(dom_k, rp_flops, peak, stats_file)
    if rp and len(same) > 1:
        rp["classes_in_this_row"] = sorted(k for k, _ in same)
        rp["gflop_per_launch_mean"] = round(rp_flops / 1e9, 3)
    rec = {"kernel": dom_k, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
           "timing": how, "rocprof": rp,
           "avg_launch_us": round(dom["ms"] / dom["launches"] * 1e3, 2), "launches_per_eval": dom["launches"] // reps,
           "gflop_per_launch": round(dom["flops"] / issue / dom["launches"] / 1e9, 3),
           "all_gemm_tflops": round(all_alg / (all_ms * 1e-3) / 1e12, 2), "all_gemm_frac": round(all_alg / (all_ms * 1e-3) / 1e12 / peak, 4)}
    if all_iss > all_alg:
        rec.update({"achieved_mfma_issued": round(issued, 2), "frac_mfma_issued": round(issued / peak, 4),
                    "all_gemm_frac_mfma_issued": round(all_iss / (all_ms * 1e-3) / 1e12 / peak, 4),
                    "flops_note": "achieved / frac / all_gemm_* = algorithmic 2MNK; *_mfma_issued = the three bf16 MFMA products the split "
                                  "(bf16x3) kernels run per fp32 product"})
    return dom_k, dom, rec


def roofline_leg(model, L, args, production=False):
    """`achieved` = algorithmic FLOPs (2*M*N*K per launch) / measured time of the GEMM instantiation with the most time per
    evaluation; `hbm` = algorithmic bytes / measured time of the memory-bound kernels against the 8 TB/s HBM peak.  With
    production=True the headline record is that of the configuration sample() runs (three streams, per-(stream, op) tile table:
    every kernel timed beside the other streams' kernels) and `standalone` holds the same measurement with each kernel alone on
    the chip; the same commands under `rocprofv3 --kernel-trace --stats` give the kernel-only durations committed in profiles/."""
    reps = 3
    agg, how = _timed_evaluation(model, L, args, reps)
    if args.shapes:       # per-shape GEMM table on stderr (tuning aid)
        a2, _ = _timed_evaluation(model, L, args, reps, shapes=True)
        tot2 = sum(a["ms"] for a in a2.values())
        for k, a in sorted(a2.items(), key=lambda kv: -kv[1]["ms"]):
            extra = ("%7.1f TF/s" % (a["flops"] / (a["ms"] * 1e-3) / 1e12)) if a["flops"] > 0 and k.startswith("gemm") else ("%7.1f GB/s" % (a["bytes"] / (a["ms"] * 1e-3) / 1e9))
            log("%-58s n/eval=%3d avg=%7.2f us share=%5.1f%% %s" % (k, a["launches"] // reps, a["ms"] / a["launches"] * 1e3, 100 * a["ms"] / tot2, extra))
    hbm = {}
    for k, a in agg.items():
        if k.startswith(("rmsnorm", "dwconv", "cfg_euler", "rope", "linear_small")):
            gbs = a["bytes"] / (a["ms"] * 1e-3) / 1e9
            hbm[k] = {"launches_per_eval": a["launches"] // reps, "avg_us": round(a["ms"] / a["launches"] * 1e3, 2),
                      "MB_per_launch": round(a["bytes"] / a["launches"] / 1e6, 3), "achieved_GBs": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4)}
            # the same kernel in the committed rocprofv3 summary of this shape: kernel-only duration, without the ~2 us of event packets
            # the live figure includes (they are a quarter of a 10 us memory-bound launch)
            # (the live figure times each kernel alone on one stream: the matching summaries are the single-stream ones)
            rp = rocprof_hbm(k, a["bytes"] / a["launches"], ROCPROF_STATS_8CLIPS_ALONE if model.engine().plan["B"] >= 8 else ROCPROF_STATS)
            if rp:
                hbm[k]["rocprof"] = rp

    def table_of(ag):
        table, tot_ms = {}, sum(a["ms"] for a in ag.values())
        for k, a in sorted(ag.items(), key=lambda kv: -kv[1]["ms"]):
            row = {"launches_per_eval": a["launches"] // reps, "avg_us": round(a["ms"] / a["launches"] * 1e3, 2),
                   "share": round(a["ms"] / tot_ms, 4)}
            if a["flops"] > 0 and k.startswith(("gemm", "attention")):
                # algorithmic rate; split-operand GEMMs issue 3 MFMA products per algorithmic product (`tflops_mfma_issued`)
                split_k = k.startswith("gemm") and "a_split" in k
                row["tflops"] = round(a["flops"] / (3.0 if split_k else 1.0) / (a["ms"] * 1e-3) / 1e12, 2)
                if split_k:
                    row["tflops_mfma_issued"] = round(a["flops"] / (a["ms"] * 1e-3) / 1e12, 2)
                if k.startswith("gemm"):
                    row["alg_MB_per_launch"] = round(a["bytes"] / a["launches"] / 1e6, 2)
            else:
                row["gbs"] = round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1)
            table[k] = row
        return table, tot_ms

    dom_k, dom, alone = _dominant(agg, reps, how, ROCPROF_STATS)
    table, tot_ms = table_of(agg)
    alone.update({"scope": "standalone: one stream, every kernel alone on the chip, the library's tile choices",
                  "eval_kernel_ms": round(tot_ms / reps, 3), "kernels": table})
    alg = dom["bytes"] / dom["launches"]
    out = {"bound": "mfma"}
    if production:
        pagg, phow = _timed_evaluation(model, L, args, reps, production=True)
        pk, pdom, prod = _dominant(pagg, reps, phow, ROCPROF_STATS_MULTI)
        ptable, ptot = table_of(pagg)
        prod.update({"scope": "production: three streams and the per-(stream, op) tile table -- each kernel timed beside the other streams' kernels",
                     "sum_of_kernel_ms_per_eval": round(ptot / reps, 3), "kernels": ptable})
        out.update(prod)
        out["standalone"] = alone
        dom_k, alg = pk, pdom["bytes"] / pdom["launches"]
    else:
        out.update(alone)
    # the committed PMC passes were taken in the configuration of the same name: three streams at one clip, or 8 clips per GPU
    out["traffic"] = pmc_traffic(dom_k, alg, PMC_FILES_8CLIPS if model.engine().plan["B"] >= 8 else PMC_FILES)
    out["hbm"] = hbm
    return out


def parity_mode_leg(v2a_amd, cfg, args, dev):
    """configs[1] on the weights / inputs of the committed oracle vector (tests/golden/sample_full.npz: 32-point grid, CFG 2.0):
    for each compute mode the throughput of sample() AND its max |delta mel| against the CPU restatement over the whole
    grid (final latents + every grid point on every 8th frame).  The oracle module is used here as the CHECKER only
    (seeded parameter / input generator of the fixture); nothing of it is timed."""
    import numpy as np
    from oracle import e2_cfm_oracle as O
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "sample_full.npz"), allow_pickle=False))
    P = O.init_params(O.DiTConfig(), 0)
    y0, text, roll, ctx, cm = O.synthetic_inputs(O.DiTConfig(), 1, 750, nc=16, seed=0)
    want, want_traj = torch.from_numpy(g["y_steps32"]), torch.from_numpy(g["traj32_sub"])
    out = {"fixture": "tests/golden/sample_full.npz (oracle, B=1, 32-point sway grid, CFG 2.0; A7 default reading: no rotary in cross-attention)",
           "tolerance_fp32": 1e-3, "timing": "same protocol as the headline: --warmup untimed calls, --steps timed calls"}
    for mode in ("fp32", "bf16x3", "bf16"):
        m = v2a_amd.E2TTS(transformer=dict(depth=cfg.depth, dim=cfg.dim, dim_text=cfg.dim_text, heads=cfg.heads, dim_head=cfg.dim_head,
                                           if_text_modules=True, if_cross_attn=True, if_audio_conv=True, if_text_conv=True),
                          num_channels=cfg.num_channels, sampling_rate=24000, if_cond_proj_in=False, compute_dtype=mode, device=dev)
        m.load_state_dict(P, strict=False)
        kw = dict(y0=y0, text_embed=text, context=ctx, context_mask=cm, frames_embed=roll, steps=32, cfg_strength=2.0,
                  remove_parallel_component=False, sway_sampling=True, return_raw_output=True)
        traj = []
        got = m.sample(torch.zeros(1, 750, 128), trajectory_out=traj, **kw)          # the checked run (captures the graph)
        torch.cuda.synchronize()
        # the line's own protocol: --warmup untimed calls, then exactly --steps timed ones between synchronisations
        for _ in range(args.warmup):
            m.sample(torch.zeros(1, 750, 128), **kw)
        torch.cuda.synchronize()
        n = args.steps
        t0 = time.perf_counter()
        for _ in range(n):
            m.sample(torch.zeros(1, 750, 128), **kw)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / n
        err = (got[0].cpu() - want).abs()
        terr = (torch.stack([t[0, ::8].cpu() for t in traj]) - want_traj).abs().amax(dim=(1, 2))
        out[mode] = {"mel_frames_per_s": round(750 / el, 2), "ms_per_step": round(el * 1e3, 3), "steps": n, "warmup": args.warmup,
                     "max_abs_delta_mel": float("%.3e" % float(err.max())), "mean_abs_delta_mel": float("%.3e" % float(err.mean())),
                     "max_abs_delta_mel_over_grid": float("%.3e" % float(terr.max())), "meets_1e-3": bool(float(err.max()) < 1e-3 and float(terr.max()) < 1e-3)}
        if mode != "fp32" and not args.no_batched:
            # the per-GPU shape of configs[2] in this mode: the SAME clip 8 times over (the launches take their 8-clip tile choices and
            # kernels; every clip must land on the one-clip oracle vector), checked on the first call and timed with the line's protocol
            rep = lambda t: t.repeat(8, *([1] * (t.dim() - 1)))
            kw8 = dict(kw, y0=rep(y0), text_embed=rep(text), context=rep(ctx), context_mask=rep(cm), frames_embed=rep(roll))
            got8 = m.sample(torch.zeros(8, 750, 128), **kw8)
            torch.cuda.synchronize()
            e8 = float((got8.cpu() - want[None]).abs().max())
            for _ in range(max(1, args.warmup) - 1):
                m.sample(torch.zeros(8, 750, 128), **kw8)
            torch.cuda.synchronize()
            n8 = max(2, args.steps // 4)
            t0 = time.perf_counter()
            for _ in range(n8):
                m.sample(torch.zeros(8, 750, 128), **kw8)
            torch.cuda.synchronize()
            el8 = (time.perf_counter() - t0) / n8
            out[mode].update({"clips8_max_abs_delta_mel": float("%.3e" % e8), "clips8_mel_frames_per_s": round(8 * 750 / el8, 2),
                              "clips8_steps": n8, "clips8_meets_1e-3": bool(e8 < 1e-3)})
        del m
        torch.cuda.empty_cache()
    return out


def configs_leg(model, cfg, args, T, NC, dev):
    """Supplementary: BASELINE configs[3] (V2P: non-zero piano roll, the CLI's 64-point grid, src/inference_v2p.py:183) and
    configs[4] (3 cascaded 32-point passes -- defined by this build, the reference has no CoT-guidance code, SURVEY 8d -- at
    the per-GPU share of B=32 on 8 GPUs: 4 clips)."""
    from v2a_amd.synth import synthetic_conditioning
    out = {}
    def run(Bc, steps, piano, passes, seed):
        y0, text, roll, ctx, cm = synthetic_conditioning(cfg, Bc, T, NC, seed=seed, piano=piano, device=dev)
        cond = torch.empty(Bc, T, cfg.num_channels, device=dev)
        lens = torch.full((Bc,), T, dtype=torch.long)
        def once():
            for _ in range(passes):
                o = model.sample(cond, y0=y0, text_embed=text, context=ctx, context_mask=cm.cpu(), frames_embed=roll, lens=lens, duration=lens,
                                 steps=steps, cfg_strength=args.cfg_strength, remove_parallel_component=False, sway_sampling=True, return_raw_output=True)
            return o
        once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 2
        for _ in range(n):
            o = once()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / n
        assert bool(torch.isfinite(o).all())
        step_ctr = int(model.engine().plan["step"].item())
        return el, step_ctr
    el, ctr = run(1, 64, True, 1, 311)
    out["v2p"] = {"workload": "configs[3]: 1 clip, V2P roll, 64-point grid (63 CFG evaluations)", "ms_per_step": round(el * 1e3, 2),
                  "mel_frames_per_s": round(T / el, 2), "step_counter": ctr}
    el, ctr = run(4, 32, False, 3, 312)
    out["cascade"] = {"workload": "configs[4]: 4 clips/GPU x 3 cascaded 32-point passes (synthetic definition)", "ms_per_step": round(el * 1e3, 2),
                      "mel_frames_per_s": round(4 * 3 * T / el, 2), "clips_per_s": round(4 / el, 3)}
    return out


def _kernel_rows(fn, kernel_key, name_col):
    """Rows of a committed rocprofv3 summary that belong to the instantiation(s) of a bench GEMM class `gemm<bf16,a_bf16|a_split,EPI,OUT[,tileN]>`:
    the LDS-DMA ring kernel (split operands: its S3 instantiations), or the 256x256 8-phase kernel (plain: `tile6`; split: `tile4` = tile_hint 5, which
    runs the same instantiations on three K passes).  rocprofv3 prints these names demangled, half demangled (`__bf16` comes out as
    `bool _Accum`) or mangled, depending on the instantiation."""
    import csv
    epi = {"store": "0", "sigmoid": "1", "geglu": "2", "resid": "3", "gate_resid": "4"}
    parts = kernel_key[kernel_key.index("<") + 1:-1].split(",")
    split = parts[1] == "a_split"
    e, bf_out = epi[parts[2]], parts[3].startswith("bf16")
    tiles = [x for x in parts[4:] if x.startswith("tile")]
    on8 = ("tile4" in tiles) if split else ("tile6" in tiles)
    bases = ("gemm_bf16_8ph_kernel",) if on8 else (("gemm_bf16_dma_kernel",) if tiles else ("gemm_bf16_dma_kernel", "gemm_bf16_8ph_kernel"))
    out = []
    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", fn))):
        k = r[name_col].replace("void ", "").replace("(anonymous namespace)::", "")
        base = next((b_ for b_ in bases if b_ in k), None)
        if base is None:
            continue
        t = k[k.find(base) + len(base):]
        if t.startswith("<"):                        # (half) demangled: <EPI, OutT, BM, BN, WGM, WGN, NST, S3>
            f = [x.strip() for x in t[1:t.find(">")].split(",")]
            ok = f[0] == e and ((f[1] in ("__bf16", "bool _Accum")) if bf_out else f[1] == "float")
            if base == "gemm_bf16_dma_kernel" and len(f) >= 8:
                ok = ok and (f[7] == "true") == split
        else:                                        # mangled: ILi{EPI}E{f | DF16b}Li{BM}E...Lb{S3}E
            ok = t.startswith("ILi%sE%s" % (e, "DF16b" if bf_out else "f"))
            if base == "gemm_bf16_dma_kernel":
                ok = ok and ("Lb1E" in t[:60]) == split
        if ok:
            out.append(r)
    return out


def rocprof_avg(kernel_key, flops_per_launch, peak, stats_file=None):
    """The same kernel class in the committed `rocprofv3 --kernel-trace --stats` summary of `bench.py --single-stream`
    (profiles/): kernel-only duration, without the event packets the live figure includes."""
    try:
        stats_file = stats_file or ROCPROF_STATS
        rows = _kernel_rows(stats_file, kernel_key, "Name")
        if not rows:
            return None
        best = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        us = float(best["AverageNs"]) / 1e3
        tf = flops_per_launch / (us * 1e-6) / 1e12
        return {"avg_us": round(us, 2), "calls": int(best["Calls"]), "achieved": round(tf, 2), "frac": round(tf / peak, 4),
                "source": "profiles/%s (%s)" % (stats_file, best["Name"].replace("void (anonymous namespace)::", "")[:70])}
    except Exception:
        return None


_HBM_KERNELS = {"rmsnorm": ("rmsnorm_kernel",), "dwconv+norm": ("dwconv_kernel<31, 4, true>", "dwconv_stream_kernel<31, true", "dwconv_group_kernel<31, 4, true>"),
                "dwconv": ("dwconv_kernel<31, 4, false>", "dwconv_kernel<31, 6, false>", "dwconv_stream_kernel<31, false"), "cfg_euler": ("cfg_euler_kernel",),
                "linear_small": ("linear_small_kernel",), "rope": ("rope_kernel",)}


def rocprof_hbm(key, bytes_per_launch, stats_file):
    """A memory-bound kernel class of the bench in a committed `rocprofv3 --kernel-trace --stats` summary: calls, average kernel-only
    duration, and the algorithmic bytes per launch over that duration against the 8 TB/s HBM peak."""
    try:
        import csv
        names = next(v for k, v in _HBM_KERNELS.items() if key.split("<")[0] == k)
        calls, ns = 0, 0.0
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", stats_file))):
            if any(n in r["Name"] for n in names):
                calls += int(r["Calls"])
                ns += float(r["TotalDurationNs"])
        if not calls:
            return None
        us = ns / calls / 1e3
        gbs = bytes_per_launch / (us * 1e-6) / 1e9
        return {"avg_us": round(us, 2), "calls": calls, "achieved_GBs": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4), "source": "profiles/" + stats_file}
    except Exception:
        return None


def pmc_traffic(kernel_key, algorithmic_bytes=None, files=None):
    """HBM/fabric bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/ files named in PMC_FILES; FETCH_SIZE and WRITE_SIZE collected in separate passes, KB units,
    FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md).  PMC collection cannot run inside
    the timed bench, so this is the recorded figure for the same kernel instantiation, or None."""
    try:
        def pick(fn):
            rs = _kernel_rows(fn, kernel_key, "kernel")
            return max(rs, key=lambda r: int(r["dispatches"])) if rs else None
        files = files or PMC_FILES
        f, w = pick(files[0]), pick(files[1])
        if f is None or w is None:
            return None
        fetch = float(f["FETCH_SIZE"]) * 1024 * 2 / int(f["dispatches"])
        write = float(w["WRITE_SIZE"]) * 1024 / int(w["dispatches"])
        out = {"bytes_per_launch": round(fetch) + round(write), "fetch_bytes": round(fetch), "write_bytes": round(write),
               "source": "profiles/%s + %s (rocprofv3 --pmc, separate passes; FETCH_SIZE x2)" % tuple(files)}
        if algorithmic_bytes:
            # operands once (A, W), the residual row read, the fp32 result and its bf16 shadow written: what one launch must move
            out["algorithmic_bytes"] = round(algorithmic_bytes)
            out["ratio"] = round((fetch + write) / algorithmic_bytes, 3)
        return out
    except Exception:
        return None


def cpu_baseline_leg(model, cfg, one_step, y0, text, roll, ctx, cm, args, T):
    """CPU restatement of the reference (oracle/, kind "port"; the reference itself cannot be
    imported here) on a bounded sample: B=1, `cpu-baseline-steps` grid points, all host
    cores, linearly extrapolated to the 31 evaluations of the headline (stated in `sample`)."""
    from oracle import e2_cfm_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    log("cpu baseline: oracle on %d threads, %d-point grid" % (cores, args.cpu_baseline_steps))
    P = model.state_dict()
    ocfg = O.DiTConfig()
    s = args.cpu_baseline_steps
    a = [t[:1].float().cpu() for t in (y0, text, roll, ctx)]
    runs = []
    for i in range(4):                  # one warm-up, median of three
        t0 = time.perf_counter()
        ref = O.sample(P, ocfg, a[0], a[1], a[2], a[3], cm[:1], steps=s, cfg_strength=args.cfg_strength, remove_parallel_component=False)
        if i > 0:
            runs.append(time.perf_counter() - t0)
    cpu_s = sorted(runs)[1]
    per_eval = cpu_s / (s - 1)
    full = per_eval * (32 - 1)
    # parity of the measured (bf16 or fp32) GPU path on the same bounded sample
    got = one_step(steps=s)[:1].float().cpu()
    err = (got - ref).abs()
    return {"cpu_baseline": {"value": round(T / full, 3), "unit": "mel-frames/s", "cores": torch.get_num_threads(), "kind": "port",
                             "sample": "B=1, steps=%d (%d of 31 CFG evaluations = %d forwards; one warm-up run, median of three: %.1f s CPU, runs %s), "
                                       "extrapolated linearly in evaluations; CPU restatement of the reference in plain torch fp32"
                                       % (s, s - 1, 2 * (s - 1), cpu_s, " / ".join("%.1f" % r for r in runs)),
                             "s_per_cfg_evaluation": round(per_eval, 3)},
            "parity": {"mode": args.dtype, "sample": "same steps=%d run, B=1 (the 32-point grid is in parity_mode)" % s, "max_abs_delta_mel": round(float(err.max()), 6),
                       "mean_abs_delta_mel": round(float(err.mean()), 7)}}


if __name__ == "__main__":
    main()
