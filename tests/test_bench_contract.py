"""bench.py reads committed profile summaries (profiles/) for the `roofline.rocprof` and `roofline.traffic` blocks of its JSON line and
swallows a missing file or an unmatched kernel name (the block becomes null).  These CPU tests keep the file names, the kernel-name
matching and the argument parser in step with what is committed."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)               # main() is guarded by __name__
    return m


def test_profile_files_named_by_bench_exist(bench):
    for fn in (*bench.PMC_FILES, *bench.PMC_FILES_8CLIPS, bench.ROCPROF_STATS, bench.ROCPROF_STATS_MULTI, bench.ROCPROF_STATS_8CLIPS, bench.ROCPROF_STATS_8CLIPS_ALONE):
        assert os.path.isfile(os.path.join(ROOT, "profiles", fn)), fn


@pytest.mark.parametrize("key,stats", [("gemm<bf16,a_split,resid,f32>", "ROCPROF_STATS_MULTI"), ("gemm<bf16,a_split,resid,f32>", "ROCPROF_STATS"),
                                       ("gemm<bf16,a_split,gate_resid,f32>", "ROCPROF_STATS_MULTI"), ("gemm<bf16,a_split,geglu,bf16>", "ROCPROF_STATS_MULTI"),
                                       ("gemm<bf16,a_split,store,f32,tile4>", "ROCPROF_STATS_MULTI"), ("gemm<bf16,a_split,resid,f32,tile5>", "ROCPROF_STATS_MULTI"),
                                       ("gemm<bf16,a_split,geglu,bf16>", "ROCPROF_STATS_8CLIPS"), ("gemm<bf16,a_split,resid,f32>", "ROCPROF_STATS_8CLIPS_ALONE")])
def test_rocprof_block_finds_the_dominant_kernel_classes(bench, key, stats):
    """The GEMM classes of the headline (bf16x3) mode in the committed summaries: split-operand ring instantiations (S3 = true), the 8-phase
    kernel for `tile4` (split tile_hint 5) and by-shape launches, the 128x256 ring on 32-wide K stages for `tile5` (split tile_hint 6: the frames stream's narrow GEMMs)."""
    r = bench.rocprof_avg(key, 6.7e9, 2.5e15 / 1e12, getattr(bench, stats))
    assert r is not None and r["calls"] > 100 and 5.0 < r["avg_us"] < 1500.0 and ("gemm_bf16_dma_kernel" in r["source"] or "gemm_bf16_8ph_kernel" in r["source"]), r
    if "tile4" in key:
        assert "8ph" in r["source"]
    if "tile5" in key:
        assert "dma_kernel" in r["source"]


def test_traffic_block_reports_algorithmic_bytes_and_ratio(bench):
    t = bench.pmc_traffic("gemm<bf16,a_split,resid,f32>", 45e6)
    assert t is not None and t["bytes_per_launch"] == t["fetch_bytes"] + t["write_bytes"]
    assert abs(t["ratio"] - t["bytes_per_launch"] / 45e6) < 1e-2 and 1.0 < t["ratio"] < 6.0, t
    t8 = bench.pmc_traffic("gemm<bf16,a_split,geglu,bf16>", 300e6, bench.PMC_FILES_8CLIPS)
    assert t8 is not None and "8clips" in t8["source"] and 1.0 < t8["ratio"] < 10.0, t8


def test_bench_help_lists_the_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    for flag in ("--gpus", "--steps", "--warmup", "--dtype", "--clips-per-gpu", "--no-cpu-baseline"):
        assert flag in out.stdout, flag


def test_scaling_reference_field(bench):
    """An N-GPU line carries the N = 1 comparator at the SAME per-GPU shape and the efficiency against it (the --gpus 1 default is one
    clip, an N > 1 run 8 clips per GPU: a ratio of the two `value`s is not a scaling number)."""
    r = bench.scaling_reference(8, 8800.0, 8 * 8800.0 * 0.9, 8)
    assert r["clips_per_gpu"] == 8 and r["n1_same_shape_mel_frames_per_s"] == 8800.0 and abs(r["efficiency"] - 0.9) < 1e-4


def test_summary_fields_pick_the_parity_qualified_mode(bench):
    mk = lambda v, e, ok, **x: {"mel_frames_per_s": v, "ms_per_step": 750e3 / v, "steps": 20, "warmup": 5, "max_abs_delta_mel": e,
                                "max_abs_delta_mel_over_grid": e, "meets_1e-3": ok, **x}
    pm = {"fp32": mk(987.0, 9e-6, True), "bf16x3": mk(2799.0, 9e-5, True, clips8_max_abs_delta_mel=9.5e-5, clips8_mel_frames_per_s=3200.0),
          "bf16": mk(5757.0, 0.049, False, clips8_max_abs_delta_mel=0.05, clips8_mel_frames_per_s=8300.0)}
    batched = {"mel_frames_per_s": 3210.0, "ms_per_step": 1869.0,
               "roofline": {"all_gemm_frac": 0.12, "all_gemm_frac_mfma_issued": 0.36,
                            "kernels": {"gemm<bf16,a_split,geglu,bf16>": {"tflops": 380.0, "tflops_mfma_issued": 1140.0, "share": 0.3},
                                        "gemm<bf16,a_split,store,f32,tile4>": {"tflops": 195.0, "tflops_mfma_issued": 585.0, "share": 0.1}}}}
    # the headline is the parity-qualified mode: its parity and the fast bf16 mode's figures travel as scalars inside `roofline`
    res = {"dtype": "bf16x3", "parity_mode": pm, "batched": batched, "roofline": {"hbm": {"clips_8": {"rmsnorm": {"frac": 0.495}}}}}
    out = bench.summary_fields(res)
    pq, roof = out["parity_qualified"], res["roofline"]
    assert pq["mode"] == "bf16x3" and pq["mel_frames_per_s"] == 2799.0 and pq["max_abs_delta_mel_over_grid"] < 1e-3 and "inside" in pq["note"]
    assert roof["parity_max_abs_delta_mel_over_grid"] == 9e-5 and roof["parity_meets_1e-3"] is True and roof["clips8_parity_max_abs_delta_mel"] == 9.5e-5
    assert roof["fast_bf16_mel_frames_per_s"] == 5757.0 and roof["fast_bf16_max_abs_delta_mel"] == 0.049 and roof["clips8_fast_bf16_mel_frames_per_s"] == 8300.0
    assert roof["clips8_parity_qualified_mel_frames_per_s"] == 3210.0 and roof["clips8_mode"] == "bf16x3"
    assert out["n1_8clips_mel_frames_per_s"] == 3210.0
    assert list(out)[-1] == "summary" and out["summary"]["batched_8clips"]["frac_geglu"] == round(380.0 / 2500.0, 4)
    assert out["summary"]["batched_8clips"]["frac_geglu_mfma_issued"] == round(1140.0 / 2500.0, 4)
    assert roof["clips8_frac_qkv_store"] == round(195.0 / 2500.0, 4) and roof["clips8_frac_qkv_store_mfma_issued"] == round(585.0 / 2500.0, 4)
    assert out["summary"]["hbm_clips_8"] == {"rmsnorm": {"live": 0.495, "rocprof": None}}
    # a bf16 headline (--dtype bf16) is flagged as OUTSIDE the tolerance and gets no parity-qualified 8-clip scalar
    res2 = {"dtype": "bf16", "parity_mode": pm, "batched": dict(batched), "roofline": {}}
    out2 = bench.summary_fields(res2)
    assert "OUTSIDE" in out2["parity_qualified"]["note"] and res2["roofline"]["parity_meets_1e-3"] is False
    assert "clips8_parity_qualified_mel_frames_per_s" not in res2["roofline"]


def test_dominant_reports_algorithmic_and_issued_fractions(bench):
    """Split-operand (bf16x3) GEMM launches are profiled with the 6MNK flops the matrix cores run; the roofline's `achieved` / `frac` are
    algorithmic (2MNK) and the issued figures sit beside them."""
    agg = {"gemm<bf16,a_split,resid,f32>": {"launches": 30, "ms": 3.0, "flops": 30 * 6.0e9 * 3, "bytes": 1e9},
           "gemm<f32,a_f32,store,f32>": {"launches": 3, "ms": 0.1, "flops": 1e9, "bytes": 1e6}, "attention<bf16x3>": {"launches": 3, "ms": 9.0, "flops": 1e9, "bytes": 1}}
    k, dom, rec = bench._dominant(agg, 3, "test", bench.ROCPROF_STATS)
    assert k == "gemm<bf16,a_split,resid,f32>" and rec["launches_per_eval"] == 10 and rec["gflop_per_launch"] == 6.0
    assert abs(rec["achieved"] - 60.0) < 1e-6 and abs(rec["achieved_mfma_issued"] - 180.0) < 1e-6
    assert abs(rec["frac"] * 3 - rec["frac_mfma_issued"]) < 2e-4 and rec["all_gemm_frac_mfma_issued"] > rec["all_gemm_frac"]


def test_rocprof_hbm_block_finds_the_memory_bound_kernels(bench):
    for key, mb in (("rmsnorm<bf16x2>", 100.0), ("dwconv+norm", 120.0), ("cfg_euler", 12.3)):
        r = bench.rocprof_hbm(key, mb * 1e6, bench.ROCPROF_STATS_8CLIPS_ALONE)
        assert r is not None and r["calls"] > 10 and 1.0 < r["avg_us"] < 200.0, (key, r)
