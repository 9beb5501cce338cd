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
    for fn in (*bench.PMC_FILES, bench.ROCPROF_STATS, bench.ROCPROF_STATS_MULTI):
        assert os.path.isfile(os.path.join(ROOT, "profiles", fn)), fn


@pytest.mark.parametrize("key,stats", [("gemm<bf16,a_bf16,resid,f32,tile12>", "ROCPROF_STATS_MULTI"), ("gemm<bf16,a_bf16,resid,f32>", "ROCPROF_STATS"),
                                       ("gemm<bf16,a_bf16,gate_resid,f32,tile14>", "ROCPROF_STATS_MULTI")])
def test_rocprof_block_finds_the_dominant_kernel_classes(bench, key, stats):
    r = bench.rocprof_avg(key, 6.7e9, 2.5e15 / 1e12, getattr(bench, stats))
    assert r is not None and r["calls"] > 100 and 5.0 < r["avg_us"] < 200.0 and "gemm_bf16_dma_kernel" in r["source"], r


def test_traffic_block_reports_algorithmic_bytes_and_ratio(bench):
    t = bench.pmc_traffic("gemm<bf16,a_bf16,resid,f32,tile12>", 23.9e6)
    assert t is not None and t["bytes_per_launch"] == t["fetch_bytes"] + t["write_bytes"]
    assert abs(t["ratio"] - t["bytes_per_launch"] / 23.9e6) < 1e-2 and 1.0 < t["ratio"] < 4.0, t


def test_bench_help_lists_the_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    for flag in ("--gpus", "--steps", "--warmup", "--dtype", "--clips-per-gpu", "--no-cpu-baseline"):
        assert flag in out.stdout, flag
