#!/usr/bin/env python3
"""How much of the chip does one evaluation need?  From a rocprofv3 kernel trace of `bench.py --single-stream` (every kernel alone on the chip)
the CU-time of each kernel is estimated as  duration x min(1, workgroups / (256 CUs x workgroups that fit one CU))  -- a launch that cannot fill the
chip leaves the rest to the other queues in the three-stream sampler -- and summed per kernel class over one evaluation.  The sum / 256 is the
wall time the evaluation would need if the three queues shared the chip perfectly; beside it the measured wall of the three-stream trace.
usage: python scripts/cu_time.py <single-stream trace dir> [<three-stream trace dir>]"""
import collections, csv, glob, re, sys


def rows_of(d):
    f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def one_eval(rows):
    end = [i for i, r in enumerate(rows) if "cfg_euler" in r["Kernel_Name"]]
    a, b = end[-4], end[-3]
    return rows[a + 1:b + 1]


def num(r, *names):
    for k in r:
        if any(n.lower() == k.lower() for n in names):
            try:
                return int(r[k])
            except ValueError:
                pass
    return 0


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)
    m = re.search(r"(gemm_bf16_dma_kernel<[^>]*>|gemm_bf16_dma_kernelI\w+?E{2}|gemm_bf16_8ph_kernel<[^>]*>|gemm_bf16_8ph_kernelI\w+?E{2}|attn_mfma\w*<[^>]*>|qproj_xattn\w*|dwconv\w*|rmsnorm\w*|cfg_euler|linear_small|split_bf16)", n)
    return (m.group(1) if m else n)[:60]


def dyn_lds(name):
    """Dynamic LDS of this library's kernels by instantiation (the trace's LDS_Block_Size holds the static part only)."""
    n = re.sub(r"\(anonymous namespace\)::|void ", "", name)
    if "gemm_bf16_8ph_kernel" in n:
        return 2 * 4 * 128 * 128 + 1024
    m = re.search(r"gemm_bf16_dma_kernel<(\d+), [^,]+, (\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (\d+)>", n)
    if not m:
        m2 = re.search(r"gemm_bf16_dma_kernelILi(\d+)E(?:f|DF16b)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELi(\d+)E", n)
        if m2:
            g = m2.groups()
            m = type("M", (), {"groups": lambda self: (g[0], g[1], g[2], g[3], g[4], g[5], "true" if g[6] == "1" else "false", g[7])})()
    if m:
        _, bm, bn, _, _, nst, s3, bk = m.groups()
        return int(nst) * (int(bm) + int(bn)) * int(bk) * 2 * (2 if s3 == "true" else 1) + int(bm) * 4 + 16
    m = re.search(r"attn_mfma_split_kernel<(\d+)", n)
    if m:
        return int(m.group(1)) * 2 * 2 * (64 * 64 + 64 * 64) * 2
    if "qproj_xattn_kernel" in n:
        return 100 * 1024 if "true" in n else 52 * 1024
    if "dwconv_stream_kernel" in n:
        return 135 * 1024
    return 0


ev = one_eval(rows_of(sys.argv[1]))
agg = collections.OrderedDict()
tot_t = tot_cu = 0.0
for r in ev:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    grid = max(1, num(r, "Grid_Size", "Grid_Size_X")) * max(1, num(r, "Grid_Size_Y")) * max(1, num(r, "Grid_Size_Z"))
    wg = max(1, num(r, "Workgroup_Size", "Workgroup_Size_X")) * max(1, num(r, "Workgroup_Size_Y")) * max(1, num(r, "Workgroup_Size_Z"))
    nwg = max(1, grid // wg)
    lds = num(r, "LDS_Block_Size", "LDS_Block_Size_v") + dyn_lds(r["Kernel_Name"])
    vg = num(r, "VGPR_Count") + num(r, "Accum_VGPR_Count")
    waves = (wg + 63) // 64
    per_simd = max(1, min(8, 512 // max(8, (vg + 7) // 8 * 8)))
    fit = max(1, min(160 * 1024 // lds if lds else 16, per_simd * 4 // waves if waves <= per_simd * 4 else 1, 32 // waves if waves <= 32 else 1))
    share = min(1.0, nwg / (256.0 * fit))
    k = short(r["Kernel_Name"])
    a = agg.setdefault(k, [0, 0.0, 0.0, 0, 0, 0])
    a[0] += 1
    a[1] += dur
    a[2] += dur * share
    a[3], a[4], a[5] = nwg, fit, lds
    tot_t += dur
    tot_cu += dur * share
print("one evaluation, every kernel alone on the chip: %d kernels, sum of durations %.1f us, chip-time demand %.1f us (= sum of duration x share of the chip)" % (len(ev), tot_t, tot_cu))
print("%-62s %5s %9s %9s %6s  (last launch: workgroups, fit per CU, LDS bytes)" % ("kernel class", "n", "sum us", "chip us", "share"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print("%-62s %5d %9.1f %9.1f %5.1f%%  (%d, %d, %d)" % (k, a[0], a[1], a[2], 100 * a[2] / tot_cu, a[3], a[4], a[5]))
if len(sys.argv) > 2:
    ev3 = one_eval(rows_of(sys.argv[2]))
    wall = (int(ev3[-1]["End_Timestamp"]) - int(ev3[0]["Start_Timestamp"])) / 1e3
    print("three-stream evaluation wall %.1f us: %.0f %% of the single-stream sum, %.0f %% above the chip-time demand" % (wall, 100 * wall / tot_t, 100 * (wall / tot_cu - 1)))
