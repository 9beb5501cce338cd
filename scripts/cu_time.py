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


ev = one_eval(rows_of(sys.argv[1]))
agg = collections.OrderedDict()
tot_t = tot_cu = 0.0
for r in ev:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    grid = max(1, num(r, "Grid_Size", "Grid_Size_X")) * max(1, num(r, "Grid_Size_Y")) * max(1, num(r, "Grid_Size_Z"))
    wg = max(1, num(r, "Workgroup_Size", "Workgroup_Size_X")) * max(1, num(r, "Workgroup_Size_Y")) * max(1, num(r, "Workgroup_Size_Z"))
    nwg = max(1, grid // wg)
    lds = num(r, "LDS_Block_Size", "LDS_Block_Size_v")
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
