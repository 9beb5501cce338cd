"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (runs on the GPU box; writes a small CSV)."""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"(gemm_bf16_dma_kernel<[^>]*>|gemm_bf16_8ph_kernel<[^>]*>|gemm_kernel<[^>]*>|attn_\w+|dwconv_kernel|rmsnorm_kernel<[^>]*>|rope_kernel|cfg_euler\w*|"
                  r"linear_small\w*|time_cond\w*|cast_bf16\w*|step_advance\w*|apg_\w+|fill_registers\w*)", n)
    if m:
        return m.group(1)
    m = re.search(r"N_1\d+(gemm_bf16_dma_kernel|gemm_bf16_8ph_kernel|gemm_kernel|rmsnorm_kernel|rope_kernel)I(\w+?)EEv", n)
    if m:
        return m.group(1) + "<" + m.group(2) + ">"
    return n[:70]


def main(d, out):
    files = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    grid = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    names = sorted({c for k in agg for c in agg[k]})
    with open(out, "w") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "dispatches"] + names)
        for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
            w.writerow([k, len(disp[k])] + [agg[k].get(c, 0.0) for c in names])
    print("wrote", out, len(agg), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
