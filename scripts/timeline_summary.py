"""Summarise a rocprofv3 kernel trace of bench.py (runs on the GPU box): per-evaluation wall, per-queue busy time,
and the largest idle gaps of each queue with the kernels around them."""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adv = [i for i, r in enumerate(rows) if "step_advance" in r["Kernel_Name"]]
a, b = adv[-4], adv[-3]
ev = rows[a + 1:b + 1]
t0, t1 = int(ev[0]["Start_Timestamp"]), int(ev[-1]["End_Timestamp"])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"(gemm_bf16_dma_kernel<[^>]*>|gemm_bf16_dma_kernelI\w+?E{2}|gemm_bf16_8ph_kernel<[^>]*>|gemm_bf16_8ph_kernelI\w+?E{2}|attn_mfma\w*<[^>]*>|dwconv|rmsnorm|rope|cfg_euler|linear_small|step_adv|cast)", n)
    return (m.group(1) if m else n)[:44]
print("eval wall us %.1f  kernels %d" % ((t1 - t0) / 1e3, len(ev)))
q = collections.defaultdict(list)
for r in ev:
    q[r["Queue_Id"]].append(r)
for k, rs in q.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print("queue %s: %d kernels, busy %.1f us (%.0f%%)" % (k, len(rs), busy / 1e3, 100 * busy / (t1 - t0)))
    gaps = []
    for x, y in zip(rs, rs[1:]):
        g = int(y["Start_Timestamp"]) - int(x["End_Timestamp"])
        gaps.append((g, short(x["Kernel_Name"]), short(y["Kernel_Name"])))
    tot = sum(g for g, _, _ in gaps if g > 0)
    print("   idle between its kernels: %.1f us; largest gaps:" % (tot / 1e3))
    for g, x, y in sorted(gaps, reverse=True)[:6]:
        print("      %.1f us  after %-44s before %s" % (g / 1e3, x, y))

# ---- how many kernels run at once, and what one layer looks like (audio = the queue with the most kernels)
evs = []
for r in ev:
    evs.append((int(r["Start_Timestamp"]), 1))
    evs.append((int(r["End_Timestamp"]), -1))
evs.sort()
hist = collections.Counter()
cur, last = 0, evs[0][0]
for t, d in evs:
    hist[cur] += t - last
    last = t
    cur += d
tot = sum(hist.values())
print("concurrency: " + "  ".join("%d kernels %.1f%%" % (k, 100 * v / tot) for k, v in sorted(hist.items())))
main_q = max(q, key=lambda k: len(q[k]))
names = {k: ("audio" if k == main_q else "side%d" % i) for i, k in enumerate(q)}
mid = ev[len(ev) // 2 - 20: len(ev) // 2 + 30]
base = int(mid[0]["Start_Timestamp"])
print("timeline of ~1.5 layers in the middle of the evaluation (us from its first kernel):")
for r in mid:
    s, e = int(r["Start_Timestamp"]) - base, int(r["End_Timestamp"]) - base
    print("  %-6s %8.1f -> %8.1f  (%6.1f)  %s  grid %s" % (names[r["Queue_Id"]], s / 1e3, e / 1e3, (e - s) / 1e3, short(r["Kernel_Name"]), r.get("Grid_Size", "")))
