"""How much of a pipelined step does the statistics / exchange tail occupy?   (rocprofv3 --kernel-trace CSV of a bench.py run)

  python tools/tail_report.py <kernel_trace.csv> [--shards S]

Kernels are split into PATH (mc_paths_kernel / mc_sweep_*) and TAIL (scan, hist, final, stats, the exchange kernel sum_u64, the
record copies and fills).  Over the steady part of the run (the middle 80 % of the path launches of the main kernel variant):
  step          = window / batches                       (batches = path launches / S)
  tail busy     = length of the UNION of the tail kernels' intervals / window   ("stats + exchange stream busy")
  per kernel    = launches per batch, mean and max duration
  path overlap  = fraction of the window in which two or more path kernels run at once
"""
import collections
import csv
import sys

path = sys.argv[1]
S = int(sys.argv[sys.argv.index("--shards") + 1]) if "--shards" in sys.argv else 1
rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("mcp::", "")
    return n[:60]


is_path = lambda n: "mc_paths_kernel" in n or "mc_sweep" in n
TAIL_KEYS = ("scan_kernel", "hist_kernel", "final_kernel", "stats_kernel", "sum_u64_kernel", "pass0_kernel", "copyBuffer", "FillFunctor", "zero_u64")
is_tail = lambda n: any(k in n for k in TAIL_KEYS)
variants = collections.Counter(short(r["Kernel_Name"]) for r in rows if is_path(r["Kernel_Name"]))
main = variants.most_common(1)[0][0]
P = [r for r in rows if short(r["Kernel_Name"]) == main]
lo, hi = int(len(P) * 0.1), int(len(P) * 0.9)
t0, t1 = int(P[lo]["Start_Timestamp"]), int(P[hi]["Start_Timestamp"])
window = t1 - t0
batches = (hi - lo) / S
inwin = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


tail = [(int(r["Start_Timestamp"]), min(int(r["End_Timestamp"]), t1)) for r in inwin if is_tail(r["Kernel_Name"])]
pth = [(int(r["Start_Timestamp"]), min(int(r["End_Timestamp"]), t1)) for r in inwin if is_path(r["Kernel_Name"])]
ev = sorted([(s, 1) for s, e in pth] + [(e, -1) for s, e in pth])
depth, last, over = 0, t0, 0
for t, d in ev:
    if depth >= 2:
        over += t - last
    depth += d
    last = t
print(f"{path}")
print(f"main path kernel: {main}; {hi - lo} launches in the window = {batches:.0f} batches of {S} shard(s); window {window / 1e6:.2f} ms")
print(f"step {window / batches / 1e3:8.1f} us   tail busy (union of statistics + exchange kernels) {union(tail) / 1e3 / batches:7.1f} us per step = "
      f"{union(tail) / window:.3f} of the step   path kernels overlapping {over / window:.2f} of the time")
agg = collections.defaultdict(list)
for r in inwin:
    if is_tail(r["Kernel_Name"]) or is_path(r["Kernel_Name"]):
        agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:62s} {len(d) / batches:5.1f} per step   mean {sum(d) / len(d):8.1f} us   max {max(d):8.1f} us   total {sum(d) / batches:8.1f} us per step")
