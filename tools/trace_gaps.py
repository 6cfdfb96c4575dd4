"""Summarise a rocprofv3 --kernel-trace CSV as a timeline: per kernel start offset, duration and the gap to the previous
kernel's end (us); optional --overlap: pairs of mc_paths_kernel launches whose intervals overlap."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
lo = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else len(rows)
prev_end = None
for i, r in enumerate(rows[lo:hi]):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void mcp::", "").replace("mcp::", "")[:48]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{lo+i:5d} {name:48s} start {(s-t0)/1e3:12.1f} us  dur {(e-s)/1e3:9.1f} us  gap {gap:8.1f} us  stream {r.get('Stream_Id', r.get('Queue_Id', '?'))}")
    prev_end = e if prev_end is None else max(prev_end, e)
if "--overlap" in sys.argv:
    paths = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "mc_paths_kernel" in r["Kernel_Name"]]
    n = sum(1 for a, b in zip(paths, paths[1:]) if b[0] < a[1])
    print(f"# {len(paths)} mc_paths_kernel launches, {n} consecutive pairs overlap in time")
