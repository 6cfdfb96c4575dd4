"""Per-kernel HBM traffic from tools/pmc_traffic.sh: mean FETCH_SIZE x 2 (gfx950: the counter reports half the bytes of a wide coalesced
stream, MI355X_MICROARCH.md section HBM) and WRITE_SIZE, KiB units -> bytes per launch, against the algorithmic bytes given on the
command line.   python tools/pmc_traffic_summary.py <dir> K n_paths"""
import collections, csv, glob, json, sys
d, K, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
val = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for p in glob.glob(f"{d}/{c}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(p)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mcp::", "")
            if name.startswith(("hist_kernel", "scan_kernel", "final_kernel", "mc_sweep", "mc_paths", "pass0")):
                val[name][c].append(float(r["Counter_Value"]) * 1024.0)
                if c == "FETCH_SIZE":
                    dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
vt = 4.0 * K * n
algo = {"hist_kernel<0, false>": (vt, 0), "hist_kernel<1, false>": (vt, 0), "hist_kernel<2, false>": (vt, 0),
        "mc_sweep_shared_kernel<4, 4, false, false>": (0, None)}
out = {}
for name, v in sorted(val.items()):
    rd = 2.0 * sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1)
    wr = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
    us = sum(dur[name]) / max(len(dur[name]), 1)
    out[name] = {"launches": len(v["FETCH_SIZE"]), "read_bytes": rd, "written_bytes": wr, "us_under_pmc": us,
                 "read_over_algorithmic": (rd / algo[name][0]) if name in algo and algo[name][0] else None,
                 "read_TB_per_s_under_pmc": rd / us / 1e6 if us else None}
print(json.dumps({"K": K, "n_paths": n, "terminal_bytes": vt, "kernels": out}, indent=1))
