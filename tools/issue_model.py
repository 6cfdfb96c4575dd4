"""Issue-cycle model of the path kernel's step loop, priced on the instructions the kernel EXECUTES.

  python tools/issue_model.py [--nb 4] [--clock-hz 1.98e9] [--pmc profiles/r03_pmc_summary.json] [-o profiles/issue_model.json]

Compiles mcp_paths_inst.hip for one NB to assembly (hipcc -S, no GPU needed), finds the T-step loop of
mc_paths_kernel<NB,1,1,false,false,false> (the loop body with the most VALU instructions), classifies every VALU
instruction by the issue cost measured on this chip (tools/microbench/valu_rates.hip -> profiles/r01_valu_rates.txt,
cycles per wave-instruction per SIMD at 8 waves/SIMD) and writes the table bench.py prices `roofline.issue_model` with:

  plain VOP1/2/3 fp32 / integer / bit op, VGPR or literal sources            2.6
  any VALU instruction with an SGPR source operand                           4.3
  v_pk_* (packed fp32)                                                       4.3
  v_cvt_*, v_bfe_*, v_mul_lo/hi_*, 64-bit shifts                             4.4
  v_mad_u64_u32                                                              4.7
  v_log/exp/sqrt/rcp/rsq/sin/cos_f32                                         8.2

The clock is the one measured INSIDE the un-profiled kernel under its own load (tools/clock_probe.py: s_memtime /
s_memrealtime stamps of a diagnostic build, profiles/r03_clock.txt: 2.30 GHz), not the nominal 2.4 GHz and not the clock under
the profiler (GRBM_GUI_ACTIVE / 8 / kernel time of a --pmc pass: 1.97 GHz, where the same kernel takes 18 % longer).  It is a builder-authored model, NOT a hardware peak: the primary roofline.frac is against the 157.3 TFLOP/s fp32
vector peak of MI355X_MICROARCH.md.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "monte_carlo_portfolio_amd", "csrc")
SGPR = re.compile(r"^(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec)$")
TRANS = ("v_log_f32", "v_exp_f32", "v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32")


def classify(op, operands):
    if op.startswith("v_mad_u64_u32"):
        return "v_mad_u64_u32", 4.7
    if op.startswith(TRANS):
        return "transcendental", 8.2
    if op.startswith("v_pk_"):
        return "packed fp32 (v_pk_*)", 4.3
    srcs = operands[1:]
    if op.startswith("v_mad_u64") or op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        srcs = operands[2:]
    if any(SGPR.match(o) for o in srcs if not o.startswith("vcc") or op.startswith("v_cndmask")):
        return "VALU with an SGPR source", 4.3
    if op.startswith(("v_cvt_", "v_bfe_", "v_mul_lo_", "v_mul_hi_", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")):
        return "cvt / bfe / 32-bit multiply", 4.4
    return "plain VOP (VGPR / literal sources)", 2.6


def step_loop(lines, pattern):
    k0 = next(i for i, l in enumerate(lines) if l.startswith("_ZN3mcp15mc_paths_kernel") and pattern in l.split(":")[0] and ":" in l)
    k1 = next(i for i in range(k0, len(lines)) if "s_endpgm" in lines[i])
    labels = {l.split(":")[0]: i for i, l in enumerate(lines[k0:k1], k0) if l.startswith(".LBB")}
    loops = []
    for i in range(k0, k1):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)", lines[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:           # backward branch = loop
            loops.append((labels[m.group(1)], i, m.group(1)))
    best = None
    for a, b, name in loops:                                                # innermost loops only: the step loop is one of them
        if any(a2 >= a and b2 <= b and (a2, b2) != (a, b) for a2, b2, _ in loops):
            continue
        body = lines[a:b + 1]
        n = sum(1 for l in body if l.strip().startswith("v_"))
        if best is None or n > best[0]:
            best = (n, body, name)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=4)
    ap.add_argument("--clock-hz", type=float, default=None)
    ap.add_argument("--pmc", default=None, help="pmc summary JSON (tools/pmc_summary.py): takes the clock under the profiler from it")
    ap.add_argument("--clock-source", default=None, help="where --clock-hz comes from (e.g. profiles/r03_clock.txt)")
    ap.add_argument("-o", "--out", default=None)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        s_path = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", f"-DMCP_NB={a.nb}", "-S",
                        "--cuda-device-only", os.path.join(CSRC, "mcp_paths_inst.hip"), "-o", s_path], check=True, capture_output=True)
        lines = open(s_path).read().split("\n")
    n, body, label = step_loop(lines, f"ILi{a.nb}ELi1ELi1ELb0ELb0ELb0EE")
    rows = collections.OrderedDict()
    other = collections.Counter()
    for l in body:
        t = l.strip()
        if not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        if not op.startswith("v_"):
            other[op.split("_")[0] + "_" + op.split("_")[1] if "_" in op else op] += 1
            continue
        operands = [o.strip() for o in t[len(op):].split(";")[0].split(",")]
        operands = [o.split()[0] for o in operands if o]
        cls, cyc = classify(op, operands)
        key = (cls, op.replace("_e32", "").replace("_e64", ""))
        rows.setdefault(key, [0, cyc])[0] += 1
    clock = a.clock_hz
    src = a.clock_source or "command line"
    prof_clock = None
    if a.pmc:
        p = json.load(open(a.pmc))
        prof_clock = p["kernel_cycles"] / (p["kernel_ms_under_pmc"] * 1e-3)
        if clock is None:
            clock = prof_clock
            src = f"{os.path.relpath(a.pmc, ROOT)}: GRBM_GUI_ACTIVE / 8 / kernel time of the PMC pass (the clock UNDER THE PROFILER)"
    if clock is None:
        clock = 2.4e9
        src = "nominal (no measurement given)"
    table = [{"class": k[0], "inst": k[1], "count": v[0], "cycles": v[1]} for k, v in rows.items()]
    total = sum(r["count"] * r["cycles"] for r in table)
    out = {
        "what": "Issue-cycle model of one wave-step (64 paths x 1 step) of mc_paths_kernel<%d,1,1,false,false,false>, priced on the "
                "instructions the kernel EXECUTES (step loop %s of the hipcc -S listing at HEAD) at the issue costs measured on this chip "
                "(profiles/r01_valu_rates.txt, 8 waves/SIMD) and at the clock measured under this kernel's load.  Builder-authored model, "
                "NOT a hardware peak: roofline.frac in the bench line is against the 157.3 TFLOP/s fp32 vector peak." % (a.nb, label),
        "generated_by": "tools/issue_model.py",
        "clock_hz": clock, "clock_source": src, "clock_hz_under_profiler": prof_clock, "simds": 1024,
        "valu_insts_per_wave_step": n, "other_insts_per_wave_step": dict(other),
        "rows": table, "cycles_per_wave_step": total,
        "by_class": {c: sum(r["count"] * r["cycles"] for r in table if r["class"] == c) for c in dict.fromkeys(r["class"] for r in table)},
    }
    text = json.dumps(out, indent=1)
    if a.out:
        open(a.out, "w").write(text + "\n")
    print(f"step loop {label}: {n} VALU instructions, {total:.0f} issue cycles per wave-step at the measured costs; clock {clock / 1e9:.3f} GHz ({src})")
    for c, v in out["by_class"].items():
        print(f"  {v:8.1f}  {c}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
