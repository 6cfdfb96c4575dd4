"""In-kernel clock of mc_paths_kernel under its own load (MI355X_MICROARCH.md, DVFS give-back item 6).

  python tools/clock_probe.py build      CPU box: a DIAGNOSTIC library (tools/lab_build/libmcport_diag.so, -DMCP_DIAG_CLOCK)
  python tools/clock_probe.py run        GPU box: ~3 s of back-to-back launches of the bench workload, then the median over
                                         workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz, and the kernel time beside it

The product library never executes a stamp; the diagnostic build's stamps go to a buffer nothing else reads.  The profiler's
figure (GRBM_GUI_ACTIVE / 8 / kernel time of a --pmc pass) is the clock UNDER THE PROFILER, where the same kernel takes 18 %
longer; this is the clock of the un-profiled kernel the bench times."""
import ctypes, os, statistics, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LAB = os.path.join(ROOT, "tools", "lab_build")
OUT = os.path.join(LAB, "libmcport_diag.so")
if sys.argv[1:2] == ["build"]:
    os.makedirs(LAB, exist_ok=True)
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "monte_carlo_portfolio_amd", "csrc"), "-j8", "BUILD=build_diag", f"OUT={OUT}",
                        "EXTRA=-DMCP_DIAG_CLOCK -fgpu-rdc", "LDEXTRA=-fgpu-rdc"], capture_output=True, text=True)
    print("diag build", "ok" if r.returncode == 0 else "FAILED\n" + r.stdout[-3000:] + r.stderr[-3000:])
    sys.exit(r.returncode)
os.environ["MCP_LIB_PATH"] = OUT
import numpy as np, torch
from monte_carlo_portfolio_amd import _ffi, synthetic
from monte_carlo_portfolio_amd.engine import PathEngine
from monte_carlo_portfolio_amd.simulate import prepare_inputs
lib = _ffi.lib()
lib.mcp_diag_read.argtypes = [np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS"), ctypes.c_int]
mu, cov = synthetic.synthetic_market(16)
mu32, L, W32 = prepare_inputs(mu, cov, synthetic.equal_weights(16))
P = 1_000_000
eng = PathEngine(mu32, L, W32, 252, P, pipeline=False)
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
t_end = time.time() + seconds
n = 0
while time.time() < t_end:                      # warm: the chip settles at its working clock
    for _ in range(50):
        eng.launch_paths_only(synthetic.BENCH_SEED)
    torch.cuda.synchronize(); n += 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    eng.launch_paths_only(synthetic.BENCH_SEED)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
nb = 3907
st = np.zeros(2 * nb, np.uint64)
assert lib.mcp_diag_read(st, nb) == 0
cyc, ref = st[0::2].astype(np.float64), st[1::2].astype(np.float64)
ok = ref > 0
ghz = cyc[ok] / ref[ok] * 0.1
print(f"mc_paths_kernel<4,1,1,false,false,false>, 10^6 paths x 252 steps, after {n} warm launches ({seconds:g} s): kernel {ms:.3f} ms (diagnostic build, HIP events); "
      f"in-kernel clock median {statistics.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}) over {int(ok.sum())} workgroups; "
      f"workgroup lifetime median {statistics.median(cyc[ok]):.0f} shader cycles = {statistics.median(cyc[ok]) / 252:.0f} per step")
