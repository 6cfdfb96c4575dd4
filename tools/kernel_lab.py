"""Kernel lab: A/B timing of path-kernel variants in ONE process with interleaved rounds
(cdna_hip_programming.md section 5.4 rule 24), each variant a full libmcport build with different -D flags.

  python tools/kernel_lab.py build  name1:"-DFLAG=1 ..." name2:"..."     (CPU box: cross-compiles)
  python tools/kernel_lab.py run [--paths P --rounds R --assets N] name1 name2 ...   (GPU box)
"""
import ctypes, os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LAB = os.path.join(ROOT, "tools", "lab_build")
CSRC = os.path.join(ROOT, "monte_carlo_portfolio_amd", "csrc")


def build(specs):
    os.makedirs(LAB, exist_ok=True)
    for spec in specs:
        name, _, flags = spec.partition(":")
        out = os.path.join(LAB, f"libmcport_{name}.so")
        r = subprocess.run(["make", "-C", CSRC, "-j8", f"BUILD=build_{name}", f"OUT={out}", f"EXTRA={flags}"],
                           capture_output=True, text=True)
        print(name, "ok" if r.returncode == 0 else "FAILED\n" + r.stdout[-2000:] + r.stderr[-2000:])


def run(names, paths, rounds, assets, steps, native, stats=True, K=1):
    import numpy as np, torch
    from monte_carlo_portfolio_amd import _ffi, synthetic
    from monte_carlo_portfolio_amd.simulate import prepare_inputs
    base = _ffi.lib()     # binds signatures; also decides the HIP runtime
    mu, cov = synthetic.synthetic_market(assets)
    mu32, L, W32 = prepare_inputs(mu, cov, synthetic.equal_weights(assets) if K == 1 else synthetic.dirichlet_weights(assets, K))
    prm = _ffi.make_params(assets, steps, K, native_math=native)
    packed = torch.from_numpy(_ffi.pack_params(mu32, L, W32)).cuda()
    term = torch.empty((K, paths), dtype=torch.float32, device="cuda")
    pivot = torch.from_numpy(_ffi.pivots(prm, mu32, L, W32)).cuda()
    partials = torch.zeros(base.mcp_ws_bytes(_ffi.WS_PARTIALS, K, paths) // 8, dtype=torch.int64, device="cuda")
    hist = torch.zeros(base.mcp_ws_bytes(_ffi.WS_HIST, K, paths) // 8, dtype=torch.int64, device="cuda")
    libs = {}
    for n in names:
        path = _ffi.LIB_PATH if n == "base" else os.path.join(LAB, f"libmcport_{n}.so")
        Lb = ctypes.CDLL(path)
        Lb.mcp_launch_paths.restype = ctypes.c_int
        Lb.mcp_launch_paths.argtypes = _ffi.SIGNATURES["mcp_launch_paths"][1]
        libs[n] = Lb
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch(Lb):
        if stats:
            hist.zero_()
        rc = Lb.mcp_launch_paths(ctypes.byref(prm), ctypes.c_void_p(packed.data_ptr()), ctypes.c_void_p(pivot.data_ptr()),
                                 synthetic.BENCH_SEED, 0, paths, ctypes.c_void_p(term.data_ptr()), paths,
                                 ctypes.c_void_p(partials.data_ptr()) if stats else None, ctypes.c_void_p(hist.data_ptr()) if stats else None, stream)
        assert rc == 0, rc

    ref = None
    for n, Lb in libs.items():
        term.zero_(); launch(Lb); torch.cuda.synchronize()
        t = term.cpu().numpy().copy()
        if ref is None:
            ref = t
        print(f"{n}: identical to {names[0]}: {np.array_equal(t.view(np.uint32), ref.view(np.uint32))}  max rel diff {np.abs(t/ref-1).max():.2e}")
    times = {n: [] for n in names}
    for r in range(rounds):
        for n, Lb in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); launch(Lb); launch(Lb); e1.record(); torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / 2)
    for n in names:
        ts = times[n]
        tf = f"  W.r product {2.0 * K * assets * paths * steps / statistics.median(ts) / 1e9:.1f} TFLOP/s" if K > 1 else ""
        print(f"{n:>14}: median {statistics.median(ts):.3f} ms  min {min(ts):.3f} ms  -> {paths / statistics.median(ts) * 1e3:.4e} paths/s{tf}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        import argparse
        ap = argparse.ArgumentParser()
        ap.add_argument("cmd"); ap.add_argument("names", nargs="+")
        ap.add_argument("--paths", type=int, default=1_000_000); ap.add_argument("--rounds", type=int, default=10)
        ap.add_argument("--assets", type=int, default=16); ap.add_argument("--steps", type=int, default=252)
        ap.add_argument("--native", action="store_true")
        ap.add_argument("--no-stats", action="store_true", help="terminal values only (no fused statistics epilogue)")
        ap.add_argument("--portfolios", type=int, default=1, help="K > 16: the MFMA sweep kernels (the launch then includes the lean digit-0 pass)")
        a = ap.parse_args()
        run(a.names, a.paths, a.rounds, a.assets, a.steps, a.native, not a.no_stats, a.portfolios)
