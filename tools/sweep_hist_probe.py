"""Timing probe for the reference's sweep on historical rows (mcp_sweep_historical): rows x assets x portfolios, host call
(PCIe-inclusive) and, under rocprofv3 --kernel-trace --stats, the kernel alone.   python tools/sweep_hist_probe.py [R N P]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monte_carlo_portfolio_amd import sweep
shapes = [tuple(int(a) for a in sys.argv[1:4])] if len(sys.argv) > 3 else [(13, 3, 2500), (252, 16, 10_000), (4096, 16, 2500), (4096, 64, 2500)]
for R, N, P in shapes:
    rng = np.random.default_rng(R + N)
    Rm = rng.normal(0.0004, 0.02, (R, N))
    W = np.random.RandomState(7).dirichlet(np.ones(N), P)
    Rc, mean, cov = sweep.sweep_inputs(Rm, 252)
    sweep.score_portfolios(Rc, mean, cov, W[:8], 0.03)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); s = sweep.score_portfolios(Rc, mean, cov, W, 0.03); ts.append(time.perf_counter() - t0)
    ref = np.percentile(Rc @ W[:64].T, (1 - 0.95) * 100, axis=0)
    print(f"rows {R:5d} x assets {N:2d} x portfolios {P:6d}: host call {min(ts) * 1e3:8.3f} ms (PCIe-inclusive) = {P / min(ts):.3e} portfolios/s; "
          f"max |VaR - np.percentile| over 64 portfolios {np.max(np.abs(s['var_95'][:64] - ref)):.1e}")
