"""Timing probe for the K-portfolio path (BASELINE configs[4] shape, scaled)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monte_carlo_portfolio_amd import _ffi, synthetic
from monte_carlo_portfolio_amd.engine import PathEngine
from monte_carlo_portfolio_amd.simulate import prepare_inputs
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
mu, cov = synthetic.synthetic_market(16)
mu32, L, W32 = prepare_inputs(mu, cov, synthetic.dirichlet_weights(16, K))
eng = PathEngine(mu32, L, W32, 252, P, pipeline=False)
for name, fn in (("paths", eng.launch_paths_only), ("full", eng.step)):
    fn(1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(1); fn(1); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 2
    flops = 2.0 * K * 16 * P * 252
    print(f"K={K} P={P} {name}: {ms:.2f} ms  -> {K*P/ms*1e3:.3e} portfolio-paths/s, W.r product {flops/ms/1e9:.1f} TFLOP/s; "
          f"configs[4] (1e4 x 1e6) would take {ms*(1e4/K)*(1e6/P)/1e3:.2f} s on this GPU")
st = eng.stats()
print("argmax sharpe", int(np.argmax(st["sharpe"])), float(st["sharpe"].max()))
