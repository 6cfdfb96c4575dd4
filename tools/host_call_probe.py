"""Latency of the host-level call (mcp_simulate through ctypes): what a Streamlit session sees.
  python tools/host_call_probe.py            table over path counts
  python tools/host_call_probe.py --split    1M paths: Python surface vs ctypes call vs kernel time (HIP events)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monte_carlo_portfolio_amd import _ffi, simulate_paths, synthetic
from monte_carlo_portfolio_amd.simulate import default_context, prepare_inputs
mu, cov = synthetic.synthetic_market(16); w = synthetic.equal_weights(16)
if "--split" in sys.argv:
    P, n = 1_000_000, 20
    mu32, L, W = prepare_inputs(mu, cov, w)
    prm = _ffi.make_params(16, 252, 1)
    ctx = default_context(0)
    ctx.simulate(prm, mu32, L, W, 1, 0, P, False)
    t = time.perf_counter()
    for i in range(n): ctx.simulate(prm, mu32, L, W, 2 + i, 0, P, False)
    t_c = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for i in range(n): simulate_paths(mu, cov, w, n_steps=252, n_paths=P, seed=2 + i)
    t_py = (time.perf_counter() - t) / n
    t = time.perf_counter()
    for i in range(n): prepare_inputs(mu, cov, w)
    t_prep = (time.perf_counter() - t) / n
    import torch
    from monte_carlo_portfolio_amd.engine import PathEngine
    eng = PathEngine(mu32, L, W, 252, P, pipeline=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.launch_paths_only(1); torch.cuda.synchronize()
    e0.record()
    for i in range(n): eng.launch_paths_only(2 + i)
    e1.record(); torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / n
    eng.step(1); torch.cuda.synchronize()
    e0.record()
    for i in range(n): eng.step(2 + i)
    e1.record(); torch.cuda.synchronize()
    s_ms = e0.elapsed_time(e1) / n
    print(f"1M paths: path kernel {k_ms:.3f} ms | kernel + statistics back to back (device time) {s_ms:.3f} ms | "
          f"mcp_simulate via ctypes {t_c*1e3:.3f} ms | simulate_paths (Python surface) {t_py*1e3:.3f} ms | prepare_inputs alone {t_prep*1e3:.3f} ms")
    print(f"ratio mcp_simulate / path kernel = {t_c*1e3/k_ms:.3f}")
    sys.exit(0)
for P, store in ((10_000, False), (100_000, False), (1_000_000, False), (1_000_000, True), (10_000_000, False)):
    simulate_paths(mu, cov, w, n_steps=252, n_paths=P, seed=1, store=store)
    t = time.perf_counter(); n = 5
    for i in range(n): simulate_paths(mu, cov, w, n_steps=252, n_paths=P, seed=2 + i, store=store)
    dt = (time.perf_counter() - t) / n
    print(f"paths={P:>10,} store={store}: {dt*1e3:8.3f} ms per call -> {P/dt:.3e} paths/s")
