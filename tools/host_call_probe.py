"""Latency of the host-level call (mcp_simulate through ctypes): what a Streamlit session sees."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monte_carlo_portfolio_amd import simulate_paths, synthetic
mu, cov = synthetic.synthetic_market(16); w = synthetic.equal_weights(16)
for P, store in ((10_000, False), (100_000, False), (1_000_000, False), (1_000_000, True), (10_000_000, False)):
    simulate_paths(mu, cov, w, n_steps=252, n_paths=P, seed=1, store=store)
    t = time.perf_counter(); n = 5
    for i in range(n): simulate_paths(mu, cov, w, n_steps=252, n_paths=P, seed=2 + i, store=store)
    dt = (time.perf_counter() - t) / n
    print(f"paths={P:>10,} store={store}: {dt*1e3:8.3f} ms per call -> {P/dt:.3e} paths/s")
