"""Latency of one host-level call at the size of configs[0] (3 assets, 10,000 paths, 252 steps) and a few sizes around it:
simulate_paths -> mcp_simulate through ctypes, synchronous, PCIe-inclusive.   python tools/small_call.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monte_carlo_portfolio_amd import simulate_paths, synthetic  # noqa: E402

for n_assets, n_paths, n_steps, K in ((3, 10_000, 252, 1), (3, 10_000, 252, 2500), (16, 10_000, 252, 1), (16, 100_000, 252, 1), (16, 1_000_000, 252, 1)):
    mu, cov = synthetic.synthetic_market(n_assets)
    rng = np.random.default_rng(1)
    W = rng.dirichlet(np.ones(n_assets), K) if K > 1 else synthetic.equal_weights(n_assets)
    for _ in range(5):
        simulate_paths(mu, cov, W, n_steps=n_steps, n_paths=n_paths, seed=7, as_array=True)
    ts = []
    for _ in range(50):
        t0 = time.perf_counter()
        simulate_paths(mu, cov, W, n_steps=n_steps, n_paths=n_paths, seed=7, as_array=True)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{n_assets:3d} assets x {n_paths:9,d} paths x {n_steps} steps x {K:5d} portfolio(s): median {ts[25] * 1e6:9.1f} us  min {ts[0] * 1e6:9.1f} us  "
          f"p90 {ts[45] * 1e6:9.1f} us per call", flush=True)
