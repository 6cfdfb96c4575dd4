"""BASELINE configs[4] at its own size through the host-level API: 10,000 Dirichlet portfolios x 16 assets x 10^6 paths x
252 steps on ONE MI355X (the 8-GPU form shards the portfolios; each GPU then does 1/8 of this).  V_T[K x n] (40 GB) is
never resident: mcp_simulate tiles the portfolios under the context's terminal budget."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monte_carlo_portfolio_amd import simulate_paths, synthetic
from monte_carlo_portfolio_amd.simulate import Context
K, P, T = 10_000, int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 252
budget = int(float(sys.argv[2]) * 2 ** 30) if len(sys.argv) > 2 else 8 << 30
mu, cov = synthetic.synthetic_market(16)
W = synthetic.dirichlet_weights(16, K)
ctx = Context(0, terminal_budget=budget)
simulate_paths(mu, cov, W[:600], n_steps=8, n_paths=4096, seed=1, as_array=True, context=ctx)      # load kernels
t0 = time.perf_counter()
st = simulate_paths(mu, cov, W, n_steps=T, n_paths=P, seed=synthetic.BENCH_SEED, as_array=True, context=ctx)
dt = time.perf_counter() - t0
tile = max(512, (budget // (4 * P)) // 512 * 512)
print(f"configs[4] full size: K={K} x paths={P:,} x steps={T}: {dt:.3f} s wall (PCIe-inclusive, {-(-K // tile)} tiles of {tile} portfolios, "
      f"budget {budget / 2**30:.0f} GiB) -> {K * P / dt:.3e} portfolio-paths/s, W.r product {2.0 * K * 16 * P * T / dt / 1e12:.1f} TFLOP/s")
i = int(np.argmax(st["sharpe"]))
print(f"max-Sharpe portfolio {i}: sharpe {st['sharpe'][i]:.6f} mean {st['mean'][i]:.6f} std {st['std'][i]:.6f} VaR95 {st['var'][i]:.6f} CVaR95 {st['cvar'][i]:.6f}; "
      f"min-VaR-loss portfolio {int(np.argmax(st['var']))}; all n == {P}: {bool(np.all(st['n'] == P))}; n_tail range {int(st['n_tail'].min())}..{int(st['n_tail'].max())}")
ana = (1.0 + W.astype(np.float32).astype(np.float64) @ mu.astype(np.float32).astype(np.float64)) ** T - 1.0
print(f"analytic mean check: max |mean - (1+w.mu)^T + 1| / standard error = {np.max(np.abs(st['mean'] - ana) / (st['std'] / np.sqrt(P))):.2f}")
