"""Soak (GPU box): the multi-shard context (logical shards of one device: streams, events, the kernel exchange) on random
shapes for a time budget, every call compared with the plain one-device context."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from monte_carlo_portfolio_amd import simulate_paths, synthetic
from monte_carlo_portfolio_amd.simulate import Context
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(99)
plain = Context(0)
ctxs = {S: Context([0] * S) for S in (2, 3, 5, 8)}
calls = wrong = 0
t_end = time.time() + budget
while time.time() < t_end:
    N = int(rng.choice([3, 16, 21])); K = int(rng.choice([1, 1, 4, 17, 400])); T = int(rng.integers(1, 20))
    P = int(rng.choice([7, 1000, 4097, 100_000, 1_000_000])) if K < 17 else int(rng.choice([65, 1000, 20_000]))
    S = int(rng.choice([2, 3, 5, 8])); shard = "portfolios" if (K >= 2 * S and rng.random() < 0.5) else "paths"
    mu, cov = synthetic.synthetic_market(N)
    W = synthetic.equal_weights(N) if K == 1 else synthetic.dirichlet_weights(N, K)
    kw = dict(n_steps=T, n_paths=P, seed=int(rng.integers(1, 1 << 40)), as_array=True, rf=0.001)
    a = simulate_paths(mu, cov, W, context=plain, **kw)
    for rep in range(3):                                   # back-to-back calls on the same context
        b = simulate_paths(mu, cov, W, context=ctxs[S], devices=[0] * S, shard=shard, **kw)
        ok = all(np.array_equal(a[k], b[k]) for k in ("n", "n_tail", "var", "x_lo", "x_hi", "min", "max")) and \
            all(np.allclose(a[k], b[k], rtol=1e-13, atol=1e-15) for k in ("mean", "std", "sharpe", "cvar", "sum_tail"))
        wrong += 0 if ok else 1
        calls += 1
print(f"multi-shard context: {calls} calls over random shapes, {wrong} differ from the one-device result")
sys.exit(1 if wrong else 0)
