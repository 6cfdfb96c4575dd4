"""First end-to-end GPU check: HIP path vs C oracle, bit-exact, plus a timing."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from monte_carlo_portfolio_amd import simulate_paths, synthetic, _ffi
from monte_carlo_portfolio_amd.simulate import prepare_inputs
from oracle import mc_oracle

def check(N, T, P, native=False):
    mu, cov = synthetic.synthetic_market(N)
    w = synthetic.equal_weights(N)
    r = simulate_paths(mu, cov, w, n_steps=T, n_paths=P, seed=synthetic.BENCH_SEED, store=True, native_math=native)
    mu32, L, W = prepare_inputs(mu, cov, w)
    t = time.time(); ref = mc_oracle.simulate(mu32, L, W, T, P, synthetic.BENCH_SEED)[0]; dt = time.time() - t
    V = r["terminal"]
    nbad = int((V.view(np.uint32) != ref.view(np.uint32)).sum())
    rel = np.abs(V.astype(np.float64) - ref) / np.abs(ref)
    x = ref.astype(np.float64) / 1.0 - 1.0
    var = np.percentile(x, (1 - 0.95) * 100)
    print(f"N={N} T={T} P={P} native={native}: mismatching terminals {nbad}/{P}  max rel {rel.max():.3e}  "
          f"oracle {P/dt:.0f} paths/s | mean {r['mean']:.12g} vs {x.mean():.12g} | std {r['std']:.12g} vs {x.std(ddof=1):.12g} | "
          f"VaR {r['var']:.15g} vs {var:.15g} | CVaR {r['cvar']:.12g} vs {x[x<=var].mean():.12g} | n_tail {r['n_tail']} vs {(x<=var).sum()}")

check(16, 252, 20000)
check(3, 252, 10000)
check(16, 252, 20000, native=True)
check(64, 20, 2000)
check(5, 17, 1000)
mu, cov = synthetic.synthetic_market(16); w = synthetic.equal_weights(16)
for native in (False, True):
    for P in (1_000_000,):
        simulate_paths(mu, cov, w, n_steps=252, n_paths=P, seed=1, native_math=native)
        t = time.time()
        for _ in range(3): simulate_paths(mu, cov, w, n_steps=252, n_paths=P, seed=1, native_math=native)
        dt = (time.time() - t) / 3
        print(f"P={P} native={native}: {dt*1e3:.2f} ms  -> {P/dt:.3e} paths/s")
