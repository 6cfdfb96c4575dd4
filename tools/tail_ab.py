"""Same-process A/B of the pipelined engine with and without exchanges in the tail (one GPU):
   1 shard  vs  S logical shards (exchange kernels + record copies in the tail), skewed and unskewed schedule.
   Interleaved rounds, every engine runs the same total number of paths per step.   python tools/tail_ab.py [--rounds 8 --steps 40]"""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monte_carlo_portfolio_amd import synthetic
from monte_carlo_portfolio_amd.engine import PathEngine
from monte_carlo_portfolio_amd.simulate import prepare_inputs

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=8); ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--paths", type=int, default=1_000_000); ap.add_argument("--shards", type=int, default=2)
a = ap.parse_args()
mu, cov = synthetic.synthetic_market(16)
mu32, L, W32 = prepare_inputs(mu, cov, synthetic.equal_weights(16))
S = a.shards
engines = {
    "1 shard": PathEngine(mu32, L, W32, 252, a.paths),
    f"{S} shards, skewed (6 buffers, 2 stats streams)": PathEngine(mu32, L, W32, 252, a.paths, logical_shards=S, skew=True),
    f"{S} shards, skewed (6 buffers, 6 stats streams)": PathEngine(mu32, L, W32, 252, a.paths, logical_shards=S, skew=True, n_stats_streams=6),
    f"{S} shards, skewed (6 buffers, 1 stats stream)": PathEngine(mu32, L, W32, 252, a.paths, logical_shards=S, skew=True, n_stats_streams=1),
    f"{S} shards, unskewed, 2 buffers": PathEngine(mu32, L, W32, 252, a.paths, logical_shards=S, skew=False, n_buffers=2),
    f"{S} shards, unskewed, 4 buffers": PathEngine(mu32, L, W32, 252, a.paths, logical_shards=S, skew=False, n_buffers=4),
    "1 shard, skewed (forced)": PathEngine(mu32, L, W32, 252, a.paths, skew=True),
    "1 shard, 4 CUs reserved": PathEngine(mu32, L, W32, 252, a.paths, cu_reserve=4),
    f"{S} shards, unskewed, 2 buffers, 4 CUs reserved": PathEngine(mu32, L, W32, 252, a.paths, logical_shards=S, n_buffers=2, cu_reserve=4),
}
ref = None
for name, e in engines.items():
    for _ in range(5):
        e.step(synthetic.BENCH_SEED)
    st = e.stats()[0]
    if ref is None:
        ref = st
    assert st["var"] == ref["var"] and st["n_tail"] == ref["n_tail"] and abs(st["sharpe"] / ref["sharpe"] - 1) < 1e-12, name
times = {n: [] for n in engines}
for r in range(a.rounds):
    for name, e in engines.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            e.step(synthetic.BENCH_SEED)
        e.synchronize(); torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / a.steps * 1e3)
base = statistics.median(times["1 shard"])
for name, ts in times.items():
    m = statistics.median(ts)
    print(f"{name:52s} median {m:.3f} ms/step  min {min(ts):.3f}  -> {a.paths / m * 1e3:.4e} paths/s   x{m / base:.3f} of 1 shard")
