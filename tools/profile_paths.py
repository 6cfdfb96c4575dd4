"""Launch the path kernel a few times (profiling target for rocprofv3; see profiles/README.md)."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from monte_carlo_portfolio_amd import synthetic
from monte_carlo_portfolio_amd.engine import PathEngine
from monte_carlo_portfolio_amd.simulate import prepare_inputs

ap = argparse.ArgumentParser()
ap.add_argument("--assets", type=int, default=16)
ap.add_argument("--steps", type=int, default=252)
ap.add_argument("--paths", type=int, default=1_000_000)
ap.add_argument("--launches", type=int, default=5)
ap.add_argument("--full", action="store_true", help="full pass (paths + statistics) instead of the path kernel alone")
ap.add_argument("--native-math", action="store_true")
a = ap.parse_args()
mu, cov = synthetic.synthetic_market(a.assets)
mu32, L, W32 = prepare_inputs(mu, cov, synthetic.equal_weights(a.assets))
eng = PathEngine(mu32, L, W32, a.steps, a.paths, native_math=a.native_math)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
(eng.step if a.full else eng.launch_paths_only)(synthetic.BENCH_SEED)
torch.cuda.synchronize()
ev0.record()
for _ in range(a.launches):
    (eng.step if a.full else eng.launch_paths_only)(synthetic.BENCH_SEED)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / a.launches
print(f"assets={a.assets} steps={a.steps} paths={a.paths} native={a.native_math}: {ms:.3f} ms/launch -> {a.paths / ms * 1e3:.4e} paths/s")
