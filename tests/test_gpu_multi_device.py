"""Several PHYSICAL GPUs (skipped on a one-GPU box): the same assertions the logical-shard tests make on one device
(tests/test_gpu_round2.py), here across distinct devices -- the in-library context over RCCL and over the peer-access
kernel exchange, the portfolio-sharded context, and one process per GPU under torch.distributed (nccl = RCCL over xGMI) through
bench.py's own rank-spawning parent.  No round so far has had a node with more than one GPU: these tests are written to be the
first thing that runs there.  Every multi-device run happens in a child process (one HIP runtime, environment switches read at
context creation, and a failure cannot take the test session's context with it)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def n_devices(mcp_lib):
    n = mcp_lib.mcp_device_count()
    if n < 2:
        pytest.skip(f"{n} HIP device(s) visible: the multi-device tests need at least 2")
    return n


CHILD = """
import json, os, sys
sys.path.insert(0, {root!r})
import numpy as np
from monte_carlo_portfolio_amd import _ffi, simulate_paths, synthetic
from monte_carlo_portfolio_amd.simulate import Context
devices = {devices!r}; K = {K}; P = {P}; shard = {shard!r}
mu, cov = synthetic.synthetic_market(16)
W = synthetic.equal_weights(16) if K == 1 else synthetic.dirichlet_weights(16, K)
kw = dict(n_steps=25, n_paths=P, seed=11, rf=0.002, as_array=True, store=True)
one = simulate_paths(mu, cov, W, **kw)
ctx = Context(devices)
try:
    many = simulate_paths(mu, cov, W, context=ctx, devices=devices, shard=shard, **kw)
    again = simulate_paths(mu, cov, W, context=ctx, devices=devices, shard=shard, **kw)      # the read-and-clear protocol across devices
    mode, note = ctx.exchange()
finally:
    ctx.close()
out = {{"mode": mode, "note": note, "terminal_equal": bool(np.array_equal(one[1].view(np.uint32), many[1].view(np.uint32))),
       "repeat_equal": one[0].dtype == again[0].dtype and many[0].tobytes() == again[0].tobytes()}}
for key in ("n", "n_tail", "var", "x_lo", "x_hi", "min", "max"):
    out["exact_" + key] = bool(np.array_equal(one[0][key], many[0][key]))
for key in ("mean", "std", "sharpe", "cvar", "sum_tail"):
    out["rel_" + key] = float(np.max(np.abs(many[0][key] - one[0][key]) / np.maximum(np.abs(one[0][key]), 1e-300)))
out["argmax_equal"] = int(np.argmax(one[0]["sharpe"])) == int(np.argmax(many[0]["sharpe"]))
print("RESULT " + json.dumps(out))
"""


def run_child(devices, K, P, shard, env=None):
    code = CHILD.format(root=ROOT, devices=list(devices), K=K, P=P, shard=shard)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def check_equal_to_one_device(out):
    assert out["terminal_equal"] and out["repeat_equal"], out
    for key in ("n", "n_tail", "var", "x_lo", "x_hi", "min", "max"):          # order statistics and counts: exact
        assert out["exact_" + key], (key, out)
    for key in ("mean", "std", "sharpe", "cvar", "sum_tail"):               # fp64 sums around one pivot: association only
        assert out["rel_" + key] < 1e-12, (key, out)
    assert out["argmax_equal"], out


@pytest.mark.parametrize("K,P", [(1, 100_001), (3, 30_000), (20, 70_000), (1, 5)])
def test_path_sharded_over_distinct_devices_through_rccl(n_devices, K, P):
    """mcp_ctx_create_multi over all visible devices, shard='paths': three all-reduces of the select histograms and one
    all-gather of the records inside the library.  The context must say it went through RCCL (torch's bundled librccl is
    preloaded by Context), and the records must equal the one-device ones."""
    out = run_child(range(n_devices), K, P, "paths")
    assert out["mode"] == "rccl", out            # a 'p2p' here means librccl could not be used: out['note'] says why
    check_equal_to_one_device(out)


def test_path_sharded_over_two_devices_through_the_peer_kernel(n_devices):
    """MCP_EXCHANGE=p2p: the kernel exchange over peer access (device 0 sums its peers' histograms in place and writes the
    totals back), the fallback when librccl cannot be used; the context reports the mode and the reason."""
    out = run_child([0, 1], 3, 50_000, "paths", env={"MCP_EXCHANGE": "p2p"})
    assert out["mode"] == "p2p" and "MCP_EXCHANGE=p2p" in out["note"], out
    check_equal_to_one_device(out)


def test_a_missing_librccl_is_reported_not_hidden(n_devices):
    out = run_child([0, 1], 1, 20_000, "paths", env={"MCP_RCCL_LIB": "/nonexistent/librccl.so"})
    assert out["mode"] in ("rccl", "p2p"), out     # torch's preloaded copy may still resolve; either way the results hold
    if out["mode"] == "p2p":
        assert out["note"], out
    check_equal_to_one_device(out)


def test_portfolio_sharded_over_distinct_devices(n_devices):
    """shard='portfolios' (BASELINE configs[4]): every device walks all paths for its slice of the weights; no exchange is set up."""
    out = run_child(range(n_devices), 1100, 4096, "portfolios")
    assert out["mode"] == "unset", out
    check_equal_to_one_device(out)


def bench(*args, timeout=900):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_one_process_per_gpu_over_nccl_equals_one_gpu(n_devices):
    """bench.py --gpus 2 (its parent starts two ranks; torch.distributed backend nccl): rank g simulates global paths
    [g P, (g+1) P), the select histograms are all-reduced and the records all-gathered per step.  The global statistics must
    equal ONE GPU simulating the same 2 P paths: order statistic and counts exactly, the fp64 aggregates to 1e-12."""
    two = bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--paths-per-gpu", "500000", "--sustain", "0")
    one = bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--paths-per-gpu", "1000000", "--sustain", "0", "--no-cpu-baseline")
    assert two["n_gpus"] == 2 and not two["metric"].startswith("REHEARSAL") and two["config"]["global_paths"] == 1_000_000
    a, b = two["stats"], one["stats"]
    assert a["n"] == b["n"] == 1_000_000 and a["n_tail"] == b["n_tail"] == 50_000
    assert a["var95"] == b["var95"]
    for key in ("mean", "std", "sharpe", "cvar95"):
        assert a[key] == pytest.approx(b[key], rel=1e-12), key
    assert two["value"] > 0.5 * one["value"]            # sanity only: two ranks are not slower than half of one


def test_every_visible_gpu_through_the_bench(n_devices):
    """--gpus N for all visible devices (at most 6 ranks on the card pool's process limit): the line must come back with the
    global path count and a skewed, pipelined schedule."""
    n = min(n_devices, 6)
    out = bench("--gpus", str(n), "--steps", "5", "--warmup", "2", "--sustain", "0")
    assert out["n_gpus"] == n and out["stats"]["n"] == n * 1_000_000 and out["stats"]["n_tail"] == n * 50_000
    assert out["scaling"] == "weak" and out["value"] > 1e8
