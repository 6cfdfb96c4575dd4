"""The N>1 path on CPU: world_size 2 over gloo.

PathEngine.step's choreography is the product code under test; the kernels behind it are replaced by
tests/fake_kernels.py (NumPy + oracle on CPU tensors, same buffer layouts).  Two ranks must produce
exactly the statistics of one rank holding both shards, and both must equal the reference
definitions (oracle/ref_stats.py) applied to the union of the paths.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from monte_carlo_portfolio_amd import synthetic
from monte_carlo_portfolio_amd.engine import PathEngine
from monte_carlo_portfolio_amd.simulate import prepare_inputs
from fake_kernels import FakeKernels
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
N, T, n_local, K = {N}, {T}, {n_local}, {K}
mu, cov = synthetic.synthetic_market(N)
W = synthetic.dirichlet_weights(N, K) if K > 1 else synthetic.equal_weights(N)
mu32, L, W32 = prepare_inputs(mu, cov, W)
eng = PathEngine(mu32, L, W32, T, n_local, device="cpu", kernels=FakeKernels(mu32, L, W32), rf=0.002,
                 group=dist.group.WORLD if world > 1 else None, world_size=world, rank=rank, **{extra})
for i in range({steps}):                      # the LAST step is the one compared (seed 77)
    eng.step(seed=77 + {steps} - 1 - i, path_base=1000)
st = eng.stats()
out = [{{k: (int(st[i][k]) if k in ("n", "n_tail") else float(st[i][k]).hex()) for k in st.dtype.names}} for i in range(K)]
open({out!r} + str(rank), "w").write(json.dumps(out))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(tmp_path, world, N, T, n_local, K, steps=1, extra=None, tag=""):
    out = str(tmp_path / f"res_w{world}{tag}_")
    script = tmp_path / f"worker_w{world}{tag}.py"
    script.write_text(WORKER.format(root=ROOT, N=N, T=T, n_local=n_local, K=K, out=out, steps=steps, extra=repr(extra or {})))
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o.decode()[-3000:]
    return [json.load(open(out + str(r))) for r in range(world)]


@pytest.mark.parametrize("K", [1, 3])
def test_two_ranks_equal_one_rank_and_reference(tmp_path, K):
    from monte_carlo_portfolio_amd import synthetic
    from monte_carlo_portfolio_amd.simulate import prepare_inputs
    from oracle import mc_oracle, ref_stats

    N, T, n_local = 8, 12, 3001          # odd shard size: ragged last block
    two = run_world(tmp_path, 2, N, T, n_local, K)
    one = run_world(tmp_path, 1, N, T, 2 * n_local, K)
    assert two[0] == two[1]                                  # every rank ends with the global statistics
    mu, cov = synthetic.synthetic_market(N)
    W = synthetic.dirichlet_weights(N, K) if K > 1 else synthetic.equal_weights(N)
    mu32, L, W32 = prepare_inputs(mu, cov, W)
    term = mc_oracle.simulate(mu32, L, W32, T, 2 * n_local, 77, path_begin=1000)
    for k in range(K):
        want = ref_stats.path_stats(term[k], rf=0.002)
        a, b = two[0][k], one[0][k]
        assert a["n"] == b["n"] == want["n"] == 2 * n_local
        assert a["n_tail"] == b["n_tail"] == want["n_tail"]
        for key in ("var", "x_lo", "x_hi", "min", "max"):     # order statistics: exact across partitions
            assert a[key] == b[key], key
        assert float.fromhex(a["var"]) == want["var"]
        for key in ("mean", "std", "sharpe", "cvar"):         # fp64 sums: association differs across ranks
            assert float.fromhex(a[key]) == pytest.approx(float.fromhex(b[key]), rel=1e-13)
            assert float.fromhex(a[key]) == pytest.approx(want[key], rel=1e-12)


def test_skewed_schedule_two_ranks_many_steps(tmp_path):
    """The software-pipelined enqueue order (stage s of batch t-s at step t, engine.py) over gloo: seven batches in flight
    through six buffers, collectives of different batches interleaved; the last batch must equal the plain one-rank result."""
    N, T, n_local, K = 8, 12, 1501, 2
    skew = run_world(tmp_path, 2, N, T, n_local, K, steps=7, extra={"skew": True}, tag="s")
    one = run_world(tmp_path, 1, N, T, 2 * n_local, K)
    assert skew[0] == skew[1]
    for k in range(K):
        a, b = skew[0][k], one[0][k]
        for key in ("n", "n_tail", "var", "x_lo", "x_hi", "min", "max"):
            assert a[key] == b[key], key
        for key in ("mean", "std", "sharpe", "cvar"):
            assert float.fromhex(a[key]) == pytest.approx(float.fromhex(b[key]), rel=1e-13)


def test_three_ranks_skewed_with_ragged_shards(tmp_path):
    """world_size 3 (an odd world, shards of 1,001 paths each), skewed schedule, five batches: every rank ends with the
    statistics of the union, equal to one rank holding all 3,003 paths."""
    N, T, n_local, K = 5, 9, 1001, 1
    three = run_world(tmp_path, 3, N, T, n_local, K, steps=5, extra={"skew": True}, tag="t3")
    one = run_world(tmp_path, 1, N, T, 3 * n_local, K)
    assert three[0] == three[1] == three[2]
    a, b = three[0][0], one[0][0]
    for key in ("n", "n_tail", "var", "x_lo", "x_hi", "min", "max"):
        assert a[key] == b[key], key
    for key in ("mean", "std", "sharpe", "cvar"):
        assert float.fromhex(a[key]) == pytest.approx(float.fromhex(b[key]), rel=1e-13)


@pytest.mark.parametrize("shards,skew", [(3, True), (2, False), (8, True)])
def test_logical_shards_equal_one_shard(shards, skew):
    """logical_shards: one process splits its path range over S shards that exchange through the sum kernel and record
    copies (what a one-GPU box can run of the N > 1 choreography); equal to the unsharded engine, also for shards
    without any path (8 shards of 5 paths)."""
    import torch  # noqa: F401
    from fake_kernels import FakeKernels
    from monte_carlo_portfolio_amd import synthetic
    from monte_carlo_portfolio_amd.engine import PathEngine
    from monte_carlo_portfolio_amd.simulate import prepare_inputs
    N, T, K = 6, 10, 3
    n = 5 if shards == 8 else 2999
    mu, cov = synthetic.synthetic_market(N)
    mu32, L, W32 = prepare_inputs(mu, cov, synthetic.dirichlet_weights(N, K))
    ref = PathEngine(mu32, L, W32, T, n, device="cpu", kernels=FakeKernels(mu32, L, W32), rf=0.001)
    ref.step(seed=5)
    want = ref.stats()
    eng = PathEngine(mu32, L, W32, T, n, device="cpu", kernels=FakeKernels(mu32, L, W32), rf=0.001, logical_shards=shards, skew=skew)
    for i in range(9):
        eng.step(seed=5 + 8 - i)
    got = eng.stats()
    assert np.array_equal(eng.terminal(), ref.terminal())
    for key in ("n", "n_tail", "var", "x_lo", "x_hi", "min", "max"):
        assert np.array_equal(got[key], want[key]), key
    for key in ("mean", "std", "sharpe", "cvar"):
        np.testing.assert_allclose(got[key], want[key], rtol=1e-13)
    eng.step(seed=5)                                     # the pipeline keeps working after a drain
    again = eng.stats()
    assert np.array_equal(again["var"], want["var"]) and np.array_equal(again["n_tail"], want["n_tail"])


PF_WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from monte_carlo_portfolio_amd import synthetic
from monte_carlo_portfolio_amd.engine import PathEngine
from monte_carlo_portfolio_amd.simulate import prepare_inputs
from fake_kernels import FakeKernels
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
mu, cov = synthetic.synthetic_market(8)
W = synthetic.dirichlet_weights(8, 5)                      # 5 portfolios over 2 ranks: 3 + 2 (ragged)
mu32, L, W32 = prepare_inputs(mu, cov, W)
kc = 3
Wl = np.zeros((kc, 8), np.float32); sl = W32[rank * kc:(rank + 1) * kc]; Wl[:len(sl)] = sl
eng = PathEngine(mu32, L, W32, 12, 2000, device="cpu", kernels=FakeKernels(mu32, L, Wl), rf=0.002,
                 group=dist.group.WORLD, world_size=world, rank=rank, shard="portfolios")
eng.step(seed=77, path_base=0)
st = eng.gathered_stats()
out = [{{k: (int(st[i][k]) if k in ("n", "n_tail") else float(st[i][k]).hex()) for k in st.dtype.names}} for i in range(len(st))]
open({out!r} + str(rank), "w").write(json.dumps(out))
dist.barrier(); dist.destroy_process_group()
"""


def test_portfolio_sharding_two_ranks(tmp_path):
    """configs[4] sharding: each rank walks all paths for its slice of W; the gathered records must equal the
    reference definitions applied per portfolio (common random numbers across ranks)."""
    from monte_carlo_portfolio_amd import synthetic
    from monte_carlo_portfolio_amd.simulate import prepare_inputs
    from oracle import mc_oracle, ref_stats
    out = str(tmp_path / "pf_")
    script = tmp_path / "pf_worker.py"
    script.write_text(PF_WORKER.format(root=ROOT, out=out))
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o.decode()[-3000:]
    res = [json.load(open(out + str(r))) for r in range(2)]
    assert res[0] == res[1] and len(res[0]) == 5
    mu, cov = synthetic.synthetic_market(8)
    W = synthetic.dirichlet_weights(8, 5)
    mu32, L, W32 = prepare_inputs(mu, cov, W)
    term = mc_oracle.simulate(mu32, L, W32, 12, 2000, 77)
    for k in range(5):
        want = ref_stats.path_stats(term[k], rf=0.002)
        assert res[0][k]["n"] == 2000 and res[0][k]["n_tail"] == want["n_tail"]
        assert float.fromhex(res[0][k]["var"]) == want["var"]
        assert float.fromhex(res[0][k]["sharpe"]) == pytest.approx(want["sharpe"], rel=1e-12)
