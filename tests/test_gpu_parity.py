"""GPU parity tests proper: the HIP path, through the C ABI, against the CPU oracle.

Bar (BASELINE.json north_star): fp32 results within 1e-6 relative; path counts exact.  The exact
math mode does better than that: terminal values are required to be BIT-IDENTICAL to the oracle,
VaR bit-identical to np.percentile on the oracle's values, fp64 moments within 1e-12.
"""
import numpy as np
import pytest

from monte_carlo_portfolio_amd import _ffi, simulate_paths, synthetic
from monte_carlo_portfolio_amd.simulate import prepare_inputs
from oracle import mc_oracle, ref_stats

pytestmark = pytest.mark.gpu

SEED = synthetic.BENCH_SEED


def run_both(N, T, P, K=1, compounding="simple", v0=1.0, path_begin=0, native=False, alpha=0.95, rf=0.0, seed=SEED):
    mu, cov = synthetic.synthetic_market(N)
    W = synthetic.equal_weights(N) if K == 1 else synthetic.dirichlet_weights(N, K)
    got = simulate_paths(mu, cov, W, n_steps=T, n_paths=P, seed=seed, v0=v0, compounding=compounding,
                         rf=rf, alpha=alpha, store=True, path_begin=path_begin, native_math=native)
    mu32, L, W32 = prepare_inputs(mu, cov, W)
    ref = mc_oracle.simulate(mu32, L, W32, T, P, seed, path_begin=path_begin, v0=v0, compounding=compounding)
    return ([got] if K == 1 else got), ref


def assert_stats(got, ref_terminal, v0, compounding, alpha, rf, exact_quantile=True):
    want = ref_stats.path_stats(ref_terminal, v0, compounding, alpha, rf)
    assert got["n"] == want["n"] and got["n_tail"] == want["n_tail"]          # integer counts: exact
    if exact_quantile:
        assert got["var"] == want["var"]                                        # bit-exact np.percentile
        assert got["min"] == want["min"] and got["max"] == want["max"]
    else:
        assert got["var"] == pytest.approx(want["var"], rel=1e-12)
    for key in ("mean", "std", "sharpe", "cvar", "sum_tail"):
        assert got[key] == pytest.approx(want[key], rel=1e-12, abs=1e-15), key


@pytest.mark.parametrize("N,T,P", [(16, 252, 20_000), (3, 252, 10_000), (1, 30, 1000), (4, 7, 257),
                                   (5, 17, 1000), (13, 40, 3000), (32, 25, 2000), (64, 20, 2000), (61, 9, 700)])
def test_terminal_values_bit_exact(gpu_ctx, N, T, P):
    got, ref = run_both(N, T, P)
    V = got[0]["terminal"]
    assert V.dtype == np.float32 and V.shape == (P,)
    assert np.array_equal(V.view(np.uint32), ref[0].view(np.uint32))
    assert_stats(got[0], ref[0], 1.0, "simple", 0.95, 0.0)


def test_every_asset_count_bit_exact(gpu_ctx):
    for N in range(1, 65):
        got, ref = run_both(N, 6, 300)
        assert np.array_equal(got[0]["terminal"].view(np.uint32), ref[0].view(np.uint32)), N


@pytest.mark.parametrize("K", [2, 8, 9, 21])
def test_multi_portfolio_common_random_numbers(gpu_ctx, K):
    got, ref = run_both(16, 30, 5000, K=K, rf=0.001)
    for k in range(K):
        assert np.array_equal(got[k]["terminal"].view(np.uint32), ref[k].view(np.uint32)), k
        assert_stats(got[k], ref[k], 1.0, "simple", 0.95, 0.001)
    sharpe = np.array([g["sharpe"] for g in got])
    want = np.array([ref_stats.path_stats(ref[k], rf=0.001)["sharpe"] for k in range(K)])
    assert int(np.argmax(sharpe)) == int(np.argmax(want))        # argmax index: exact


def test_log_compounding_and_v0(gpu_ctx):
    got, ref = run_both(8, 50, 4000, compounding="log")
    assert np.array_equal(got[0]["terminal"].view(np.uint32), ref[0].view(np.uint32))
    assert_stats(got[0], ref[0], 1.0, "log", 0.95, 0.0, exact_quantile=False)
    got, ref = run_both(8, 50, 4000, v0=10000.0, alpha=0.99, rf=0.01)
    assert np.array_equal(got[0]["terminal"].view(np.uint32), ref[0].view(np.uint32))
    assert_stats(got[0], ref[0], 10000.0, "simple", 0.99, 0.01)


def test_path_offsets_partition_invariance(gpu_ctx):
    """Counter-based RNG on the global path id: 4 shards == one run; also across the 2^32 boundary."""
    mu, cov = synthetic.synthetic_market(16)
    w = synthetic.equal_weights(16)
    whole = simulate_paths(mu, cov, w, n_steps=20, n_paths=4096, seed=7, store=True)["terminal"]
    parts = np.concatenate([simulate_paths(mu, cov, w, n_steps=20, n_paths=1024, seed=7, store=True,
                                           path_begin=1024 * g)["terminal"] for g in range(4)])
    assert np.array_equal(whole, parts)
    got, ref = run_both(16, 12, 1000, path_begin=(1 << 32) - 500)
    assert np.array_equal(got[0]["terminal"].view(np.uint32), ref[0].view(np.uint32))
    got, ref = run_both(4, 12, 600, seed=0xFEDCBA9876543210)
    assert np.array_equal(got[0]["terminal"].view(np.uint32), ref[0].view(np.uint32))


@pytest.mark.parametrize("P", [1, 2, 3, 20, 21, 255, 256, 257, 513])
def test_tiny_and_ragged_path_counts(gpu_ctx, P):
    got, ref = run_both(16, 10, P)
    assert np.array_equal(got[0]["terminal"].view(np.uint32), ref[0].view(np.uint32))
    assert_stats(got[0], ref[0], 1.0, "simple", 0.95, 0.0)


def test_zero_steps_and_degenerate_sigma(gpu_ctx):
    mu, cov = synthetic.synthetic_market(4)
    r = simulate_paths(mu, cov, np.ones(4) / 4, n_steps=0, n_paths=100, seed=1, store=True)
    assert np.all(r["terminal"] == 1.0) and r["std"] == 0.0 and r["sharpe"] == 0.0 and r["var"] == 0.0
    # zero covariance via an explicit zero factor: deterministic growth, all paths identical (ties)
    r = simulate_paths(np.full(4, 1e-3), None, np.ones(4) / 4, n_steps=10, n_paths=1000, seed=1, store=True,
                       chol=np.zeros((4, 4)))
    assert np.all(r["terminal"] == r["terminal"][0]) and r["n_tail"] == 1000 and r["var"] == r["cvar"] == r["min"]


def test_native_math_is_statistically_equivalent(gpu_ctx):
    """MCP_FLAG_NATIVE_MATH draws its normals by hardware Box-Muller instead of the spec's inverse-CDF table: other
    values from the same Philox words, same distribution.  Aggregates must agree within Monte-Carlo error."""
    N, T, P = 16, 252, 400_000
    mu, cov = synthetic.synthetic_market(N)
    w = synthetic.equal_weights(N)
    a = simulate_paths(mu, cov, w, n_steps=T, n_paths=P, seed=SEED)
    b = simulate_paths(mu, cov, w, n_steps=T, n_paths=P, seed=SEED, native_math=True)
    se = a["std"] / np.sqrt(P)
    assert a["n"] == b["n"] == P and b["n_tail"] == a["n_tail"]
    assert abs(a["mean"] - b["mean"]) < 6 * se * np.sqrt(2)
    assert abs(a["std"] - b["std"]) < 9 * a["std"] / np.sqrt(P)
    assert abs(a["var"] - b["var"]) < 25 * se                  # quantile s.e. ~ sqrt(p(1-p)/n)/f(q) ~ 2.1 se at p = 5 %
    assert abs(a["cvar"] - b["cvar"]) < 30 * se


def test_config1_scale_properties(gpu_ctx):
    """BASELINE config 1 at full size (16 assets, 1M paths, 252 steps): size-independent properties.
    The first 50k paths are checked bit-exactly against the oracle; the rest through the statistics:
    the lognormal-ish mean must match the analytic drift, n_tail = floor((n-1)q)+1, sortedness of the
    order statistics, and shard-sum consistency of the moments."""
    N, T, P = 16, 252, 1_000_000
    mu, cov = synthetic.synthetic_market(N)
    w = synthetic.equal_weights(N)
    r = simulate_paths(mu, cov, w, n_steps=T, n_paths=P, seed=SEED, store=True)
    V = r["terminal"]
    mu32, L, W32 = prepare_inputs(mu, cov, w)
    ref = mc_oracle.simulate(mu32, L, W32, T, 50_000, SEED)
    assert np.array_equal(V[:50_000].view(np.uint32), ref[0].view(np.uint32))
    x = V.astype(np.float64) - 1.0
    assert r["n"] == P and r["n_tail"] == 50_000
    assert r["var"] == np.percentile(x, (1 - 0.95) * 100)
    assert r["x_lo"] <= r["var"] <= r["x_hi"] and r["min"] <= r["x_lo"] and r["x_hi"] <= r["max"]
    assert r["mean"] == pytest.approx(x.mean(), rel=1e-12) and r["std"] == pytest.approx(x.std(ddof=1), rel=1e-12)
    analytic = (1.0 + float(w @ mu)) ** T - 1.0                 # E[prod(1+rho_t)], iid steps
    assert abs(r["mean"] - analytic) < 5 * r["std"] / np.sqrt(P)
    halves = [simulate_paths(mu, cov, w, n_steps=T, n_paths=P // 2, seed=SEED, path_begin=g * (P // 2)) for g in range(2)]
    assert halves[0]["mean"] * 0.5 + halves[1]["mean"] * 0.5 == pytest.approx(r["mean"], rel=1e-13)


def test_path_engine_single_rank_matches_host_level(gpu_ctx):
    """The torch-plumbed device pipeline (engine.PathEngine + HipKernels) == mcp_simulate."""
    from monte_carlo_portfolio_amd.engine import PathEngine
    N, T, P, K = 16, 40, 30_000, 3
    mu, cov = synthetic.synthetic_market(N)
    W = synthetic.dirichlet_weights(N, K)
    mu32, L, W32 = prepare_inputs(mu, cov, W)
    eng = PathEngine(mu32, L, W32, T, P, rf=0.001)
    eng.step(SEED, path_base=500)
    st = eng.stats()
    host = simulate_paths(mu, cov, W, n_steps=T, n_paths=P, seed=SEED, rf=0.001, path_begin=500, store=True)
    term = eng.terminal()
    for k in range(K):
        assert np.array_equal(term[k], host[k]["terminal"])
        for key in st.dtype.names:
            assert st[k][key] == host[k][key], key


@pytest.mark.parametrize("kw", [dict(logical_shards=2), dict(logical_shards=3, skew=True), dict(logical_shards=2, skew=True, n_stats_streams=1),
                                dict(logical_shards=8), dict(cu_reserve=4), dict(skew=True), dict(n_buffers=3, n_stats_streams=3)])
@pytest.mark.parametrize("K", [1, 20])
def test_path_engine_layouts_equal_the_plain_engine(gpu_ctx, kw, K):
    """Every stream / schedule / shard layout of PathEngine on the real kernels: logical shards exchanging through the sum
    kernel (the N > 1 choreography on one GPU), the skewed enqueue order, one or several statistics streams, CU-masked path
    streams.  Nine pipelined batches over a cycle of seeds; the last must equal the plain engine's result for its seed:
    terminal values, counts and order statistics exactly, fp64 sums to 1e-13."""
    from monte_carlo_portfolio_amd.engine import PathEngine
    N, T, P = 16, 24, 30_011
    mu, cov = synthetic.synthetic_market(N)
    W = synthetic.equal_weights(N) if K == 1 else synthetic.dirichlet_weights(N, K)
    mu32, L, W32 = prepare_inputs(mu, cov, W)
    ref = PathEngine(mu32, L, W32, T, P, rf=0.001, pipeline=False)
    ref.step(SEED)
    want, want_term = ref.stats(), ref.terminal()
    eng = PathEngine(mu32, L, W32, T, P, rf=0.001, **kw)
    for i in range(9):
        eng.step(SEED + 8 - i)
    got = eng.stats()
    assert np.array_equal(eng.terminal().view(np.uint32), want_term.view(np.uint32))
    for key in ("n", "n_tail", "var", "x_lo", "x_hi", "min", "max"):
        assert np.array_equal(got[key], want[key]), key
    for key in ("mean", "std", "sharpe", "cvar"):
        np.testing.assert_allclose(got[key], want[key], rtol=1e-13)
    eng.step(SEED)
    assert np.array_equal(eng.stats()["var"], want["var"])
    eng.close()


ENGINE_WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from monte_carlo_portfolio_amd import synthetic
from monte_carlo_portfolio_amd.engine import PathEngine
from monte_carlo_portfolio_amd.simulate import prepare_inputs
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)     # both ranks share the box's one GPU
mu, cov = synthetic.synthetic_market(16)
W = synthetic.dirichlet_weights(16, 2)
mu32, L, W32 = prepare_inputs(mu, cov, W)
eng = PathEngine(mu32, L, W32, 30, 20_001, rf=0.001, group=dist.group.WORLD, world_size=world, rank=rank)
eng.step(seed=99, path_base=0)
st = eng.stats()
out = [{{k: (int(st[i][k]) if k in ("n", "n_tail") else float(st[i][k]).hex()) for k in st.dtype.names}} for i in range(2)]
open({out!r} + str(rank), "w").write(json.dumps(out))
dist.barrier(); dist.destroy_process_group()
"""


def test_path_engine_two_ranks_on_one_gpu(gpu_ctx, tmp_path):
    """HIP kernels + the collective choreography together: 2 processes (gloo carrying the CUDA
    buffers) sharing this box's single MI355X, against one process holding both shards."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "res_")
    script = tmp_path / "worker.py"
    script.write_text(ENGINE_WORKER.format(root=root, out=out))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o.decode()[-3000:]
    two = [json.load(open(out + str(r))) for r in range(2)]
    assert two[0] == two[1]
    mu, cov = synthetic.synthetic_market(16)
    W = synthetic.dirichlet_weights(16, 2)
    host = simulate_paths(mu, cov, W, n_steps=30, n_paths=40_002, seed=99, rf=0.001)
    for k in range(2):
        assert two[0][k]["n"] == 40_002 and two[0][k]["n_tail"] == host[k]["n_tail"]
        for key in ("var", "x_lo", "x_hi", "min", "max"):
            assert float.fromhex(two[0][k][key]) == host[k][key], key
        for key in ("mean", "std", "sharpe", "cvar"):
            assert float.fromhex(two[0][k][key]) == pytest.approx(host[k][key], rel=1e-13), key


@pytest.mark.parametrize("N,K,P,T,comp", [(16, 17, 4000, 30, "simple"), (16, 130, 3000, 25, "simple"), (16, 300, 1100, 12, "log"),
                                          (12, 64, 2049, 20, "simple"), (3, 40, 1000, 15, "simple"),
                                          (16, 384, 300, 10, "simple"), (16, 700, 257, 8, "log"), (10, 513, 130, 6, "simple"),
                                          (4, 400, 100, 9, "simple"), (32, 40, 700, 9, "simple"), (64, 300, 200, 5, "log"),
                                          (20, 17, 1000, 11, "simple"), (61, 257, 130, 4, "simple")])
def test_mfma_sweep_kernel_bit_exact(gpu_ctx, N, K, P, T, comp):
    """K >= 17, N <= 16 runs mc_sweep_kernel (W.r on v_mfma_f32_32x32x2_f32): same bits as the oracle's fma chain."""
    got, ref = run_both(N, T, P, K=K, compounding=comp, rf=0.0005)
    for k in range(K):
        assert np.array_equal(got[k]["terminal"].view(np.uint32), ref[k].view(np.uint32)), k
    for k in (0, K // 2, K - 1):
        assert_stats(got[k], ref[k], 1.0, comp, 0.95, 0.0005, exact_quantile=(comp == "simple"))
    sharpe = np.array([g["sharpe"] for g in got])
    want = np.array([ref_stats.path_stats(ref[k], compounding=comp, rf=0.0005)["sharpe"] for k in range(K)])
    assert int(np.argmax(sharpe)) == int(np.argmax(want))


# Every branch of the sweep dispatch (csrc/mcp_api.cpp: sweep_plan), each against the oracle bit for bit: whole shared-draw
# workgroups (512 / 256 portfolios at N <= 16, 256 / 128 beyond), then the remainder on the smallest tiling that covers it.
SWEEP_CASES = [
    # N <= 16: per-wave kernel with 1 / 2 / 4 tiles (K <= 128), shared MT 2 (129..256), shared MT 2 + per-wave (257..384), shared MT 4
    (16, 17), (16, 33), (16, 65), (16, 128), (16, 129), (16, 256), (16, 257), (16, 300), (16, 384), (16, 385), (16, 512),
    # ... behind whole 512-portfolio workgroups: remainders 1, 8, 17, 128, 226 (configs[4] over 8 GPUs: K = 1,250), 256, 288, 452
    (16, 513), (16, 520), (16, 529), (16, 640), (16, 1250), (16, 768), (16, 800), (16, 964),
    (5, 700), (12, 1100),
    # 16 < N <= 64: shared MT 1 (128) / MT 2 (256), and the remainder behind whole 256-portfolio workgroups
    (20, 17), (20, 128), (20, 129), (20, 256), (20, 300), (33, 513), (64, 130),
]


@pytest.mark.parametrize("N,K", SWEEP_CASES)
@pytest.mark.parametrize("mode", ["simple", "log", "native"])
def test_sweep_dispatch_every_branch(gpu_ctx, N, K, mode):
    if mode != "simple" and (K, N) not in ((300, 16), (520, 16), (1250, 16), (129, 20), (300, 20), (128, 16)):
        pytest.skip("the other compounding / math modes on a subset of the branches")
    T, P = 6, 333                                     # ragged: 333 paths = five 64-path wave tiles + 13
    comp = "log" if mode == "log" else "simple"
    if mode == "native":                              # hardware Box-Muller: other values, so the check is internal consistency
        mu, cov = synthetic.synthetic_market(N)
        W = synthetic.dirichlet_weights(N, K)
        a = simulate_paths(mu, cov, W, n_steps=T, n_paths=P, seed=SEED, store=True, native_math=True, as_array=True)
        b = [simulate_paths(mu, cov, W[k], n_steps=T, n_paths=P, seed=SEED, store=True, native_math=True) for k in (0, K // 2, K - 1)]
        for i, k in enumerate((0, K // 2, K - 1)):    # the K-portfolio kernels against the one-portfolio kernel
            np.testing.assert_allclose(a[1][k], b[i]["terminal"], rtol=2e-6)
            assert a[0][k]["n_tail"] == b[i]["n_tail"]
        return
    got, ref = run_both(N, T, P, K=K, compounding=comp, rf=0.0005)
    for k in range(K):
        assert np.array_equal(got[k]["terminal"].view(np.uint32), ref[k].view(np.uint32)), k
    for k in (0, 1, K // 3, K // 2, K - 2, K - 1):
        assert_stats(got[k], ref[k], 1.0, comp, 0.95, 0.0005, exact_quantile=(comp == "simple"))


def test_config0_csv_to_simulated_stats(gpu_ctx):
    """BASELINE configs[0] end to end: the three daily CSVs -> returns -> (mu, Sigma) -> 10k paths x 252 steps,
    against the oracle on the same inputs (the reference itself cannot load these files)."""
    import io, os
    from monte_carlo_portfolio_amd import ingest
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
    files = []
    for n in ("Bitcoin Historical Data.csv", "Ethereum Historical Data.csv", "XAU_USD Historical Data.csv"):
        b = io.BytesIO(open(os.path.join(data, n), "rb").read()); b.name = n; files.append(b)
    names, prices, res = ingest.load_prices(files, resample_rule="D", report=lambda m: None)
    rets = ingest.returns_matrix(res)
    mu, cov = rets.mean().values, rets.cov().values            # per step (daily), app.py:679-680 before annualising
    w = np.ones(3) / 3
    got = simulate_paths(mu, cov, w, n_steps=252, n_paths=10_000, seed=SEED, store=True, v0=10000.0, rf=0.03)
    mu32, L, W32 = prepare_inputs(mu, cov, w)
    ref = mc_oracle.simulate(mu32, L, W32, 252, 10_000, SEED, v0=10000.0)
    assert np.array_equal(got["terminal"].view(np.uint32), ref[0].view(np.uint32))
    assert_stats(got, ref[0], 10000.0, "simple", 0.95, 0.03)


def test_simulated_sweep_optimum(gpu_ctx):
    """configs[4] shape, scaled: Dirichlet weights as the reference draws them, all on common random numbers;
    the max-Sharpe / min-VaR indices must equal the ones computed from the oracle's terminal values."""
    from monte_carlo_portfolio_amd import simulate_sweep
    mu, cov = synthetic.synthetic_market(16)
    K, P, T = 200, 4000, 40
    out = {m: simulate_sweep(mu, cov, n_portfolios=K, n_steps=T, n_paths=P, seed=SEED, rf=0.001, method=m, np_seed=7)
           for m in ("Monte Carlo", "VaR", "CVaR")}
    W = out["Monte Carlo"]["all_weights"]
    np.random.seed(7)
    assert np.array_equal(W, np.array([np.random.dirichlet(np.ones(16), size=1)[0] for _ in range(K)]))
    mu32, L, W32 = prepare_inputs(mu, cov, W)
    ref = mc_oracle.simulate(mu32, L, W32, T, P, SEED)
    st = [ref_stats.path_stats(ref[k], rf=0.001) for k in range(K)]
    assert out["Monte Carlo"]["opt_idx"] == int(np.argmax([s["sharpe"] for s in st]))
    assert out["VaR"]["opt_idx"] == int(np.argmin([-s["var"] for s in st]))
    assert out["CVaR"]["opt_idx"] == int(np.argmin([-s["cvar"] for s in st]))
    assert np.array_equal(out["VaR"]["stats"]["var"], np.array([s["var"] for s in st]))


def test_device_normals_match_oracle_everywhere(gpu_ctx):
    """The device normal generator against the oracle's, bit for bit: both ends of every (octave, bin) of the table,
    u == 1/2, the deepest tail, and 2^24 random words."""
    import ctypes
    import torch
    L = _ffi.lib()
    rng = np.random.default_rng(11)
    # words whose u lands on every bin boundary: v = round((1 + j/32) 2^(E-127) 2^32 - 1/2) +- 2
    E, j = np.meshgrid(np.arange(94, 127), np.arange(32), indexing="ij")
    edge_u = (2.0 ** (E - 127.0) * (1.0 + j / 32.0)).ravel()
    v = np.clip(np.round(edge_u * 2.0 ** 32 - 0.5), 0, 2 ** 31 - 1).astype(np.int64)
    near = np.concatenate([np.clip(v + d, 0, 2 ** 31 - 1) for d in (-2, -1, 0, 1, 2)]).astype(np.uint32)
    x = np.concatenate([near, near | np.uint32(0x80000000), np.arange(0, 70000, dtype=np.uint32),
                        rng.integers(0, 2 ** 32, 1 << 24, dtype=np.uint32)]).astype(np.uint32)
    d_x = torch.from_numpy(x.view(np.int32)).cuda()
    z = torch.empty(x.size, dtype=torch.float32, device="cuda")
    _ffi.check(L.mcp_launch_normals(d_x.data_ptr(), x.size, z.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert np.array_equal(z.cpu().numpy().view(np.uint32), mc_oracle.normals(x).view(np.uint32))


def test_pipelined_engine_batches_are_independent_and_correct(gpu_ctx):
    """Two-stream double-buffered PathEngine: statistics of batch i overlap the path kernel of batch i+1; every
    batch must still equal the plain host-level call for its seed (no buffer is clobbered early)."""
    from monte_carlo_portfolio_amd.engine import PathEngine
    N, T, P = 16, 60, 50_000
    mu, cov = synthetic.synthetic_market(N)
    w = synthetic.equal_weights(N)
    mu32, L, W32 = prepare_inputs(mu, cov, w)
    eng = PathEngine(mu32, L, W32, T, P)
    assert eng.pipeline and eng.n_buf == 2
    eng4 = PathEngine(mu32, L, W32, T, P, n_buffers=4)
    seeds = [11, 12, 13, 14, 15]
    for s in seeds:                       # back to back, no host sync in between
        eng.step(s)
    eng.synchronize()
    want = {s: simulate_paths(mu, cov, w, n_steps=T, n_paths=P, seed=s) for s in seeds[-2:]}
    for back, s in ((0, seeds[-1]), (1, seeds[-2])):       # the two batches still resident in the double buffer
        b = eng.bufs[(eng.last - back) % eng.n_buf][0]
        raw = b["ws"][_ffi.WS_STATS].cpu().numpy().view(np.uint8)[:_ffi.STATS_DTYPE.itemsize].view(_ffi.STATS_DTYPE)[0]
        for key in raw.dtype.names:
            assert raw[key] == want[s][key], (s, key)
    plain = PathEngine(mu32, L, W32, T, P, pipeline=False)
    plain.step(seeds[-1])
    assert plain.stats()[0] == eng.stats()[0]
    for s in seeds + [16, 17, 18, seeds[-1]]:             # 9 batches through 4 buffers
        eng4.step(s)
    assert eng4.n_buf == 4 and eng4.stats()[0] == plain.stats()[0]


def test_config3_shape_64_assets_1260_steps(gpu_ctx):
    """BASELINE configs[3] shape (64 assets, 1260 steps) on 20k paths: first 1,500 paths bit-exact against the
    oracle, the rest through size-independent properties (partition invariance, analytic mean)."""
    N, T, P = 64, 1260, 20_000
    mu, cov = synthetic.synthetic_market(N)
    w = synthetic.equal_weights(N)
    r = simulate_paths(mu, cov, w, n_steps=T, n_paths=P, seed=SEED, store=True)
    mu32, L, W32 = prepare_inputs(mu, cov, w)
    ref = mc_oracle.simulate(mu32, L, W32, T, 1500, SEED)
    assert np.array_equal(r["terminal"][:1500].view(np.uint32), ref[0].view(np.uint32))
    tail = simulate_paths(mu, cov, w, n_steps=T, n_paths=5000, seed=SEED, store=True, path_begin=15_000)["terminal"]
    assert np.array_equal(tail, r["terminal"][15_000:])
    analytic = (1.0 + float(w @ mu)) ** T - 1.0
    assert abs(r["mean"] - analytic) < 5 * r["std"] / np.sqrt(P)
    x = r["terminal"].astype(np.float64) - 1.0
    assert r["var"] == np.percentile(x, (1 - 0.95) * 100) and r["n_tail"] == int((x <= r["var"]).sum())


def test_config2_shard_scale_properties(gpu_ctx):
    """BASELINE configs[2]: 100M paths over 8 GPUs = 12.5M per GPU.  One shard at full size: moments and order
    statistics against NumPy on the downloaded values, and consistency of two half-shards with the whole."""
    N, T, P = 16, 252, 12_500_000
    mu, cov = synthetic.synthetic_market(N)
    w = synthetic.equal_weights(N)
    r = simulate_paths(mu, cov, w, n_steps=T, n_paths=P, seed=SEED, store=True, path_begin=3 * P)
    x = r["terminal"].astype(np.float64) - 1.0
    assert r["n"] == P and r["var"] == np.percentile(x, (1 - 0.95) * 100)
    assert r["n_tail"] == int((x <= r["var"]).sum()) == 625_000
    assert r["mean"] == pytest.approx(x.mean(), rel=1e-12) and r["std"] == pytest.approx(x.std(ddof=1), rel=1e-11)
    assert r["min"] == x.min() and r["max"] == x.max()
    mu32, L, W32 = prepare_inputs(mu, cov, w)
    ref = mc_oracle.simulate(mu32, L, W32, T, 4096, SEED, path_begin=3 * P + P - 4096)      # the shard's last paths
    assert np.array_equal(r["terminal"][-4096:].view(np.uint32), ref[0].view(np.uint32))


def test_enqueue_only_steps_capture_into_a_hip_graph(gpu_ctx):
    """The device-level entry points only enqueue (no allocation, no sync, no host copy): a whole pass captured into
    a hipGraph through torch's capture stream replays to the same bits."""
    import torch
    from monte_carlo_portfolio_amd.engine import PathEngine
    N, T, P = 16, 30, 40_000
    mu, cov = synthetic.synthetic_market(N)
    mu32, L, W32 = prepare_inputs(mu, cov, synthetic.dirichlet_weights(N, 3))
    eng = PathEngine(mu32, L, W32, T, P, pipeline=False)
    eng.step(SEED)                                  # warm-up: uploads the per-device inverse-CDF table outside capture
    want = eng.stats()
    term = eng.terminal().copy()
    eng.d_terminal.zero_()
    eng.ws[_ffi.WS_STATS].zero_()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.step(SEED)
    for _ in range(3):
        eng.d_terminal.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(eng.terminal(), term)
        got = eng.stats()
        for key in want.dtype.names:
            assert np.array_equal(got[key], want[key]), key


def test_context_is_thread_safe(gpu_ctx):
    """Streamlit runs one script thread per session: concurrent calls on the shared context must
    serialise inside the library and each return its own, correct result."""
    import threading
    mu, cov = synthetic.synthetic_market(8)
    w = synthetic.equal_weights(8)
    seeds = list(range(100, 108))
    want = {s: simulate_paths(mu, cov, w, n_steps=20, n_paths=20_000 + s, seed=s) for s in seeds}
    got, errs = {}, []

    def work(s):
        try:
            for _ in range(3):
                got[s] = simulate_paths(mu, cov, w, n_steps=20, n_paths=20_000 + s, seed=s)
        except Exception as e:      # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(s,)) for s in seeds]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs
    for s in seeds:
        assert got[s] == want[s]


def test_extreme_inputs_negative_values_and_alpha_range(gpu_ctx):
    """Huge per-step volatility drives 1 + rho below zero: terminal values of both signs, zeros and wide dynamic
    range go through the order-preserving key of the radix select; alpha near both ends; one-step walks."""
    rng = np.random.default_rng(5)
    N = 4
    A = rng.normal(size=(N, N))
    cov = A @ A.T * 0.3 + np.eye(N) * 0.05                      # ~60-100 % vol PER STEP
    mu = np.full(N, 0.01)
    w = np.array([0.7, 0.1, 0.1, 0.1])
    mu32, L, W32 = prepare_inputs(mu, cov, w)
    for T, P, alpha in ((9, 30_000, 0.95), (1, 5000, 0.999), (3, 20_001, 0.5), (25, 10_000, 0.9)):
        got = simulate_paths(mu, cov, w, n_steps=T, n_paths=P, seed=SEED + T, store=True, alpha=alpha, rf=0.01)
        ref = mc_oracle.simulate(mu32, L, W32, T, P, SEED + T)
        assert np.array_equal(got["terminal"].view(np.uint32), ref[0].view(np.uint32))
        if T > 1:
            assert (ref[0] < 0).any() and (ref[0] > 0).any()
        assert_stats(got, ref[0], 1.0, "simple", alpha, 0.01)


def test_randomised_configurations_bit_exact(gpu_ctx):
    """Forty random problems (assets, steps, paths, portfolios, seeds, offsets, compounding): every terminal value
    from every kernel variant (thread-per-path, KT = 8, MFMA per-wave, MFMA shared-draw) equals the oracle's."""
    rng = np.random.default_rng(2025)
    for it in range(40):
        N = int(rng.integers(1, 65))
        K = int(rng.choice([1, 1, 2, 5, 16, 17, 33, 100, 384, 520]))
        T = int(rng.integers(1, 24))
        P = int(rng.integers(1, 1500))
        comp = "log" if rng.random() < 0.3 else "simple"
        seed = int(rng.integers(0, 2 ** 63))
        pb = int(rng.integers(0, 2 ** 40))
        mu, cov = synthetic.synthetic_market(N, rng_seed=int(rng.integers(1, 1000)))
        W = np.random.RandomState(it).dirichlet(np.ones(N), K)
        got = simulate_paths(mu, cov, W if K > 1 else W[0], n_steps=T, n_paths=P, seed=seed, path_begin=pb, store=True,
                             compounding=comp, as_array=True)
        mu32, L, W32 = prepare_inputs(mu, cov, W)
        ref = mc_oracle.simulate(mu32, L, W32, T, P, seed, path_begin=pb, compounding=comp)
        assert np.array_equal(got[1].view(np.uint32), ref.view(np.uint32)), (it, N, K, T, P, comp)
        k = int(rng.integers(0, K))
        want = ref_stats.path_stats(ref[k], compounding=comp)
        assert got[0][k]["n"] == P and got[0][k]["n_tail"] == want["n_tail"]
        assert got[0][k]["var"] == pytest.approx(want["var"], rel=1e-12, abs=1e-15)


def test_normal_transform_integrates_to_unit_variance(gpu_ctx):
    """A regular grid of 2^27 words per sign (every 16th value of v) pushed through the device transform is a
    midpoint quadrature of the moments of N(0,1): mean 0 (exact by symmetry), variance 1, fourth moment 3."""
    import ctypes
    import torch
    n = 1 << 27
    v = torch.arange(n, dtype=torch.int64, device="cuda") * 16 + 8
    z = torch.empty(n, dtype=torch.float32, device="cuda")
    _ffi.check(_ffi.lib().mcp_launch_normals(v.to(torch.int32).data_ptr(), n, z.data_ptr(),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    z64 = z.double()
    m2 = float((z64 * z64).mean())
    m4 = float((z64 ** 4).mean())
    assert float(z.min()) >= 0.0 and abs(m2 - 1.0) < 2e-6 and abs(m4 - 3.0) < 3e-5     # positive half; the other is its mirror


PF_ENGINE_WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from monte_carlo_portfolio_amd import synthetic
from monte_carlo_portfolio_amd.engine import PathEngine
from monte_carlo_portfolio_amd.simulate import prepare_inputs
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
mu, cov = synthetic.synthetic_market(16)
W = synthetic.dirichlet_weights(16, 45)                  # 45 portfolios over 2 ranks: 23 + 22
mu32, L, W32 = prepare_inputs(mu, cov, W)
eng = PathEngine(mu32, L, W32, 20, 6000, rf=0.001, group=dist.group.WORLD, world_size=world, rank=rank,
                 shard="portfolios", pipeline=False)
eng.step(seed=321)
st = eng.gathered_stats()
np.save({out!r} + str(rank) + ".npy", st)
dist.barrier(); dist.destroy_process_group()
"""


def test_portfolio_sharded_engine_two_ranks_on_one_gpu(gpu_ctx, tmp_path):
    """configs[4] sharding with the real kernels: each of two processes walks all paths for its slice of W (MFMA
    sweep kernel), one all_gather of the records; must equal one process scoring all 45 portfolios."""
    import os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "pf_")
    script = tmp_path / "pf_worker.py"
    script.write_text(PF_ENGINE_WORKER.format(root=root, out=out))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o.decode()[-3000:]
    a, b = np.load(out + "0.npy"), np.load(out + "1.npy")
    assert a.shape == (45,) and np.array_equal(a, b)
    mu, cov = synthetic.synthetic_market(16)
    W = synthetic.dirichlet_weights(16, 45)
    want = simulate_paths(mu, cov, W, n_steps=20, n_paths=6000, seed=321, rf=0.001, as_array=True)
    for key in want.dtype.names:
        assert np.array_equal(a[key], want[key]), key


@pytest.mark.parametrize("N,T,P,comp", [(16, 252, 20_000, "simple"), (3, 40, 5000, "simple"), (37, 12, 2000, "log"), (64, 9, 1000, "simple")])
def test_folded_fast_path(gpu_ctx, N, T, P, comp):
    """MCP_FLAG_FOLD (SPEC.md 4.1): rho = w.mu + (L^T w).z, one portfolio.  Bit-identical to the oracle's folded mode;
    against the unfolded recurrence only the rounding differs (same normals): terminal values within 2e-6 relative,
    statistics within north_star's 1e-6."""
    mu, cov = synthetic.synthetic_market(N)
    w = synthetic.dirichlet_weights(N, 1, seed=N)[0]
    got = simulate_paths(mu, cov, w, n_steps=T, n_paths=P, seed=SEED, store=True, compounding=comp, fold=True)
    mu32, L, W32 = prepare_inputs(mu, cov, w)
    ref_fold = mc_oracle.simulate(mu32, L, W32, T, P, SEED, compounding=comp, fold=True)
    assert np.array_equal(got["terminal"].view(np.uint32), ref_fold[0].view(np.uint32))
    ref = mc_oracle.simulate(mu32, L, W32, T, P, SEED, compounding=comp)
    diff = np.abs(got["terminal"].astype(np.float64) - ref[0])
    assert np.max(diff / (1.0 if comp == "log" else np.abs(ref[0]))) < (2e-7 if comp == "log" else 2e-6)   # S_T absolute, V_T relative
    want = ref_stats.path_stats(ref[0], compounding=comp)
    assert got["n_tail"] == want["n_tail"]
    for key in ("mean", "std", "sharpe", "var", "cvar"):
        assert abs(got[key] - want[key]) <= 1e-6 * max(1.0, abs(want[key])), key


def test_fold_is_refused_for_several_portfolios(gpu_ctx):
    mu, cov = synthetic.synthetic_market(4)
    with pytest.raises(_ffi.McpError, match="MCP_FLAG_FOLD"):
        simulate_paths(mu, cov, synthetic.dirichlet_weights(4, 2), n_steps=3, n_paths=64, fold=True)
