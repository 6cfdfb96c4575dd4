"""Overhead of the in-library sharding (mcp_ctx_create_multi) on ONE GPU: the same 8 x 10^6 paths as 1, 2, 4 and 8 logical
shards of device 0 (kernel exchange), and through a one-rank RCCL communicator (MCP_FORCE_RCCL=1 in a child process).
On distinct GPUs the shards run concurrently; here they share the chip, so equal times mean the choreography is free."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from monte_carlo_portfolio_amd import _ffi, simulate_paths, synthetic
    from monte_carlo_portfolio_amd.simulate import Context
    _ffi.preload_rccl()
    mu, cov = synthetic.synthetic_market(16); w = synthetic.equal_weights(16)
    S, P = int(sys.argv[2]), 8_000_000
    ctx = Context([0] * S)
    simulate_paths(mu, cov, w, n_steps=252, n_paths=P, seed=1, context=ctx, devices=[0] * S, shard="paths")
    t = time.perf_counter(); n = 4
    for i in range(n): r = simulate_paths(mu, cov, w, n_steps=252, n_paths=P, seed=2 + i, context=ctx, devices=[0] * S, shard="paths")
    dt = (time.perf_counter() - t) / n
    print(f"shards={S} rccl={os.environ.get('MCP_FORCE_RCCL', '0')}: {dt*1e3:8.2f} ms per call of {P:,} paths -> {P/dt:.3e} paths/s  (VaR {r['var']:.9f}, n_tail {r['n_tail']})")
    sys.exit(0)
for S, force in ((1, "0"), (1, "1"), (2, "0"), (4, "0"), (8, "0")):
    env = dict(os.environ, MCP_FORCE_RCCL=force)
    r = subprocess.run([sys.executable, __file__, "child", str(S)], env=env, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-500:])
