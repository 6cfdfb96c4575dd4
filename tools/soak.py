"""Soak test (GPU box): hunts for rare races.  (1) the K-portfolio statistics chain, many repetitions on re-used buffers,
every VaR compared with np.percentile; (2) the pipelined PathEngine (2 and 4 buffers), thousands of steps over a cycle of
seeds, every finished batch compared with the record the plain host call gives for its seed."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monte_carlo_portfolio_amd import _ffi, simulate_paths, synthetic
from monte_carlo_portfolio_amd.engine import PathEngine
from monte_carlo_portfolio_amd.simulate import prepare_inputs

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
lib = _ffi.lib()
t_end = time.time() + budget / 2
rng = np.random.default_rng(0)
reps = bad = 0
while time.time() < t_end:
    K, n = int(rng.choice([700, 1100, 2000, 37])), int(rng.choice([1000, 4096, 20000]))
    v = (1.0 + 0.05 * rng.standard_normal((K, n))).astype(np.float32)
    prm = _ffi.make_params(4, 1, K)
    term = torch.from_numpy(v).cuda()
    ws = [torch.zeros((lib.mcp_ws_bytes(w, K, n) + 7) // 8, dtype=torch.int64, device="cuda") for w in range(_ffi.WS_COUNT)]
    ws[_ffi.WS_PIVOT] = torch.from_numpy(v[:, 0].astype(np.float64) - 1.0).cuda()
    p = [ctypes.c_void_p(t.data_ptr()) for t in ws]
    lo, hi, g = _ffi.percentile_rank(n, 0.95)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    T, B = ctypes.c_void_p(term.data_ptr()), ctypes.byref(prm)
    P, R, S, H, Q, O, BL, C = (p[_ffi.WS_PARTIALS], p[_ffi.WS_RECORD], p[_ffi.WS_STATE], p[_ffi.WS_HIST], p[_ffi.WS_QUANT], p[_ffi.WS_STATS],
                               p[_ffi.WS_BELOW], p[_ffi.WS_PIVOT])
    want = np.percentile(v.astype(np.float64) - 1.0, 5.000000000000004, axis=1)
    for rep in range(20):
        lib.mcp_launch_pass0(B, T, n, n, C, P, H, st); lib.mcp_launch_scan(B, 0, n, lo, hi, P, BL, C, H, S, R, st)
        lib.mcp_launch_hist(B, 1, T, n, n, S, C, BL, H, st); lib.mcp_launch_scan(B, 1, n, lo, hi, P, BL, C, H, S, R, st)
        lib.mcp_launch_hist(B, 2, T, n, n, S, C, BL, H, st); lib.mcp_launch_final(B, n, g, lo, hi, BL, H, S, R, Q, O, st)
        torch.cuda.synchronize()
        rec = ws[_ffi.WS_STATS].cpu().numpy().view(np.uint8)[:K * _ffi.STATS_DTYPE.itemsize].view(_ffi.STATS_DTYPE)
        bad += int((rec["var"] != want).sum()); reps += 1
print(f"statistics chain: {reps} repetitions, {bad} wrong VaR values")

N, Tn, Pn = 16, 40, 60_000
mu, cov = synthetic.synthetic_market(N); w = synthetic.equal_weights(N)
mu32, L, W32 = prepare_inputs(mu, cov, w)
seeds = [11, 12, 13, 14, 15]
want = {s: simulate_paths(mu, cov, w, n_steps=Tn, n_paths=Pn, seed=s) for s in seeds}
steps = wrong = 0
t_end = time.time() + budget / 2
for nb in (2, 4):
    eng = PathEngine(mu32, L, W32, Tn, Pn, n_buffers=nb)
    hist = []
    t_stop = time.time() + budget / 4
    while time.time() < t_stop:
        for i in range(200):
            s = seeds[(steps + i) % len(seeds)]
            eng.step(s); hist.append(s)
        steps += 200
        eng.synchronize()
        for back in range(nb):                      # the nb most recent batches are still resident
            b = eng.bufs[(eng.last - back) % eng.n_buf][0]
            raw = b["ws"][_ffi.WS_STATS].cpu().numpy().view(np.uint8)[:_ffi.STATS_DTYPE.itemsize].view(_ffi.STATS_DTYPE)[0]
            s = hist[-1 - back]
            wrong += int(any(raw[k] != want[s][k] for k in raw.dtype.names))
print(f"pipelined engine: {steps} steps, {wrong} wrong records")
sys.exit(1 if (bad or wrong) else 0)
