"""The reference's random-weight portfolio sweep (what app.py calls "Monte Carlo"), on the GPU.

Reference lines restated here (script code of tab 2, app.py:655-764; function twin app.py:265-284):
  * weights: up to 100 tries of `np.random.dirichlet(np.ones(N), size=1)[0]` per portfolio from NumPy's
    global legacy generator, accepted iff inside [min_weights, max_weights]; a portfolio with no
    accepted draw is skipped (app.py:699-707).  Drawn HERE ON THE HOST WITH THE SAME NUMPY CALLS, so the
    weights are bit-identical to the reference's for the same seed and the optimum index is decidable.
  * scoring: app.py:708-713 for all accepted weight vectors at once -> libmcport.so
    (`mcp_sweep_historical`, HIP kernel `sweep_hist_kernel`, binary64).
  * metric / optimum: sharpe | -var | -cvar, argmax / argmin / 0 (app.py:672-676, 717, 747).
  * allocation: weights * investment_amount (app.py:763-764).
Quirks kept on purpose: Q2 `user_rf` is subtracted as given (default 3.0,
"percent"); Q7 the four random methods share one RNG stream in the order Monte Carlo, VaR, CVaR, MPT;
Q8 skipped portfolios shorten the arrays; Q9 `efficient_frontier` keeps the last rejected draw.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _ffi
from .simulate import default_context

METHODS = ("Monte Carlo", "VaR", "CVaR", "MPT", "Equal Weight")          # app.py:671-677, dict order
_METRIC = {"Monte Carlo": "sharpe", "VaR": "var_95", "CVaR": "cvar_95", "MPT": "sharpe", "Equal Weight": "sharpe"}


def draw_weights(n_assets, n_portfolios=2500, min_weights=None, max_weights=None):
    """app.py:699-707.  Uses (and advances) NumPy's global legacy RNG exactly like the reference: per portfolio up to 100
    tries of `np.random.dirichlet(np.ones(N), size=1)[0]`, the first one inside [min, max] is kept, a portfolio without one
    is skipped.  The legacy generator fills a `size=B` request row by row with the same arithmetic as B requests of size 1,
    so the candidates are drawn in blocks and the accept / retry / skip walk replayed over them; the generator is then put
    where the reference's loop would have left it (the four random methods share one stream, quirk Q7).  Same weights,
    same stream position, two orders of magnitude less host time (`_draw_weights_loop` is the literal form; the tests compare)."""
    lo = np.zeros(n_assets) if min_weights is None else np.asarray(min_weights, float)
    hi = np.ones(n_assets) if max_weights is None else np.asarray(max_weights, float)
    ones = np.ones(n_assets)
    out = []
    remaining, tries_used = int(n_portfolios), 0          # tries already spent on the portfolio being drawn
    while remaining > 0:
        state = np.random.get_state()
        B = max(256, 2 * remaining)
        C = np.random.dirichlet(ones, size=B)              # the next B candidates of the stream
        ok = np.all(C >= lo, axis=1) & np.all(C <= hi, axis=1)
        hits = np.flatnonzero(ok)
        pos = 0                                            # candidates of this block consumed so far
        if hits.size and hits[0] < 100 - tries_used and (np.diff(hits) <= 100).all():
            take = min(remaining, hits.size)               # no portfolio runs out of tries before the last hit taken:
            out.append(C[hits[:take]])                     # portfolios simply take consecutive accepted candidates
            remaining -= take
            pos = int(hits[take - 1]) + 1
            tries_used = 0
        pos, tries_used, remaining = _walk_block(C, ok, out, remaining, tries_used, pos)
        if pos < B:                                        # stopped inside the block: rewind and consume exactly `pos` rows
            np.random.set_state(state)
            if pos:
                np.random.dirichlet(ones, size=pos)
    return np.concatenate(out).reshape(-1, n_assets) if out else np.empty((0, n_assets))


def _walk_block(C, ok, out, remaining, tries_used, pos=0):
    """The literal accept / retry / skip walk of app.py:699-707 over one block of candidates, from row `pos`.  Returns
    (rows consumed, tries spent on the unfinished portfolio, portfolios still to draw)."""
    B = len(ok)
    while remaining > 0 and pos < B:
        window = 100 - tries_used
        seg = ok[pos:pos + window]
        hit = np.flatnonzero(seg)
        if hit.size:                                       # accepted on try tries_used + hit[0] + 1
            out.append(C[pos + hit[0]][None, :])
            pos += int(hit[0]) + 1
            remaining -= 1
            tries_used = 0
        elif len(seg) == window:                           # 100 tries failed: the portfolio is skipped (Q8)
            pos += window
            remaining -= 1
            tries_used = 0
        else:                                              # block ends inside this portfolio's tries
            tries_used += len(seg)
            pos = B
    return pos, tries_used, remaining


def _draw_weights_loop(n_assets, n_portfolios=2500, min_weights=None, max_weights=None):
    """The reference's loop as written (app.py:699-707); kept as the yardstick for draw_weights."""
    lo = np.zeros(n_assets) if min_weights is None else np.asarray(min_weights, float)
    hi = np.ones(n_assets) if max_weights is None else np.asarray(max_weights, float)
    out = []
    ones = np.ones(n_assets)
    for _ in range(n_portfolios):
        for _ in range(100):
            ws = np.random.dirichlet(ones, size=1)[0]
            if np.all(ws >= lo) and np.all(ws <= hi):
                out.append(ws)
                break
    return np.array(out).reshape(len(out), n_assets)


def sweep_inputs(returns_df, annual_factor):
    """mean_returns, cov_matrix of app.py:679-680 (pandas cov: ddof = 1) and the [R, N] float64 matrix."""
    if hasattr(returns_df, "cov"):                       # pandas DataFrame: literally the reference's calls
        mean = (returns_df.mean() * annual_factor).to_numpy(dtype=np.float64)
        cov = (returns_df.cov() * annual_factor).to_numpy(dtype=np.float64)
        R = returns_df.to_numpy(dtype=np.float64)
    else:                                                # plain [R, N] array (the pandas-free ingest): the same numbers,
        from .ingest_np import pandas_cov, pandas_mean   # bit for bit, as the DataFrame route (tests/test_ingest_np.py)
        R = np.asarray(returns_df, np.float64)
        mean = pandas_mean(R) * annual_factor
        cov = pandas_cov(R) * annual_factor
    return np.ascontiguousarray(R), np.ascontiguousarray(mean), np.ascontiguousarray(cov)


def score_portfolios(R, mean, cov, W, rf, alpha=0.95, device=0):
    """app.py:708-713 for all rows of W on the GPU -> dict of float64 [P] arrays."""
    W = np.ascontiguousarray(W, np.float64)
    P, N = W.shape
    outs = [np.empty(P, np.float64) for _ in range(5)]
    if P:
        ctx = default_context(device)
        _ffi.check(_ffi.lib().mcp_sweep_historical(ctx._h, N, R.shape[0], P, R, mean, cov, W, float(rf), float(alpha), *outs))
    return dict(zip(("port_return", "port_std", "sharpe", "var_95", "cvar_95"), outs))


def select_optimum(method, metrics):
    """`opt_crit` of app.py:672-676 applied as at app.py:747 (the VaR/CVaR arrays hold -var, -cvar)."""
    if method == "Equal Weight":
        return 0
    if len(metrics) == 0:
        raise ValueError("no portfolio satisfied the weight constraints (the reference raises here too, Q8)")
    return int(np.argmax(metrics)) if _METRIC[method] == "sharpe" else int(np.argmin(metrics))


def run_sweep(returns_df, method="Monte Carlo", n_portfolios=2500, min_weights=None, max_weights=None,
              user_rf=3.0, annual_factor=12, seed=None, alpha=0.95, device=0):
    """One method of the loop at app.py:682-722 -> (all_risks, all_returns, all_weights, all_metrics, opt_idx).

    `seed` (if given) seeds NumPy's global legacy RNG first; pass None to continue the current stream,
    which is how the reference's four random methods follow each other (Q7)."""
    if method not in METHODS:
        raise ValueError(f"unknown method {method!r}")
    R, mean, cov = sweep_inputs(returns_df, annual_factor)
    N = R.shape[1]
    if seed is not None:
        np.random.seed(seed)
    if method == "Equal Weight":
        w = np.ones(N) / N                                                      # app.py:686
        lo = np.zeros(N) if min_weights is None else np.asarray(min_weights, float)
        hi = np.ones(N) if max_weights is None else np.asarray(max_weights, float)
        W = w[None, :] if (np.all(w >= lo) and np.all(w <= hi)) else np.empty((0, N))
    else:
        W = draw_weights(N, n_portfolios, min_weights, max_weights)
    s = score_portfolios(R, mean, cov, W, user_rf, alpha, device)
    metric = {"sharpe": s["sharpe"], "var_95": -s["var_95"], "cvar_95": -s["cvar_95"]}[_METRIC[method]]   # app.py:717
    if method == "Equal Weight" and len(metric) == 0:
        raise IndexError("equal weights violate the constraints (the reference fails at app.py:749, Q8)")
    return s["port_std"], s["port_return"], W, metric, select_optimum(method, metric)


def run_all_methods(returns_df, n_portfolios=2500, min_weights=None, max_weights=None, user_rf=3.0,
                    annual_factor=12, seed=None, investment_amount=10000.0, device=0):
    """The whole loop of app.py:682-783 (without the plots): dict method -> results, RNG stream shared (Q7)."""
    if seed is not None:
        np.random.seed(seed)
    out = {}
    for m in METHODS:
        risks, rets, W, metrics, opt = run_sweep(returns_df, m, n_portfolios, min_weights, max_weights, user_rf,
                                                 annual_factor, None, 0.95, device)
        out[m] = {"all_risks": risks, "all_returns": rets, "all_weights": W, "all_metrics": metrics, "opt_idx": opt,
                  "weights": W[opt], "dollar_vals": allocation(W[opt], investment_amount)}
    return out


def allocation(weights, investment_amount=10000.0):
    """app.py:763-764."""
    return np.asarray(weights, float) * investment_amount


def capital_allocation_line(all_risks, all_metrics, user_rf, opt_idx, n=100):
    """The CAL overlay of the 'MPT' method, app.py:737-746 (x, y in percent)."""
    sharpe_star = all_metrics[opt_idx]
    cal_x = np.linspace(0, all_risks.max() * 1.3 * 100, n)
    return cal_x, user_rf * 100 + sharpe_star * cal_x


def efficient_frontier(mean_returns, cov_matrix, points=200, min_weights=None, max_weights=None, device=0):
    """app.py:265-284 (never called by the reference's UI): (results[3, points], weights[points, N]);
    rows of results: std, return, return/std (no risk-free rate here, unlike the live loop).  Keeps Q9:
    if all 100 draws of a point violate the constraints, the last rejected draw is used."""
    mean = np.ascontiguousarray(np.asarray(mean_returns, np.float64))
    cov = np.ascontiguousarray(np.asarray(cov_matrix, np.float64))
    N = len(mean)
    ones = np.ones(N)
    W = np.empty((points, N))
    for i in range(points):
        for _ in range(100):
            w = np.random.dirichlet(ones, size=1)[0]
            if min_weights is not None and not np.all(w >= min_weights):
                continue
            if max_weights is not None and not np.all(w <= max_weights):
                continue
            break
        W[i] = w
    results = np.zeros((3, points))
    if points:
        s = score_portfolios(np.zeros((1, N)), mean, cov, W, 0.0, 0.95, device)
        results[0], results[1], results[2] = s["port_std"], s["port_return"], s["sharpe"]
    return results, W
