"""simulate_paths: the Monte Carlo path simulator behind the Streamlit-shaped surface.

Replaces, for a *simulated* terminal-value distribution, what the reference computes per portfolio
on *historical* returns at app.py:708-713 (portfolio mean/vol, Sharpe, 5 % VaR, CVaR).  Inputs are
the same objects the reference builds at app.py:679-680 (mean vector, covariance matrix), here per
step.  All arithmetic on the path happens in libmcport.so's HIP kernels (SPEC.md); this module only
validates, factors Sigma (float64 Cholesky on the host, cast to fp32) and marshals arrays.
"""
from __future__ import annotations

import atexit
import ctypes
import threading

import numpy as np

from . import _ffi

_CTX_LOCK = threading.Lock()
_CTX: dict[tuple, "Context"] = {}


class Context:
    """Owns one mcp_ctx: device buffers + one stream per device.  `device` is a device index or a sequence of them
    (one context over several GPUs, exchanging through RCCL inside the library; a device listed
    more than once holds several logical shards).  Calls are serialised by the library."""

    def __init__(self, device=0, terminal_budget: int | None = None):
        devs = [int(device)] if np.isscalar(device) else [int(d) for d in device]
        self._h = ctypes.c_void_p()
        if len(devs) > 1 and len(set(devs)) == len(devs):
            _ffi.preload_rccl()
        arr = (ctypes.c_int * len(devs))(*devs)
        _ffi.check(_ffi.lib().mcp_ctx_create_multi(arr, len(devs), ctypes.byref(self._h)))
        self.device = devs[0]
        self.devices = tuple(devs)
        if terminal_budget is not None:
            self.set_terminal_budget(terminal_budget)

    def set_terminal_budget(self, nbytes: int):
        """Upper bound on resident terminal values per device; larger sweeps are produced and reduced in tiles of
        portfolios (include/mcport.h, mcp_ctx_set_terminal_budget)."""
        _ffi.check(_ffi.lib().mcp_ctx_set_terminal_budget(self._h, int(nbytes)))

    def exchange(self):
        """(mode, note): how the shards of this context exchange histograms and records -- 'unset' before the first
        path-sharded call, then 'none' (one shard), 'rccl', 'kernel' (logical shards of one device) or 'p2p' (distinct devices
        without RCCL: the kernel over peer access; `note` then says why RCCL was not used)."""
        lib = _ffi.lib()
        names = {_ffi.EXCHANGE_UNSET: "unset", _ffi.EXCHANGE_NONE: "none", _ffi.EXCHANGE_RCCL: "rccl", _ffi.EXCHANGE_KERNEL: "kernel",
                 _ffi.EXCHANGE_P2P: "p2p"}
        return names[lib.mcp_ctx_exchange_mode(self._h)], lib.mcp_ctx_exchange_note(self._h).decode("utf-8", "replace")

    def close(self):
        if self._h:
            _ffi.lib().mcp_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def simulate(self, prm: _ffi.McpParams, mu, chol, W, seed: int, path_begin: int, n_paths: int, store: bool):
        K = prm.n_portfolios
        stats = np.zeros(K, _ffi.STATS_DTYPE)
        term = np.empty((K, n_paths), np.float32) if store else None
        _ffi.check(_ffi.lib().mcp_simulate(
            self._h, ctypes.byref(prm), mu, chol, W, seed, path_begin, n_paths,
            term.ctypes.data_as(ctypes.c_void_p) if store else None,
            stats.ctypes.data_as(ctypes.c_void_p)))
        return stats, term


def _close_default_contexts() -> None:
    """Interpreter exit: destroy the cached contexts (streams, device buffers, RCCL communicators) while the HIP runtime
    and the library are still loaded."""
    with _CTX_LOCK:
        for ctx in list(_CTX.values()):
            try:
                ctx.close()
            except Exception:
                pass
        _CTX.clear()


atexit.register(_close_default_contexts)


def default_context(device=0) -> Context:
    """Process-wide context per device (or per tuple of devices)."""
    key = (int(device),) if np.isscalar(device) else tuple(int(d) for d in device)
    with _CTX_LOCK:
        if key not in _CTX:
            _CTX[key] = Context(key)
        return _CTX[key]


def cholesky_factor(cov: np.ndarray) -> np.ndarray:
    """Lower Cholesky factor of Sigma, float64 -> float32.  Raises ValueError if Sigma is not PD
    (e.g. fewer return rows than assets)."""
    cov = np.asarray(cov, np.float64)
    if cov.ndim != 2 or cov.shape[0] != cov.shape[1]:
        raise ValueError(f"cov must be square, got {cov.shape}")
    try:
        L = np.linalg.cholesky(cov)
    except np.linalg.LinAlgError as e:
        raise ValueError(f"covariance matrix is not positive definite: {e}") from None
    return np.ascontiguousarray(L, np.float32)


def prepare_inputs(mu, cov, weights, chol=None):
    mu = np.ascontiguousarray(mu, np.float32).ravel()
    n = mu.shape[0]
    if not 1 <= n <= _ffi.MCP_MAX_ASSETS:
        raise ValueError(f"n_assets={n} outside [1, {_ffi.MCP_MAX_ASSETS}]")
    L = cholesky_factor(cov) if chol is None else np.ascontiguousarray(np.tril(chol), np.float32)
    if L.shape != (n, n):
        raise ValueError(f"cov/chol shape {L.shape} does not match mu ({n})")
    W = np.ascontiguousarray(np.atleast_2d(np.asarray(weights, np.float32)))
    if W.shape[1] != n:
        raise ValueError(f"weights have {W.shape[1]} columns, expected {n}")
    return mu, L, W


def stats_to_dict(rec) -> dict:
    return {name: (int(rec[name]) if name in ("n", "n_tail") else float(rec[name])) for name in rec.dtype.names}


def simulate_paths(mu, cov, weights, n_steps=252, n_paths=10_000, seed=0, v0=1.0, compounding="simple",
                   rf=0.0, alpha=0.95, devices=None, store=False, path_begin=0, chol=None,
                   native_math=False, as_array=False, fold=False, shard="auto", context=None):
    """Simulate `n_paths` correlated return paths and reduce them to risk statistics.

    mu [N], cov [N,N] are per-step mean and covariance (the reference's `mean_returns`, `cov_matrix`
    of app.py:679-680 divided by `annual_factor`); weights [N] or [K,N].  Returns a dict for a single
    weight vector or a list of dicts for K portfolios (as_array=True: the [K] record array of mcp_stats
    instead); with store=True the dict carries 'terminal' (float32 [n_paths] or [K, n_paths]).

    devices: None / [d] -> one GPU; [d0, d1, ...] -> the path range sharded over those GPUs inside the library (RCCL
    all-reduce of the radix-select histograms, one all-gather of the moment records).
    shard="portfolios" (or "auto" with K >= 512 per device) shards the weight matrix instead: every GPU walks all
    paths for its slice of the portfolios, no collective at all (BASELINE configs[4]).
    """
    single = np.asarray(weights).ndim == 1
    mu32, L, W = prepare_inputs(mu, cov, weights, chol)
    devs = (0,) if not devices else tuple(int(d) for d in devices)
    if shard not in ("auto", "paths", "portfolios"):
        raise ValueError("shard must be 'auto', 'paths' or 'portfolios'")
    by_portfolio = len(devs) > 1 and (shard == "portfolios" or (shard == "auto" and W.shape[0] >= 512 * len(devs)))
    prm = _ffi.make_params(mu32.shape[0], n_steps, W.shape[0], compounding, v0, alpha, rf, native_math, fold, by_portfolio)
    ctx = context if context is not None else default_context(devs)
    stats, term = ctx.simulate(prm, mu32, L, W, int(seed), int(path_begin), int(n_paths), store)
    if as_array:                      # [K] structured array (fields of mcp_stats), for large sweeps
        return (stats, term) if store else stats
    out = [stats_to_dict(stats[k]) for k in range(W.shape[0])]
    if store:
        for k, d in enumerate(out):
            d["terminal"] = term[k]
    return out[0] if single else out


def simulate_sweep(mu, cov, weights=None, n_portfolios=2500, min_weights=None, max_weights=None, n_steps=252,
                   n_paths=100_000, seed=0, rf=0.0, alpha=0.95, method="Monte Carlo", np_seed=None, **kw):
    """The reference's sweep (app.py:682-722) scored on SIMULATED terminal values instead of historical rows:
    weights drawn exactly as app.py:699-707 (host, NumPy legacy RNG; pass `weights` to supply them), all
    portfolios on common random numbers in one launch (MFMA kernel for K >= 17), metric and optimum as at
    app.py:672-676 / 717 / 747.  -> dict(all_weights, stats [K] record array, all_metrics, opt_idx)."""
    from .sweep import _METRIC, draw_weights, select_optimum
    if weights is None:
        if np_seed is not None:
            np.random.seed(np_seed)
        weights = draw_weights(len(np.atleast_1d(mu)), n_portfolios, min_weights, max_weights)
    W = np.atleast_2d(np.asarray(weights, np.float64))
    stats = simulate_paths(mu, cov, W, n_steps=n_steps, n_paths=n_paths, seed=seed, rf=rf, alpha=alpha, as_array=True, **kw)
    metric = {"sharpe": stats["sharpe"], "var_95": -stats["var"], "cvar_95": -stats["cvar"]}[_METRIC[method]]
    return {"all_weights": W, "stats": stats, "all_metrics": metric, "opt_idx": select_optimum(method, metric),
            "all_risks": stats["std"], "all_returns": stats["mean"]}
