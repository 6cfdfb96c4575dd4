"""NumPy restatement of the reference's risk reductions, applied to simulated terminal values.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function cites the reference lines it
follows; the goldens under tests/golden/ (generated from the reference itself by
tests/golden/make_goldens.py) pin `var`, `cvar` and the sweep loop.
"""
from __future__ import annotations

import numpy as np


def var(returns, alpha=0.95):
    """app.py:258-259 -- np.percentile(returns, (1-alpha)*100), default 'linear' method."""
    return np.percentile(returns, (1 - alpha) * 100)


def cvar(returns, alpha=0.95):
    """app.py:261-263 -- mean of the returns at or below VaR; VaR itself if that set is empty."""
    returns = np.asarray(returns)
    v = var(returns, alpha)
    mask = returns <= v
    return returns[mask].mean() if np.any(mask) else v


def terminal_to_x(terminal: np.ndarray, v0: float = 1.0, compounding: str = "simple") -> np.ndarray:
    """x = V_T/V0 - 1 with V0 rounded to binary32 (SPEC.md section 5); 'log': expm1(S_T)."""
    t = np.asarray(terminal, np.float32).astype(np.float64)
    if compounding == "log":
        return np.expm1(t)
    return t / np.float64(np.float32(v0)) - 1.0


def path_stats(terminal: np.ndarray, v0: float = 1.0, compounding: str = "simple", alpha: float = 0.95,
               rf: float = 0.0) -> dict:
    """Statistics of one portfolio's terminal values with the reference's definitions:
    std ddof=1 (app.py:234), Sharpe (mean-rf)/std or 0 (app.py:711), VaR/CVaR (app.py:258-263)."""
    x = terminal_to_x(terminal, v0, compounding)
    n = x.size
    mean = x.mean()
    std = x.std(ddof=1) if n > 1 else 0.0
    v = var(x, alpha)
    mask = x <= v
    return {
        "n": n, "mean": float(mean), "std": float(std),
        "sharpe": float((mean - rf) / std) if std > 0 else 0.0,
        "var": float(v), "cvar": float(x[mask].mean()) if mask.any() else float(v),
        "n_tail": int(mask.sum()), "sum_tail": float(x[mask].sum()),
        "min": float(x.min()), "max": float(x.max()),
    }
