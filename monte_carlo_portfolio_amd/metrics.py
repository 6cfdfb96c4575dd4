"""Per-asset risk metrics with the reference's conventions (app.py:231-263, 286-335).

Host logic in NumPy/pandas (a dozen numbers per asset).  The same definitions drive the GPU
reductions: ddof = 1 standard deviation, prod(1+r) compounding, percentile VaR, tail-mean CVaR.
Quirks kept: Q2 `risk_free` is divided by `ann_factor` whatever its unit; Q4 a quarterly rule uses
ann_factor 12 inside calc_asset_stats; Sortino falls back to 1e-4 when no return is negative.
"""
from __future__ import annotations

import numpy as np

from .ingest import resample_last


def sharpe_ratio(returns, risk_free=0, ann_factor=12):
    """app.py:231-236."""
    ex = returns - risk_free / ann_factor
    sd = np.std(ex, ddof=1)
    return 0 if sd == 0 else (np.mean(ex) / sd) * np.sqrt(ann_factor)


def sortino_ratio(returns, risk_free=0, ann_factor=12):
    """app.py:238-243: downside deviation = ddof-1 std of the NEGATIVE excess returns, 1e-4 if there are none."""
    ex = returns - risk_free / ann_factor
    neg = ex[ex < 0]
    down = np.std(neg, ddof=1) if len(neg) > 0 else 0.0001
    return (np.mean(ex) / down) * np.sqrt(ann_factor)


def annual_volatility(returns, ann_factor=12):
    """app.py:245-246."""
    return np.std(returns, ddof=1) * np.sqrt(ann_factor)


def annual_return(returns, ann_factor=12):
    """app.py:248-250: prod(1+r)^(ann/len) - 1."""
    return np.prod(1 + returns) ** (ann_factor / len(returns)) - 1


def max_drawdown(returns):
    """app.py:252-256: min over time of (cumprod - running max)/running max."""
    wealth = np.cumprod(1 + returns)
    peak = np.maximum.accumulate(wealth)
    return np.min((wealth - peak) / peak)


def var(returns, alpha=0.95):
    """app.py:258-259: np.percentile(returns, (1-alpha)*100); note (1-0.95)*100 == 5.000000000000004 (Q5)."""
    return np.percentile(returns, (1 - alpha) * 100)


def cvar(returns, alpha=0.95):
    """app.py:261-263."""
    v = var(returns, alpha)
    return returns[returns <= v].mean() if np.any(returns <= v) else v


STAT_KEYS = ("sharpe", "sortino", "volatility_ann", "total_return_ann", "implied_vol", "mean_ann", "mean_month",
             "std_ann", "std_month", "min_ann", "max_ann", "min_month", "max_month", "max_drawdown", "var_95", "cvar_95")
TABLE_COLUMNS = ("sharpe", "sortino", "volatility_ann", "total_return_ann", "implied_vol", "mean_ann", "mean_month",
                 "std_ann", "std_month", "min_ann", "min_month", "max_ann", "max_month", "var_95", "cvar_95")   # app.py:490-494


def calc_asset_stats(prices, freq="M", risk_free=0):
    """app.py:286-335: dict with the 16 scalar keys above plus 'returns'."""
    if freq == "D":
        r, ann = prices.pct_change().dropna(), 252
    else:
        r = resample_last(prices, freq).pct_change().dropna()
        ann = {"M": 12, "W": 52}.get(freq, 12)               # anything else, 'Q' included, annualises by 12 (Q4)
    sd = np.std(r, ddof=1)
    out = {
        "sharpe": sharpe_ratio(r, risk_free, ann), "sortino": sortino_ratio(r, risk_free, ann),
        "volatility_ann": annual_volatility(r, ann), "total_return_ann": annual_return(r, ann),
        "implied_vol": sd * np.sqrt(ann), "mean_ann": np.mean(r) * ann, "mean_month": np.mean(r),
        "std_ann": sd * np.sqrt(ann), "std_month": sd, "min_ann": np.min(r) * ann, "max_ann": np.max(r) * ann,
        "min_month": np.min(r), "max_month": np.max(r), "max_drawdown": max_drawdown(r),
        "var_95": var(r, 0.95), "cvar_95": cvar(r, 0.95), "returns": r,
    }
    return out


def stats_table(resampled_prices, freq="M", risk_free=0):
    """The table of tab 0 (app.py:484-495): one row per asset, columns in the reference's order."""
    import pandas as pd
    rows = {name: calc_asset_stats(resampled_prices[name], freq, risk_free) for name in resampled_prices.columns}
    return pd.DataFrame(rows).T[list(TABLE_COLUMNS)]
