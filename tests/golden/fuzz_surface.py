"""Deterministic random inputs for the host-surface functions (test DATA generator; no reference code).  Shared by
tests/golden/make_fuzz_goldens.py (runs the REFERENCE's functions on them) and tests/test_surface_fuzz.py."""
import numpy as np

T_BUY, T_SELL = "خرید دارایی", "فروش دارایی"
T_LCALL, T_SCALL, T_LPUT, T_SPUT, T_SFUT = "خرید کال", "فروش کال", "خرید پوت", "فروش پوت", "فروش فیوچرز"
ROW_TYPES = (T_BUY, T_SELL, T_LCALL, T_SCALL, T_LPUT, T_SPUT, T_SFUT)


def metric_cases(n_cases=120, seed=11):
    """-> list of dict(returns, rf, ann, alpha)"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_cases):
        n = int(rng.integers(2, 80))
        kind = i % 4
        r = rng.normal(0.01, 0.08, n)
        if kind == 1:
            r = np.round(r, 2)                                   # ties
        elif kind == 2:
            r = np.abs(r)                                        # no negative excess return: sortino's fallback
        elif kind == 3:
            r[0] = 0.0                                           # the leading 0.0 row of pct_change().fillna(0)
        out.append({"returns": r, "rf": float(rng.choice([0.0, 3.0, 0.03])), "ann": int(rng.choice([12, 4, 52, 252])),
                    "alpha": float(rng.choice([0.95, 0.99, 0.9, 0.5]))})
    return out


def option_cases(n_cases=80, seed=12):
    """-> list of dict(rows, prices, spot, purchase, grid)"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_cases):
        n = int(rng.integers(2, 40))
        prices = np.abs(100 * np.cumprod(1 + rng.normal(0.0, 0.06, n))) + 1.0
        spot = float(prices[-1])
        rows = []
        for _ in range(int(rng.integers(1, 5))):
            t = ROW_TYPES[int(rng.integers(0, len(ROW_TYPES)))]
            rows.append((t, float(spot * rng.uniform(0.7, 1.3)), float(spot * rng.uniform(0.0, 0.08)), float(rng.choice([0.5, 1.0, 2.0]))))
        out.append({"rows": rows, "prices": prices, "spot": spot, "purchase": float(spot * rng.uniform(0.8, 1.2)),
                    "grid": np.linspace(0.5 * spot, 1.5 * spot, 100)})
    return out


def asset_cases(n_cases=48, seed=13):
    """-> list of dict(days, prices, freq, rf): a price series on (possibly gappy, possibly descending) daily dates"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_cases):
        n = int(rng.integers(70, 500))
        days = np.sort(rng.choice(np.arange(19000, 19000 + 2 * n), size=n, replace=False))      # days since 1970-01-01, with gaps
        prices = 50 * np.cumprod(1 + rng.normal(0.0005, 0.03, n)) + 0.5
        if i % 3 == 0:
            days, prices = days[::-1].copy(), prices[::-1].copy()                                # newest first, like the CSV exports
        out.append({"days": days, "prices": prices, "freq": ["M", "W", "Q", "D"][i % 4], "rf": float(rng.choice([0.0, 3.0]))})
    return out
