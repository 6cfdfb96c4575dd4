"""The scalar metrics (app.py:231-263) and the options overlay / payoff functions (app.py:164-229) on random inputs against
what the REFERENCE's own functions returned for the same inputs (tests/golden/make_fuzz_goldens.py -> ref_fuzz_surface.json;
the inputs come from the shared deterministic generator tests/golden/fuzz_surface.py).  Bit for bit.  CPU only."""
import hashlib
import json
import os
import sys
import warnings

import numpy as np
import pandas as pd
import pytest

from monte_carlo_portfolio_amd import metrics, options

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import fuzz_surface  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "ref_fuzz_surface.json")))


def dig(a):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a, np.float64)).tobytes()).hexdigest()[:16]


def same(got, want_hex):
    want = float.fromhex(want_hex)
    return (np.isnan(got) and np.isnan(want)) or float(got) == want


def test_metrics_equal_the_reference_on_random_series():
    cases = fuzz_surface.metric_cases()
    assert len(cases) == len(GOLD["metrics"]) == 120
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for c, g in zip(cases, GOLD["metrics"]):
            r = pd.Series(c["returns"])
            assert same(metrics.sharpe_ratio(r, c["rf"], c["ann"]), g["sharpe"])
            assert same(metrics.sortino_ratio(r, c["rf"], c["ann"]), g["sortino"])
            assert same(metrics.annual_volatility(r, c["ann"]), g["vol"])
            assert same(metrics.annual_return(r, c["ann"]), g["ret"])
            assert same(metrics.max_drawdown(r), g["mdd"])
            assert same(metrics.var(r, c["alpha"]), g["var"])
            assert same(metrics.cvar(r, c["alpha"]), g["cvar"])


def test_options_equal_the_reference_on_random_strategies():
    cases = fuzz_surface.option_cases()
    assert len(cases) == len(GOLD["options"]) == 80
    for c, g in zip(cases, GOLD["options"]):
        ser = options.calc_options_series(c["rows"], pd.Series(c["prices"]))
        assert len(ser) == g["series_len"] and dig(ser.values) == g["series"]
        pay = options.calculate_payoff(c["rows"], c["spot"], c["purchase"], c["grid"])
        assert dig(pay) == g["payoff"]
        be = options.calculate_breakeven(c["rows"], c["purchase"])
        assert (be is None) == (g["breakeven"] is None) and (be is None or same(be, g["breakeven"]))
        qty_asset = sum(q for t, k, p, q in c["rows"] if t == fuzz_surface.T_BUY) or 1.0
        assert dig(options.calculate_profit_loss_percent(pay, c["purchase"], qty_asset)) == g["pl"]


def test_calc_asset_stats_equals_the_reference_on_random_price_series():
    """app.py:286-335 on 48 random price series (gappy daily dates, a third of them newest-first) for freq in M / W / Q / D:
    all 16 scalar keys and the resampled return series, bit for bit."""
    cases = fuzz_surface.asset_cases()
    assert len(cases) == len(GOLD["assets"]) == 48
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for c, g in zip(cases, GOLD["assets"]):
            ser = pd.Series(c["prices"], index=pd.to_datetime(c["days"], unit="D"))
            d = metrics.calc_asset_stats(ser, c["freq"], c["rf"])
            assert len(d["returns"]) == g["n_returns"] and dig(d["returns"].values) == g["returns"], c["freq"]
            for k, want in g.items():
                if k not in ("returns", "n_returns"):
                    assert same(d[k], want), (k, c["freq"])
