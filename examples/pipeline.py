#!/usr/bin/env python3
"""CSV-in -> (weights, stats, P/L curve): the reference's tab 0 / 1 / 2 flow, headless, on the MI355X engine.

    python examples/pipeline.py [csv ...]        (defaults to three of the CSV files the reference ships)

Mirrors app.py's order of operations with the package's drop-in functions: read_csv_file (app.py:89) ->
align / resample (app.py:466-482) -> per-asset statistics table (app.py:484-495) -> option overlay and payoff
curve of one asset (app.py:499-653) -> returns matrix (app.py:658-667) -> the five-method random-weight sweep on
historical rows (app.py:682-783, GPU) -> the optimum re-scored on one million SIMULATED one-year paths (GPU).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import monte_carlo_portfolio_amd as mcp                                   # noqa: E402
from monte_carlo_portfolio_amd import ingest, options                     # noqa: E402


def main(paths, resample_rule="M", user_rf=3.0, investment=10000.0, seed=12345, n_paths=1_000_000):
    files = [open(p, "rb") for p in paths]
    names, prices, resampled = mcp.load_prices(files, resample_rule=resample_rule, report=lambda m: print("skip:", m))
    af = ingest.ANNUAL_FACTOR[resample_rule]
    print("assets:", names, "| rows after alignment:", len(prices), "| periods:", len(resampled))
    print(mcp.stats_table(resampled, resample_rule, user_rf).round(4).to_string())

    a0 = names[0]                                                          # a protective put on the first asset
    S = float(resampled[a0].iloc[-1])
    rows = options.strategy_rows("Protective Put", S, premium_put=0.02)
    grid = options.payoff_grid(S)
    pay = mcp.calculate_payoff(rows, S, S, grid)
    print(f"{a0}: protective put breakeven {mcp.calculate_breakeven(rows, S):.4f}, P/L at -50 % / +50 %: {pay[0]:.3f} / {pay[-1]:.3f}")

    returns_df = mcp.returns_matrix(resampled, {a0: rows})
    res = mcp.run_all_methods(returns_df, user_rf=user_rf, annual_factor=af, seed=seed, investment_amount=investment)
    for m, r in res.items():
        print(f"{m:12s} opt_idx {r['opt_idx']:5d}  risk {r['all_risks'][r['opt_idx']] * 100:8.3f} %  "
              f"return {r['all_returns'][r['opt_idx']] * 100:8.3f} %  dollars {np.round(r['dollar_vals'], 2)}")

    w = res["Monte Carlo"]["weights"]
    mu_step, cov_step = returns_df.mean().values, returns_df.cov().values  # per period, before annualising
    sim = mcp.simulate_paths(mu_step, cov_step, w, n_steps=af, n_paths=n_paths, seed=seed, v0=investment, rf=user_rf / 100)
    print(f"max-Sharpe weights on {n_paths:,} simulated {af}-period paths: mean {sim['mean']:+.4f}  std {sim['std']:.4f}  "
          f"VaR95 {sim['var']:+.4f}  CVaR95 {sim['cvar']:+.4f}  Sharpe {sim['sharpe']:.4f}")
    return res, sim


if __name__ == "__main__":
    data = os.path.join(ROOT, "tests", "golden", "data")
    default = [os.path.join(data, f) for f in ("Avalanche Historical Data.csv", "Cardano Historical Data.csv",
                                               "NEAR_USD Binance Historical Data.csv")]
    main(sys.argv[1:] or default)
