#!/usr/bin/env python3
"""A thin Streamlit front end over monte_carlo_portfolio_amd: the four tabs of the reference's app.py, rebuilt on the
package's functions (optional shim).  Nothing is computed here: every number comes from the
surface the package exposes under the reference's names.

    streamlit run examples/streamlit_app.py

Tabs (reference lines they stand for): per-asset statistics (app.py:463-497), option strategy and P/L curve
(app.py:499-653), the five-method random-weight sweep on historical rows plus the optimum re-scored on simulated paths
(app.py:655-783 + the MI355X path engine), forecast (app.py:785-809: ARIMA/GARCH, out of scope -- the tab says so).
Streamlit is not part of the test image; tests/test_gpu_shim.py runs this file under a recording stand-in module.
"""
import os
import sys

import numpy as np
import streamlit as st

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import monte_carlo_portfolio_amd as mcp                      # noqa: E402
from monte_carlo_portfolio_amd import ingest, options, sweep  # noqa: E402

st.set_page_config(page_title="Monte Carlo portfolio (MI355X engine)", layout="wide")
state = st.session_state
state.setdefault("frames", [])                               # [(asset name, DataFrame[Date, Price])]
state.setdefault("investment_amount", 10000.0)

# ---- sidebar: files and settings (app.py:378-458) -------------------------------------------------------------------
with st.sidebar:
    compat = st.checkbox("reference CSV parsing (thousands separators are NOT parsed)", value=False)
    for up in st.file_uploader("price CSV files (Date + Price/Close column)", type=["csv"], accept_multiple_files=True) or []:
        if up.name not in [n for n, _ in state["frames"]]:
            df = mcp.read_csv_file(up, compat=compat, report=st.error)
            if df is not None:
                state["frames"].append((ingest.asset_name(up.name), df))
    period = st.selectbox("analysis period", ["M", "Q", "W"], index=0)
    annual_factor = ingest.ANNUAL_FACTOR[period]
    user_rf = st.number_input("risk-free rate (the reference's units: 3.0 means '3 %', used as entered)", value=3.0)
    state["investment_amount"] = st.number_input("capital", value=float(state["investment_amount"]))
    n_paths = int(st.number_input("simulated paths for the optimum", value=1_000_000, step=100_000))

if not state["frames"]:
    st.info("upload at least one CSV file")
    st.stop()

names, prices, resampled = ingest.align_prices(state["frames"], period)
with st.sidebar:
    lo = np.array([st.number_input(f"min weight {n}", 0.0, 1.0, 0.0) for n in names])
    hi = np.array([st.number_input(f"max weight {n}", 0.0, 1.0, 1.0) for n in names])

tab_stats, tab_options, tab_sweep, tab_forecast = st.tabs(["statistics", "options", "portfolio", "forecast"])

with tab_stats:                                               # app.py:484-495
    st.dataframe(mcp.stats_table(resampled, period, user_rf))

option_rows = {}
with tab_options:                                             # app.py:499-653
    asset = st.selectbox("asset", names)
    strategy = st.selectbox("strategy", list(options.STRATEGIES))
    spot = float(resampled[asset].iloc[-1])
    if strategy != options.STRATEGIES[0]:
        rows = options.strategy_rows(strategy, spot, premium_put=st.number_input("put premium", value=0.02),
                                     premium_call=st.number_input("call premium", value=0.02))
        option_rows[asset] = rows
        grid = options.payoff_grid(spot)
        st.line_chart({"price": grid, "P/L": mcp.calculate_payoff(rows, spot, spot, grid)})
        st.write({"breakeven": mcp.calculate_breakeven(rows, spot)})

with tab_sweep:                                               # app.py:655-783
    returns_df = mcp.returns_matrix(resampled, option_rows)
    results = mcp.run_all_methods(returns_df, min_weights=lo, max_weights=hi, user_rf=user_rf, annual_factor=annual_factor,
                                  investment_amount=state["investment_amount"])
    for method, r in results.items():
        i = r["opt_idx"]
        st.subheader(method)
        st.scatter_chart({"risk %": r["all_risks"] * 100, "return %": r["all_returns"] * 100})
        st.write({"optimum": i, "risk %": float(r["all_risks"][i] * 100), "return %": float(r["all_returns"][i] * 100),
                  "allocation": dict(zip(names, np.round(r["dollar_vals"], 2).tolist()))})
    w = results["Monte Carlo"]["weights"]
    mu_step, cov_step = returns_df.mean().values, returns_df.cov().values          # per period (app.py:679-680 before annualising)
    sim = mcp.simulate_paths(mu_step, cov_step, w, n_steps=annual_factor, n_paths=n_paths, seed=12345,
                             v0=state["investment_amount"], rf=user_rf / 100.0)
    st.subheader("max-Sharpe weights on simulated one-year paths (MI355X path engine)")
    st.write({k: sim[k] for k in ("n", "mean", "std", "sharpe", "var", "cvar", "min", "max")})

with tab_forecast:                                            # app.py:785-809
    st.info("ARIMA/GARCH forecasting is outside the scope of this package (SURVEY.md section 2, component 11).")
