#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in this container.

The reference is referred to by path only (/root/reference/app.py); this script contains none of its
source.  It is loaded two ways:
  * function level: every top-level function definition of app.py except `forecast_prices` is compiled
    from the file's own AST into a namespace holding numpy, pandas and a stub `st`;
  * whole script: app.py is executed headless with runpy under stub `streamlit`, `yfinance`,
    `statsmodels`, `arch` modules (none of them is installed here), the sidebar's file uploader fed
    with the reference's own CSVs, and np.random.seed(S) set beforehand.
Only numbers are written out (JSON / NPZ).  On a machine without /root/reference the script exits 0
without touching the fixtures (the GPU box only needs the committed fixtures).

Usage: python tests/golden/make_goldens.py
"""
import ast
import hashlib
import io
import json
import os
import runpy
import sys
import types
import warnings

import numpy as np
import pandas as pd

REF = "/root/reference"
APP = os.path.join(REF, "app.py")
HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(REF, "data")

T_BUY, T_SELL = "خرید دارایی", "فروش دارایی"
T_LCALL, T_SCALL, T_LPUT, T_SPUT, T_SFUT = "خرید کال", "فروش کال", "خرید پوت", "فروش پوت", "فروش فیوچرز"


# --------------------------------------------------------------------------- stub streamlit
class _Ctx:
    """No-op object: callable, context manager, attribute sink."""

    def __init__(self, st):
        self._st = st

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __getattr__(self, name):
        return getattr(self._st, name)


class _SessionState(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


class StubStreamlit(types.ModuleType):
    def __init__(self, files=(), overrides=None):
        super().__init__("streamlit")
        self.session_state = _SessionState()
        self.files = list(files)
        self.overrides = overrides or {}
        self.records = {"error": [], "write": [], "plotly_chart": [], "warning": [], "success": [], "markdown": []}
        self.sidebar = _Ctx(self)

    # widgets
    def file_uploader(self, *a, **k):
        return self.files

    def _store(self, key, val):
        if key is not None:
            self.session_state[key] = val
        return val

    def number_input(self, label, value=0.0, key=None, **k):
        return self._store(key, self.overrides.get(key, self.overrides.get(label, value)))

    def selectbox(self, label, options, key=None, **k):
        return self._store(key, self.overrides.get(key, self.overrides.get(label, options[0])))

    def text_input(self, *a, **k):
        return ""

    def date_input(self, label, value=None, **k):
        return value

    def button(self, *a, **k):
        return False

    def form_submit_button(self, *a, **k):
        return False

    def checkbox(self, label, value=False, **k):
        return value

    def columns(self, spec, **k):
        n = spec if isinstance(spec, int) else len(spec)
        return [_Ctx(self) for _ in range(n)]

    def tabs(self, names):
        return [_Ctx(self) for _ in names]

    def expander(self, *a, **k):
        return _Ctx(self)

    def form(self, *a, **k):
        return _Ctx(self)

    def spinner(self, *a, **k):
        return _Ctx(self)

    def container(self, *a, **k):
        return _Ctx(self)

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def sink(*a, **k):
            self.records.setdefault(name, []).append((a, k))
            return None

        return sink


def _raising_module(name, attrs):
    m = types.ModuleType(name)
    for a in attrs:
        def boom(*x, _a=a, **k):
            raise RuntimeError(f"{name}.{_a} is not available offline")
        setattr(m, a, boom)
    return m


def install_stubs(st):
    mods = {"streamlit": st, "yfinance": _raising_module("yfinance", ["download"])}
    sm = types.ModuleType("statsmodels"); tsa = types.ModuleType("statsmodels.tsa")
    arima = types.ModuleType("statsmodels.tsa.arima"); model = _raising_module("statsmodels.tsa.arima.model", ["ARIMA"])
    sm.tsa = tsa; tsa.arima = arima; arima.model = model
    mods.update({"statsmodels": sm, "statsmodels.tsa": tsa, "statsmodels.tsa.arima": arima,
                 "statsmodels.tsa.arima.model": model, "arch": _raising_module("arch", ["arch_model"])})
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    return saved


def restore(saved):
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def load_functions():
    """Top-level FunctionDefs of app.py (except forecast_prices) compiled from the file's own AST."""
    tree = ast.parse(open(APP, encoding="utf-8").read(), APP)
    st = StubStreamlit()
    ns = {"np": np, "pd": pd, "st": st}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name != "forecast_prices":
            exec(compile(ast.Module([node], []), APP, "exec"), ns)
    return ns, st


def upload(name):
    b = io.BytesIO(open(os.path.join(DATA, name), "rb").read())
    b.name = name
    return b


def arr_digest(a):
    a = np.ascontiguousarray(np.asarray(a, np.float64))
    return hashlib.sha256(a.tobytes()).hexdigest()


def fl(x):
    """float -> JSON-safe exact representation (hex string)."""
    return float(x).hex()


# --------------------------------------------------------------------------- function-level goldens
def golden_functions():
    ns, st = load_functions()
    out = {"numpy": np.__version__, "pandas": pd.__version__}

    # G1: read_csv_file on every CSV the reference ships
    g1 = {}
    for name in sorted(os.listdir(DATA)):
        if not name.endswith(".csv"):
            continue
        st.records["error"].clear()
        df = ns["read_csv_file"](upload(name))
        if df is None:
            g1[name] = {"result": None, "n_errors": len(st.records["error"])}
        else:
            g1[name] = {"result": "ok", "rows": int(len(df)), "columns": list(df.columns),
                        "first_date": str(df["Date"].iloc[0].date()), "last_date": str(df["Date"].iloc[-1].date()),
                        "min_date": str(df["Date"].min().date()), "max_date": str(df["Date"].max().date()),
                        "first_price": fl(df["Price"].iloc[0]), "last_price": fl(df["Price"].iloc[-1]),
                        "min_price": fl(df["Price"].min()), "max_price": fl(df["Price"].max()),
                        "sum_price": fl(df["Price"].sum()), "dtypes": [str(t) for t in df.dtypes]}
    out["G1_read_csv_file"] = g1

    # header sniffing / price-column choice on synthetic uploads
    def mk(text, name="x.csv"):
        b = io.BytesIO(text.encode("utf-8")); b.name = name
        return b
    cases = {
        "close_first": "Date,Close,Open\n2024-01-01,10,9\n2024-01-02,11,10\n",
        "open_before_close": "Date,Open,Close\n2024-01-01,9,10\n2024-01-02,10,11\n",
        "adj_close": "date,Volume,Adj Close\n2024-01-01,5,10.5\n2024-01-02,6,11.5\n",
        "no_known_price": "Date,Foo,Bar\n2024-01-01,1.5,2\n2024-01-02,2.5,3\n",
        "header_on_row_2": "junk,junk2\nmore,junk\nDate,Price\n2024-01-01,10\n2024-01-02,12\n",
        "no_date": "A,B\n1,2\n3,4\n",
        "bad_rows": "Date,Price\n2024-01-01,10\nnot a date,11\n2024-01-03,abc\n2024-01-04,13\n",
        "thousands": 'Date,Price\n2024-01-01,"1,234.5"\n2024-01-02,999.0\n',
        "all_bad": "Date,Price\nxx,yy\n",
        "empty": "",
    }
    g1b = {}
    for key, text in cases.items():
        st.records["error"].clear()
        df = ns["read_csv_file"](mk(text))
        g1b[key] = {"text": text, "result": None if df is None else
                    {"dates": [str(d.date()) for d in df["Date"]], "prices": [fl(p) for p in df["Price"]]},
                    "n_errors": len(st.records["error"])}
    out["G1b_read_csv_synthetic"] = g1b

    # G6: var / cvar known answers (incl. ties, tiny n, the g >= 0.5 branch of numpy's lerp)
    g6 = []
    case = 0
    for n in (1, 2, 13, 20, 21, 100, 1000, 4097):
        for kind in ("normal", "ties"):
            case += 1
            x = np.random.RandomState(1000 + case).standard_normal(n) * 0.05      # regenerated by the test
            if kind == "ties":
                x = np.round(x, 2)
            for alpha in (0.95, 0.99, 0.9):
                rec = {"n": n, "kind": kind, "alpha": alpha, "rs_seed": 1000 + case,
                       "var": fl(ns["var"](x, alpha)), "cvar": fl(ns["cvar"](pd.Series(x), alpha)),
                       "cvar_ndarray": fl(ns["cvar"](x, alpha))}
                if n <= 21:
                    rec["x"] = [fl(v) for v in x]
                g6.append(rec)
    out["G6_var_cvar"] = g6

    # scalar risk metrics on a seeded monthly-like series
    r = pd.Series(np.random.RandomState(20250614).standard_normal(36) * 0.08 + 0.01)
    out["G_metrics"] = {
        "returns": [fl(v) for v in r],
        "sharpe_ratio": fl(ns["sharpe_ratio"](r, 3.0, 12)), "sharpe_ratio_rf0": fl(ns["sharpe_ratio"](r)),
        "sortino_ratio": fl(ns["sortino_ratio"](r, 3.0, 12)), "sortino_all_positive": fl(ns["sortino_ratio"](r.abs() + 1.0, 0, 12)),
        "annual_volatility": fl(ns["annual_volatility"](r, 52)), "annual_return": fl(ns["annual_return"](r, 12)),
        "max_drawdown": fl(ns["max_drawdown"](r)), "sharpe_zero_std": fl(ns["sharpe_ratio"](pd.Series([0.01] * 5))),
    }

    # G5: efficient_frontier, seeded, with and without constraints, and an infeasible set (Q9)
    g5 = {}
    mean_returns = np.array([0.02, 0.15, -0.03, 0.08])
    A = np.array([[0.9, 0.2, 0.1, 0.0], [0.2, 1.4, 0.3, 0.1], [0.1, 0.3, 0.7, 0.2], [0.0, 0.1, 0.2, 0.5]])
    cov = A @ A.T
    for tag, kw in {"free": {}, "bounded": {"min_weights": np.array([0.05] * 4), "max_weights": np.array([0.6] * 4)},
                    "infeasible": {"min_weights": np.array([0.3] * 4), "max_weights": np.array([0.31] * 4)}}.items():
        np.random.seed(777)
        res, wts = ns["efficient_frontier"](mean_returns, cov, points=40, **kw)
        g5[tag] = {"results": [[fl(v) for v in row] for row in res], "weights": [[fl(v) for v in row] for row in wts]}
    out["G5_efficient_frontier"] = {"mean_returns": [fl(v) for v in mean_returns], "cov": [[fl(v) for v in row] for row in cov],
                                    "seed": 777, "points": 40, "cases": g5}

    # G7: payoff / breakeven / P&L % for the seven strategies of the UI table, non-zero premiums
    S = 100.0
    strategies = {
        "Married Put": [(T_BUY, 0, 0, 2.0), (T_LPUT, 90.0, 0.03, 2.0)],
        "Covered Call": [(T_SCALL, 110.0, 0.02, 1.5)],
        "Collar": [(T_LPUT, 90.0, 0.03, 1.0), (T_SCALL, 110.0, 0.02, 1.0)],
        "Bear Put Spread": [(T_LPUT, 100.0, 0.05, 1.0), (T_SPUT, 90.0, 0.02, 1.0)],
        "Synthetic Put": [(T_SFUT, 0, 0, 1.0), (T_LCALL, 100.0, 0.04, 1.0)],
        "Long Straddle": [(T_LCALL, 100.0, 0.04, 1.0), (T_LPUT, 100.0, 0.035, 1.0)],
        "Short asset + zero qty put": [(T_SELL, 0, 0, 1.0), (T_LPUT, 95.0, 0.01, 0.0)],
        "asset only": [(T_BUY, 0, 0, 3.0)],
    }
    grid = np.linspace(S * 0.5, S * 1.5, 100)
    g7 = {}
    for name, rows in strategies.items():
        pay = ns["calculate_payoff"](rows, S, 97.0, grid)
        g7[name] = {"rows": [[r[0], r[1], r[2], r[3]] for r in rows], "payoff": [fl(v) for v in pay],
                    "breakeven": fl(ns["calculate_breakeven"](rows, 97.0)),
                    "pl_percent": [fl(v) for v in ns["calculate_profit_loss_percent"](pay, 97.0, 2.0)],
                    "pl_percent_zero_investment": [fl(v) for v in ns["calculate_profit_loss_percent"](pay[:3], 97.0, 0.0)]}
    out["G7_payoff"] = {"current_price": S, "purchase_price": 97.0, "cases": g7}

    # G8: calc_options_series on a 13-point price path, every row type
    prices = pd.Series([100, 104, 99, 97, 103, 110, 108, 95, 90, 96, 101, 107, 105.0],
                       index=pd.date_range("2024-01-31", periods=13, freq="ME"))
    g8 = {"prices": [fl(v) for v in prices]}
    for t in (T_BUY, T_SELL, T_LCALL, T_SCALL, T_LPUT, T_SPUT, T_SFUT, "unknown"):
        g8[t] = [fl(v) for v in ns["calc_options_series"]([(t, 100.0, 1.5, 2.0)], prices)]
    g8["collar"] = [fl(v) for v in ns["calc_options_series"]([(T_BUY, 0, 0, 1.0), (T_LPUT, 95.0, 1.0, 1.0), (T_SCALL, 108.0, 0.8, 1.0)], prices)]
    g8["zero_prev"] = [fl(v) for v in ns["calc_options_series"]([(T_BUY, 0, 0, 1.0)], pd.Series([0.0, 1.0, 2.0]))]
    out["G8_options_series"] = g8
    return out, ns


# --------------------------------------------------------------------------- whole-script goldens
class DirichletRecorder:
    def __init__(self):
        self.orig = np.random.dirichlet
        self.draws = []

    def __call__(self, alpha, size=None):
        r = self.orig(alpha, size=size)
        self.draws.append(np.array(r, copy=True))
        return r


def run_script(files, seed, overrides=None):
    st = StubStreamlit([upload(f) for f in files], overrides)
    saved = install_stubs(st)
    rec = DirichletRecorder()
    np.random.dirichlet = rec
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            np.random.seed(seed)
            g = runpy.run_path(APP, run_name="__golden__")
    finally:
        np.random.dirichlet = rec.orig
        restore(saved)
    return g, st, rec


def golden_script(npz):
    files = ["Avalanche Historical Data.csv", "Cardano Historical Data.csv", "NEAR_USD Binance Historical Data.csv"]
    out = {"files": files}
    methods = ["Monte Carlo", "VaR", "CVaR", "MPT", "Equal Weight"]
    for tag, overrides, seeds in (("monthly", {}, (12345, 0, 1)),
                                  ("weekly", {"بازه تحلیل بازده": "هفتگی"}, (12345,)),
                                  ("monthly_collar", {"strategy_Cardano Historical Data": "Collar",
                                                      "premium_put_Cardano Historical Data": 0.01,
                                                      "premium_call_Cardano Historical Data": 0.02}, (12345,))):
        for seed in seeds:
            g, st, rec = run_script(files, seed, overrides)
            key = f"{tag}_seed{seed}"
            rd = g["returns_df"]
            entry = {"asset_names": list(g["asset_names"]), "annual_factor": g["annual_factor"], "user_rf": g["user_rf"],
                     "returns_shape": list(rd.shape), "n_dirichlet_draws": len(rec.draws),
                     "mean_returns": [fl(v) for v in g["mean_returns"].values],
                     "cov_matrix": [[fl(v) for v in row] for row in g["cov_matrix"].values]}
            if seed == seeds[0]:
                npz[f"{key}__returns_df"] = rd.values
                npz[f"{key}__resampled_prices"] = g["resampled_prices"].values
                entry["resampled_index"] = [str(d.date()) for d in g["resampled_prices"].index]
                entry["prices_rows"] = int(len(g["prices_df"]))
                stats_df = st.records["write"][0][0][0]
                entry["stats_columns"] = list(stats_df.columns)
                entry["stats_df"] = [[fl(v) for v in row] for row in stats_df.values.astype(float)]
            figs = [a[0] for a, k in st.records["plotly_chart"]]
            # per method: frontier figure then pie figure, in script order, after any payoff figures of tab 1
            n_payoff = len(figs) - 2 * len(methods) - len(g["asset_names"])    # tab 3 adds one forecast figure per asset
            mfigs = figs[n_payoff:n_payoff + 2 * len(methods)]
            per = {}
            for i, m in enumerate(methods):
                fr, pie = mfigs[2 * i], mfigs[2 * i + 1]
                risks = np.asarray(fr.data[0].x, float) / 100.0
                rets = np.asarray(fr.data[0].y, float) / 100.0
                metrics = np.asarray(fr.data[0].marker.color, float)
                opt = fr.data[-1]
                dollars = np.asarray(pie.data[0].values, float)
                info = {"n": int(len(risks)), "opt_point_pct": [fl(opt.x[0]), fl(opt.y[0])],
                        "dollar_vals": [fl(v) for v in dollars],
                        "sha_risks_pct": arr_digest(np.asarray(fr.data[0].x, float)),
                        "sha_returns_pct": arr_digest(np.asarray(fr.data[0].y, float)),
                        "sha_metrics": arr_digest(metrics),
                        "metric_min": fl(metrics.min()), "metric_max": fl(metrics.max())}
                # opt_idx: recomputed from the recorded metric array with the script's own rule is what the
                # marker shows; recover it from the marker position instead (exact match of x*100)
                xs = np.asarray(fr.data[0].x, float); ys = np.asarray(fr.data[0].y, float)
                hit = np.flatnonzero((xs == opt.x[0]) & (ys == opt.y[0]))
                info["opt_idx"] = int(hit[0])
                if m == "MPT":
                    cal = fr.data[1]
                    info["cal_y_first_last"] = [fl(cal.y[0]), fl(cal.y[-1])]
                    info["cal_x_last"] = fl(cal.x[-1])
                if seed == seeds[0] and tag != "weekly":
                    npz[f"{key}__{m}__risks_pct"] = np.asarray(fr.data[0].x, float)
                    npz[f"{key}__{m}__returns_pct"] = np.asarray(fr.data[0].y, float)
                    npz[f"{key}__{m}__metrics"] = metrics
                per[m] = info
            entry["methods"] = per
            if tag == "monthly_collar":
                pay = figs[0]
                entry["payoff_fig_traces"] = len(pay.data)
            out[key] = entry
    # config 0 inputs: the reference's loader rejects these three files (thousands separators, Q1)
    g, st, rec = run_script(["Bitcoin Historical Data.csv", "Ethereum Historical Data.csv", "XAU_USD Historical Data.csv"], 1)
    out["config0_reference_loader"] = {"n_errors": len(st.records["error"]), "uploaded": len(g["st"].session_state["uploaded_dfs"]),
                                       "has_returns_df": "returns_df" in g}
    return out


def main():
    if not os.path.exists(APP):
        print("reference not present at", APP, "- fixtures left untouched")
        return 0
    f, ns = golden_functions()
    json.dump(f, open(os.path.join(HERE, "ref_functions.json"), "w"), ensure_ascii=False, indent=1)
    npz = {}
    s = golden_script(npz)
    json.dump(s, open(os.path.join(HERE, "ref_script.json"), "w"), ensure_ascii=False, indent=1)
    np.savez_compressed(os.path.join(HERE, "ref_script_arrays.npz"), **npz)
    print("wrote ref_functions.json, ref_script.json, ref_script_arrays.npz")
    return 0


if __name__ == "__main__":
    sys.exit(main())
