#!/usr/bin/env python3
"""Run the REFERENCE's read_csv_file (loaded from /root/reference/app.py exactly as make_goldens.py does: the function's own
AST compiled into a namespace with numpy, pandas and a stub `st`) on the random price files of fuzz_csv.py and store what
it returns: None, or the row count and exact digests of the dates and prices.  Output: ref_fuzz_csv.json (numbers only).
On a machine without /root/reference the script exits 0 without touching the fixture."""
import hashlib
import io
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fuzz_csv           # noqa: E402
import make_goldens       # noqa: E402


def main():
    if not os.path.exists(make_goldens.APP):
        print("reference not present - fixture left untouched")
        return 0
    ns, st = make_goldens.load_functions()
    out = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for text, bom in fuzz_csv.cases():
            b = io.BytesIO((("﻿" if bom else "") + text).encode("utf-8"))
            b.name = "f.csv"
            st.records["error"].clear()
            df = ns["read_csv_file"](b)
            if df is None:
                out.append(None)
            else:
                days = df["Date"].values.astype("datetime64[D]").astype(np.int64)
                price = df["Price"].values.astype(np.float64)
                out.append({"rows": int(len(df)), "dates": hashlib.sha256(np.ascontiguousarray(days).tobytes()).hexdigest()[:16],
                            "prices": hashlib.sha256(np.ascontiguousarray(price).tobytes()).hexdigest()[:16]})
    json.dump({"n": len(out), "none": sum(o is None for o in out), "cases": out}, open(os.path.join(HERE, "ref_fuzz_csv.json"), "w"))
    print("wrote ref_fuzz_csv.json:", len(out), "cases,", sum(o is None for o in out), "rejected by the reference")

    # the scalar metrics (app.py:231-263) and the options overlay / payoff functions (app.py:164-229) on random inputs
    import fuzz_surface
    import pandas as pd
    fl = make_goldens.fl

    def dig(a):
        return hashlib.sha256(np.ascontiguousarray(np.asarray(a, np.float64)).tobytes()).hexdigest()[:16]
    met = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for c in fuzz_surface.metric_cases():
            r = pd.Series(c["returns"])
            met.append({"sharpe": fl(ns["sharpe_ratio"](r, c["rf"], c["ann"])), "sortino": fl(ns["sortino_ratio"](r, c["rf"], c["ann"])),
                        "vol": fl(ns["annual_volatility"](r, c["ann"])), "ret": fl(ns["annual_return"](r, c["ann"])),
                        "mdd": fl(ns["max_drawdown"](r)), "var": fl(ns["var"](r, c["alpha"])), "cvar": fl(ns["cvar"](r, c["alpha"]))})
        opt = []
        for c in fuzz_surface.option_cases():
            ser = ns["calc_options_series"](c["rows"], pd.Series(c["prices"]))
            pay = ns["calculate_payoff"](c["rows"], c["spot"], c["purchase"], c["grid"])
            be = ns["calculate_breakeven"](c["rows"], c["purchase"])
            qty_asset = sum(q for t, k, p, q in c["rows"] if t == fuzz_surface.T_BUY) or 1.0
            pl = ns["calculate_profit_loss_percent"](pay, c["purchase"], qty_asset)
            opt.append({"series": dig(ser.values), "series_len": int(len(ser)), "payoff": dig(pay), "breakeven": None if be is None else fl(be),
                        "pl": dig(pl)})
    ast_ = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for c in fuzz_surface.asset_cases():
            ser = pd.Series(c["prices"], index=pd.to_datetime(c["days"], unit="D"))
            d = ns["calc_asset_stats"](ser, c["freq"], c["rf"])
            ast_.append({k: (fl(v) if k != "returns" else dig(v.values)) for k, v in d.items()} | {"n_returns": int(len(d["returns"]))})
    json.dump({"metrics": met, "options": opt, "assets": ast_}, open(os.path.join(HERE, "ref_fuzz_surface.json"), "w"))
    print("wrote ref_fuzz_surface.json:", len(met), "metric cases,", len(opt), "option cases,", len(ast_), "asset-statistics cases")
    return 0


if __name__ == "__main__":
    sys.exit(main())
