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
    return 0


if __name__ == "__main__":
    sys.exit(main())
