"""CSV-in: price-file ingest and the price-alignment step in front of the sweep / simulator.

Restates app.py:89-134 (`read_csv_file`) and the tab-0 pipeline app.py:466-482, app.py:658-667.
Host logic (pandas); nothing here is on the GPU path.

`compat=True` reproduces the reference exactly, including quirk Q1: `pd.read_csv` is called without
`thousands=`, so prices such as "86,493.0" are coerced to NaN and dropped; a file with no surviving row
is rejected (this is what happens to the BTC / ETH / XAU files the reference ships).  The default
`compat=False` strips thousands separators first, which is what makes BASELINE configs[0] loadable.
"""
from __future__ import annotations

import warnings
from collections import Counter

import numpy as np
import pandas as pd

PRICE_NAMES = ("price", "close", "adj close", "open")


def _default_report(msg: str) -> None:
    warnings.warn(msg, stacklevel=3)


def _norm(c) -> str:
    return str(c).strip().lower()


def read_csv_file(file, compat: bool = False, report=None):
    """file-like (seekable, with .name) -> DataFrame[Date: datetime64, Price: float64] or None.

    Failures are reported through `report(message)` (the reference calls st.error, app.py:133) and
    yield None; they never raise."""
    report = report or _default_report
    try:
        file.seek(0)
        first = pd.read_csv(file)
        if "date" in [_norm(c) for c in first.columns]:
            df = first.copy()
        else:                                               # header is not on the first line: sniff 5 rows
            file.seek(0)
            raw = pd.read_csv(file, header=None)
            hdr = next((i for i in range(min(5, len(raw))) if "date" in [_norm(x) for x in raw.iloc[i].tolist()]), None)
            if hdr is None:
                raise ValueError("no header row containing 'date' in the first 5 rows")
            df = raw.iloc[hdr + 1:].reset_index(drop=True)
            df.columns = raw.iloc[hdr].tolist()
        date_cols = [c for c in df.columns if _norm(c) == "date"]
        if not date_cols:
            raise ValueError("no 'Date' column")
        date_col = date_cols[0]
        candidates = [c for c in df.columns if _norm(c) in PRICE_NAMES] or [c for c in df.columns if c != date_col]
        if not candidates:
            raise ValueError("no price column")
        price_col = candidates[0]                            # first in FILE order, not in PRICE_NAMES order
        df = df[[date_col, price_col]].dropna()
        if df.empty:
            raise ValueError("no rows left after dropping empty cells")
        df = df.rename(columns={date_col: "Date", price_col: "Price"})
        df["Date"] = pd.to_datetime(df["Date"], errors="coerce")
        price = df["Price"]
        if not compat and price.dtype == object:
            price = price.astype(str).str.replace(",", "", regex=False).str.strip()
        df["Price"] = pd.to_numeric(price, errors="coerce")
        df = df.dropna(subset=["Date", "Price"])
        if df.empty:
            raise ValueError("no valid rows left after type conversion")
        return df
    except Exception as e:                                  # the reference catches everything here
        report(f"error reading {getattr(file, 'name', '<file>')}: {e}")
        return None


def asset_name(filename: str) -> str:
    """app.py:389: name = file.name.split('.')[0] (Streamlit upload names carry no directory; a path's directory
    part is dropped here so files opened from disk get the same names)."""
    import os
    return os.path.basename(filename).split(".")[0]


def dedupe_names(names):
    """app.py:443-447 / 466-472: the 2nd, 3rd ... occurrence of a name gets ' (2)', ' (3)' ..."""
    seen = Counter()
    out = []
    for n in names:
        seen[n] += 1
        out.append(n if seen[n] == 1 else f"{n} ({seen[n]})")
    return out


_RULE = {"M": "ME", "Q": "QE"}      # pandas >= 2.2 spells month/quarter END this way; same bins as the reference's 'M'/'Q'


def resample_last(obj, rule: str):
    try:
        return obj.resample(_RULE.get(rule, rule)).last()
    except ValueError:               # older pandas
        return obj.resample(rule).last()


def align_prices(named_frames, resample_rule: str = "M"):
    """[(name, DataFrame[Date, Price]), ...] -> (asset_names, prices_df, resampled_prices), app.py:466-482:
    inner join on dates, then last observation of each period, rows with a missing asset dropped."""
    names = dedupe_names([n for n, _ in named_frames])
    cols = []
    for name, (_, df) in zip(names, named_frames):
        t = df.rename(columns={"Price": name}).dropna(subset=[name]).set_index("Date")
        cols.append(t[[name]])
    prices = pd.concat(cols, axis=1, join="inner")
    if not isinstance(prices.index, pd.DatetimeIndex):
        prices.index = pd.to_datetime(prices.index)
    return names, prices, resample_last(prices, resample_rule).dropna()


ANNUAL_FACTOR = {"M": 12, "Q": 4, "W": 52, "D": 252}       # sidebar, app.py:427 (+ daily for configs[0])


def returns_matrix(resampled_prices, option_rows=None):
    """app.py:658-667: per asset `pct_change().fillna(0)` (the leading 0.0 row is KEPT, Q3) or the option
    overlay series when the asset has strategy rows; then dropna."""
    from .options import calc_options_series
    option_rows = option_rows or {}
    cols = {}
    for name in resampled_prices.columns:
        rows = option_rows.get(name, [])
        cols[name] = calc_options_series(rows, resampled_prices[name]) if rows else resampled_prices[name].pct_change().fillna(0)
    return pd.DataFrame(cols).dropna()


def load_prices(files, resample_rule="M", compat=False, report=None):
    """Convenience: uploaded files -> (asset_names, prices_df, resampled_prices); rejected files are skipped
    exactly as the sidebar does (app.py:385-390)."""
    frames = []
    for f in files:
        df = read_csv_file(f, compat=compat, report=report)
        if df is not None:
            frames.append((asset_name(f.name), df))
    if not frames:
        raise ValueError("no file could be read")
    return align_prices(frames, resample_rule)
