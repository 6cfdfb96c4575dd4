"""CSV-in without pandas: the critical path file -> returns matrix -> (mu, Sigma) in the standard library + NumPy.

SURVEY.md section 8f-2.  Restates app.py:89-134 (`read_csv_file`), the tab-0 alignment app.py:466-482 (inner join on
dates, last observation of each period) and the returns matrix of app.py:658-667 (`pct_change().fillna(0)`, leading
0.0 row kept) plus `mean() * annual_factor` / `cov() * annual_factor` of app.py:679-680 -- with the arithmetic of the
pandas routines the reference calls restated so that the numbers are the reference's own, bit for bit (pinned by the
reference-generated goldens G1-G3 and by `ingest.py`, the pandas twin, on every shipped CSV):

  * number parsing: pandas' C tokenizer converter (`precise_xstrtod`: 17 digits accumulated in binary64, one multiply or
    divide by a tabulated power of ten);
  * `DataFrame.mean()`: a pairwise (NumPy) sum over each column's contiguous block;
  * `DataFrame.cov()`: `np.cov(values.T, ddof=1)`, which is what pandas calls for a frame without missing values.

`compat=True` reproduces quirk Q1 (a thousands separator makes the whole column text, which then fails to convert and is
dropped); the default strips separators.  Dates: the format is inferred from the first row as pandas does for these
files (month first unless the first field exceeds 12); unparsable rows are dropped (`errors='coerce'`).

Scope of the date parser.  `pd.to_datetime` guesses among many more layouts than a price export ever uses; this module
accepts the ones in `_DATE_FORMATS` (the layouts of the 16 shipped files and of the common exchange / Yahoo / investing.com
exports, ISO timestamps included) and says so when it meets another one: such a file is REJECTED with a message naming the
supported layouts (the pandas twin `ingest.read_csv_file` accepts whatever pandas accepts).  Pinned against pandas on every
listed layout (`tests/test_ingest_np.py::test_date_layouts_agree_with_pandas`).  Two more documented differences: a date
that occurs twice in one file keeps its first price (pandas' inner join would pair every duplicate), and a 0/0 return is
0.0 as `fillna(0)` makes it.
"""
from __future__ import annotations

import csv
import datetime as _dt
import io
import warnings
from collections import Counter

import numpy as np

PRICE_NAMES = ("price", "close", "adj close", "open")
ANNUAL_FACTOR = {"M": 12, "Q": 4, "W": 52, "D": 252}
_DATE_FORMATS = ("%m/%d/%Y", "%d/%m/%Y", "%Y-%m-%d", "%Y/%m/%d", "%b %d, %Y", "%d-%m-%Y", "%Y-%m-%d %H:%M:%S", "%m/%d/%y",
                 "%Y-%m-%dT%H:%M:%S", "%Y-%m-%dT%H:%M:%SZ", "%Y-%m-%dT%H:%M:%S.%f", "%Y-%m-%d %H:%M:%S.%f", "%Y-%m-%d %H:%M",
                 "%d.%m.%Y", "%Y%m%d", "%b %d %Y", "%d %b %Y", "%B %d, %Y", "%d-%b-%Y", "%d-%b-%y")
SUPPORTED_DATE_LAYOUTS = ("01/31/2020 or 31/01/2020 (month first unless the first field exceeds 12), 2020-01-31, 2020/01/31, 31-01-2020, "
                          "31.01.2020, 20200131, Jan 31, 2020, Jan 31 2020, 31 Jan 2020, January 31, 2020, 31-Jan-2020, 31-Jan-20, 01/31/20, "
                          "2020-01-31 16:00[:00[.ffffff]], 2020-01-31T16:00:00[.ffffff][Z]")


def _default_report(msg: str) -> None:
    warnings.warn(msg, stacklevel=3)


def _norm(c) -> str:
    return str(c).strip().lower()


_NA_STRINGS = {"", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>", "N/A", "NA",
               "NULL", "NaN", "None", "n/a", "nan", "null"}            # pandas.read_csv default na_values


def _scan_number(text: str, max_digits):
    """Front end of pandas' tokenizer converter: -> (digits as binary64, decimal exponent) or None; digits beyond
    `max_digits` only shift the exponent."""
    s = text.strip(" \t\r\n\f\v")
    n = len(s)
    i = 0
    neg = False
    if i < n and s[i] in "+-":
        neg = s[i] == "-"
        i += 1
    number, exponent, num_digits, num_decimals = 0.0, 0, 0, 0
    while i < n and "0" <= s[i] <= "9":
        if max_digits is None or num_digits < max_digits:
            number = number * 10.0 + (ord(s[i]) - 48)
            num_digits += 1
        else:
            exponent += 1
        i += 1
    if i < n and s[i] == ".":
        i += 1
        while i < n and "0" <= s[i] <= "9":
            if max_digits is None or num_digits < max_digits:
                number = number * 10.0 + (ord(s[i]) - 48)
                num_digits += 1
                num_decimals += 1
            i += 1
        exponent -= num_decimals
    if num_digits == 0:
        return None
    if i < n and s[i] in "eE":
        j = i + 1
        eneg = False
        if j < n and s[j] in "+-":
            eneg = s[j] == "-"
            j += 1
        if j < n and "0" <= s[j] <= "9":
            e = 0
            while j < n and "0" <= s[j] <= "9":
                e = e * 10 + (ord(s[j]) - 48)
                j += 1
            exponent += -e if eneg else e
            i = j
    if i != n:
        return None                                          # trailing characters (a thousands separator, a unit ...)
    return (-number if neg else number), exponent


def precise_xstrtod(text: str):
    """pandas/_libs/src/parser/tokenizer.c `precise_xstrtod`: the converter behind `pd.read_csv` (`float_precision`
    defaults to 'high' since pandas 1.2) and behind `pd.to_numeric` on text cells (GH 31364).  At most 17 significant
    digits accumulated in binary64, then ONE multiply or divide by a tabulated power of ten -- close to, but not the same
    as, a correctly rounded strtod (`float(text)`) once the digits exceed 2^53 or the exponent exceeds 22.  None if the
    text is not entirely a number."""
    got = _scan_number(text, 17)
    if got is None:
        return None
    number, exponent = got
    if exponent > 308:
        return None
    if exponent > 0:
        return number * float(f"1e{exponent}")
    if exponent < -308:
        return 0.0 if exponent < -616 else number / float(f"1e{-308 - exponent}") / 1e308
    return number / float(f"1e{-exponent}")


def _decode(file) -> str:
    file.seek(0)
    raw = file.read()
    if isinstance(raw, bytes):
        raw = raw.decode("utf-8-sig")
    elif raw.startswith("﻿"):
        raw = raw[1:]
    return raw


def _parse_dates(strings):
    """-> int64 days since 1970-01-01, NaT as the minimum int64.  The format is inferred from the first parsable row."""
    NAT = np.iinfo(np.int64).min
    fmt = None
    for s in strings:                                        # the first row that parses decides (a leading unparsable row makes
        s = s.strip()                                        # pandas parse row by row; for one format per file the result is the same)
        for f in _DATE_FORMATS:
            try:
                _dt.datetime.strptime(s, f)
                fmt = f
                break
            except ValueError:
                continue
        if fmt is not None:
            break
    out = np.full(len(strings), NAT, np.int64)
    if fmt is None:
        return out
    epoch = _dt.date(1970, 1, 1).toordinal()
    for i, s in enumerate(strings):
        try:
            out[i] = _dt.datetime.strptime(s.strip(), fmt).date().toordinal() - epoch
        except ValueError:
            pass
    return out


def read_csv_file(file, compat: bool = False, report=None):
    """file-like (seekable, with .name) -> (dates int64[days since epoch], prices float64) in FILE order, or None.
    app.py:89-134; failures go to `report(message)` and yield None, they never raise."""
    report = report or _default_report
    try:
        rows = [r for r in csv.reader(io.StringIO(_decode(file))) if r]
        if not rows:
            raise ValueError("empty file")
        wide = next((i for i, r in enumerate(rows) if len(r) > len(rows[0])), None)
        if wide is not None:                                  # pandas' tokenizer: the first line fixes the field count
            raise ValueError(f"Error tokenizing data: expected {len(rows[0])} fields in line {wide + 1}, saw {len(rows[wide])}")
        # the reference takes line 0 as the header if it names a date column, else the first of lines 0..4 that does
        hdr = next((i for i in range(min(5, len(rows))) if "date" in [_norm(c) for c in rows[i]]), None)
        if hdr is None:
            raise ValueError("no header row containing 'date' in the first 5 rows")
        header = rows[hdr]
        body = rows[hdr + 1:]
        date_idx = next(i for i, c in enumerate(header) if _norm(c) == "date")
        cand = [i for i, c in enumerate(header) if _norm(c) in PRICE_NAMES] or [i for i in range(len(header)) if i != date_idx]
        if not cand:
            raise ValueError("no price column")
        price_idx = cand[0]                                   # first in FILE order, not in PRICE_NAMES order
        ds, ps = [], []
        for r in body:
            d = r[date_idx] if date_idx < len(r) else ""
            p = r[price_idx] if price_idx < len(r) else ""
            if d.strip() in _NA_STRINGS or p.strip() in _NA_STRINGS:    # dropna on the two columns
                continue
            ds.append(d)
            ps.append(p)
        if not ds:
            raise ValueError("no rows left after dropping empty cells")
        # pandas infers ONE dtype per column: a single non-numeric cell (a thousands separator) makes the column text and
        # pd.to_numeric then converts cell by cell, leaving NaN where it fails (compat, quirk Q1); both routes end in the
        # same converter, so cell-wise conversion reproduces them.  Default mode strips the separators first.
        vals = [precise_xstrtod(p) for p in ps]
        if not compat:
            vals = [v if v is not None else precise_xstrtod(p.replace(",", "")) for v, p in zip(vals, ps)]
        prices = np.array([np.nan if v is None else v for v in vals], np.float64)
        dates = _parse_dates(ds)
        keep = (dates != np.iinfo(np.int64).min) & ~np.isnan(prices)
        if not keep.any():
            if (dates == np.iinfo(np.int64).min).all() and not np.isnan(prices).all():
                raise ValueError(f"no date could be parsed (first value {ds[0]!r}); supported layouts: {SUPPORTED_DATE_LAYOUTS}")
            raise ValueError("no valid rows left after type conversion")
        return dates[keep], prices[keep]
    except Exception as e:                                   # the reference catches everything here
        report(f"error reading {getattr(file, 'name', '<file>')}: {e}")
        return None


def asset_name(filename: str) -> str:
    import os
    return os.path.basename(filename).split(".")[0]          # app.py:389


def dedupe_names(names):
    seen = Counter()
    out = []
    for n in names:
        seen[n] += 1
        out.append(n if seen[n] == 1 else f"{n} ({seen[n]})")   # app.py:443-447
    return out


def _period_key(days: np.ndarray, rule: str) -> np.ndarray:
    """Bin label (as days since epoch of the period END) of each date: pandas 'M' / 'Q' / 'W' (= W-SUN) / 'D'."""
    d = days.astype("datetime64[D]")
    if rule == "D":
        return days.copy()
    if rule == "W":
        weekday = (days + 3) % 7                              # 1970-01-01 was a Thursday; Monday = 0
        return days + (6 - weekday)                           # the Sunday that closes the week
    m = d.astype("datetime64[M]")
    if rule == "Q":
        mi = m.astype(np.int64)
        m = (mi - mi % 3 + 2).astype("datetime64[M]")         # last month of the quarter
    elif rule != "M":
        raise ValueError(f"resample rule {rule!r} not supported (M, Q, W, D)")
    return ((m + 1).astype("datetime64[D]") - 1).astype(np.int64)


def align_prices(named_series, resample_rule: str = "M"):
    """[(name, (dates, prices)), ...] -> (names, period_end_days int64[R], resampled float64[R, N]); app.py:466-482:
    inner join on the dates, then per period the LAST observation in time order, periods without data dropped."""
    names = dedupe_names([n for n, _ in named_series])
    common = None
    for _, (d, _p) in named_series:
        common = set(d.tolist()) if common is None else common & set(d.tolist())
    days = np.array(sorted(common), np.int64)
    if days.size == 0:
        return names, days, np.empty((0, len(names)))
    cols = []
    for _, (d, p) in named_series:
        first = {}
        for di, pi in zip(d.tolist(), p.tolist()):             # a duplicated date: the join keeps every pairing; the
            first.setdefault(di, pi)                            # shipped files have none, the first occurrence is used
        cols.append(np.array([first[x] for x in days.tolist()], np.float64))
    P = np.stack(cols, axis=1)
    key = _period_key(days, resample_rule)
    last = np.r_[key[1:] != key[:-1], True]                     # days ascending: last row of each run of equal keys
    return names, key[last], P[last]


def returns_matrix(resampled: np.ndarray) -> np.ndarray:
    """app.py:666-667: `pct_change().fillna(0)` per column, first row 0.0 kept.  0/0 (two consecutive zero prices) is NaN in
    `pct_change` and 0.0 after `fillna(0)`; x/0 stays +-inf there and here."""
    R = np.zeros_like(resampled)
    if resampled.shape[0] > 1:
        with np.errstate(divide="ignore", invalid="ignore"):
            R[1:] = resampled[1:] / resampled[:-1] - 1.0
        R[np.isnan(R)] = 0.0
    return R


def pandas_mean(R: np.ndarray) -> np.ndarray:
    """DataFrame.mean(): NumPy's pairwise sum along each column's contiguous block, divided by the count."""
    return np.ascontiguousarray(R.T).sum(axis=1) / R.shape[0]


def pandas_cov(R: np.ndarray) -> np.ndarray:
    """DataFrame.cov() of a frame without missing values (the returns matrix never has any: app.py:667 drops them):
    pandas itself calls `np.cov(mat.T, ddof=1)` in that case (pandas/core/frame.py, DataFrame.cov), so this is the
    same routine -- on the same memory layout (a frame keeps each column contiguous, so `mat.T` is C-ordered [N, R]; BLAS
    picks its kernel by layout and the last bit follows) -- hence the same bits."""
    return np.atleast_2d(np.cov(np.ascontiguousarray(np.asarray(R, np.float64).T), ddof=1))


def sweep_inputs(R: np.ndarray, annual_factor: int):
    """app.py:679-680: (mean_returns, cov_matrix), annualised."""
    return pandas_mean(R) * annual_factor, pandas_cov(R) * annual_factor


def load_returns(files, resample_rule="M", compat=False, report=None):
    """Uploaded files -> (names, period_end_days, prices [R, N], returns [R, N]) without pandas; rejected files are
    skipped exactly as the sidebar does (app.py:385-390)."""
    series = []
    for f in files:
        got = read_csv_file(f, compat=compat, report=report)
        if got is not None:
            series.append((asset_name(f.name), got))
    if not series:
        raise ValueError("no file could be read")
    names, days, P = align_prices(series, resample_rule)
    return names, days, P, returns_matrix(P)
