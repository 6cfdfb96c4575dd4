"""Options / hedging overlay and the P/L curve (app.py:164-229, strategy table app.py:515-581).

A strategy is a list of rows `(row_type, strike, premium, qty)`; `row_type` is one of the seven Persian
labels the reference's UI uses (kept verbatim so saved rows interoperate).  Host logic, vectorised over
the time axis / the price grid with the reference's accumulation order (rows in list order, starting
from 0), so results are bit-identical to the scalar loops.
Quirks kept: Q13 `calculate_breakeven` returns at the FIRST put/call row; premiums are fractions of the
purchase price in the payoff functions but absolute in `calc_option_return`.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

BUY_ASSET, SELL_ASSET = "خرید دارایی", "فروش دارایی"
LONG_CALL, SHORT_CALL = "خرید کال", "فروش کال"
LONG_PUT, SHORT_PUT = "خرید پوت", "فروش پوت"
SHORT_FUTURES = "فروش فیوچرز"
ROW_TYPES = (BUY_ASSET, SELL_ASSET, LONG_CALL, SHORT_CALL, LONG_PUT, SHORT_PUT, SHORT_FUTURES)


def _row_return(row_type, price, prev, strike, premium):
    """One period's return of one unit of a row (app.py:164-180); arrays or scalars; 0 where prev == 0."""
    if row_type == BUY_ASSET:
        num = price - prev
    elif row_type in (SELL_ASSET, SHORT_FUTURES):
        num = prev - price
    elif row_type == LONG_CALL:
        num = np.maximum(price - strike, 0) - premium
    elif row_type == SHORT_CALL:
        num = premium - np.maximum(price - strike, 0)
    elif row_type == LONG_PUT:
        num = np.maximum(strike - price, 0) - premium
    elif row_type == SHORT_PUT:
        num = premium - np.maximum(strike - price, 0)
    else:
        return np.zeros_like(np.asarray(price, float)) if np.ndim(price) else 0
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where(prev != 0, num / np.where(prev != 0, prev, 1), 0)
    return out if np.ndim(out) else out.item()


def calc_option_return(row_type, price, prev_price, strike, premium, qty):
    """app.py:164-180 (scalar form; `qty` is accepted and ignored exactly like the reference, Q13)."""
    return _row_return(row_type, price, prev_price, strike, premium)


def calc_options_series(option_rows, prices: pd.Series) -> pd.Series:
    """app.py:182-193: return series of a strategy over a price series; element 0 is 0.0."""
    p = prices.to_numpy(dtype=float)
    rets = np.zeros(len(p))
    if len(p) > 1:
        cur, prev = p[1:], p[:-1]
        acc = 0
        for row_type, strike, premium, qty in option_rows:          # rows accumulate in list order
            acc = acc + qty * _row_return(row_type, cur, prev, strike, premium)
        rets[1:] = acc
    return pd.Series(rets, index=prices.index)


def _total_premium(option_rows, purchase_price):
    return sum(qty * premium * purchase_price for _, _, premium, qty in option_rows if premium != 0)


def calculate_payoff(option_rows, current_price, purchase_price, price_range):
    """app.py:195-216: P/L of the position at each price of the grid (list of floats)."""
    grid = np.asarray(price_range, float)
    total = 0
    for row_type, strike, premium, qty in option_rows:
        if row_type == BUY_ASSET:
            leg = qty * (grid - purchase_price)
        elif row_type in (SELL_ASSET, SHORT_FUTURES):
            leg = qty * (purchase_price - grid)
        elif row_type == LONG_CALL:
            leg = qty * (np.maximum(grid - strike, 0) - premium * purchase_price)
        elif row_type == SHORT_CALL:
            leg = qty * (premium * purchase_price - np.maximum(grid - strike, 0))
        elif row_type == LONG_PUT:
            leg = qty * (np.maximum(strike - grid, 0) - premium * purchase_price)
        elif row_type == SHORT_PUT:
            leg = qty * (premium * purchase_price - np.maximum(strike - grid, 0))
        else:
            continue
        total = total + leg
    out = (total - _total_premium(option_rows, purchase_price)) + np.zeros_like(grid)
    return [float(v) for v in out]


def calculate_breakeven(option_rows, purchase_price):
    """app.py:218-225."""
    prem = _total_premium(option_rows, purchase_price)
    for row_type, strike, premium, qty in option_rows:
        if row_type in (LONG_PUT, LONG_CALL):
            return strike + (prem / qty) if qty != 0 else purchase_price
        if row_type in (SHORT_PUT, SHORT_CALL):
            return strike - (prem / qty) if qty != 0 else purchase_price
    return purchase_price + prem


def calculate_profit_loss_percent(payoffs, purchase_price, qty_asset):
    """app.py:227-229."""
    inv = purchase_price * qty_asset
    return [(p / inv) * 100 if inv != 0 else 0 for p in payoffs]


def payoff_grid(current_price, n=100):
    """app.py:593: np.linspace(0.5 S, 1.5 S, 100)."""
    return np.linspace(current_price * 0.5, current_price * 1.5, n)


STRATEGIES = ("-", "Married Put", "Protective Put", "Covered Call", "Collar", "Bear Put Spread", "Synthetic Put",
              "Long Straddle/Strangle")


def strategy_rows(strategy, current_price, qty_asset=1.0, qty_contract=1.0, strike_put=None, premium_put=0.0,
                  strike_call=None, premium_call=0.0, strike_put_low=None, premium_put_low=0.0):
    """The UI's strategy -> rows table (app.py:515-581) with its default strikes (0.9 S put, 1.1 S call, S for the
    straddle and the synthetic put's call)."""
    S = current_price
    if strategy == "-":
        return []
    if strategy in ("Married Put", "Protective Put"):
        return [(BUY_ASSET, 0, 0, qty_asset), (LONG_PUT, S * 0.9 if strike_put is None else strike_put, premium_put, qty_contract)]
    if strategy == "Covered Call":
        return [(SHORT_CALL, S * 1.1 if strike_call is None else strike_call, premium_call, qty_contract)]
    if strategy == "Collar":
        return [(LONG_PUT, S * 0.9 if strike_put is None else strike_put, premium_put, qty_contract),
                (SHORT_CALL, S * 1.1 if strike_call is None else strike_call, premium_call, qty_contract)]
    if strategy == "Bear Put Spread":
        return [(LONG_PUT, S if strike_put is None else strike_put, premium_put, qty_contract),
                (SHORT_PUT, S * 0.9 if strike_put_low is None else strike_put_low, premium_put_low, qty_contract)]
    if strategy == "Synthetic Put":
        return [(SHORT_FUTURES, 0, 0, qty_asset), (LONG_CALL, S if strike_call is None else strike_call, premium_call, qty_contract)]
    if strategy == "Long Straddle/Strangle":
        return [(LONG_CALL, S if strike_call is None else strike_call, premium_call, qty_contract),
                (LONG_PUT, S if strike_put is None else strike_put, premium_put, qty_contract)]
    raise ValueError(f"unknown strategy {strategy!r}")
