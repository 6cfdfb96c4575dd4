"""monte_carlo_portfolio_amd -- MI355X-native Monte Carlo portfolio path engine.

Host side of the hot path named by BASELINE.json: a thin ctypes layer over libmcport.so
(hand-written HIP for gfx950) with the reference's function surface (app.py) above it, under the
reference's own names so an app.py-shaped Streamlit script can import them unchanged.
"""
from ._ffi import McpError, build, lib  # noqa: F401
from . import ingest_np  # noqa: F401  (the same ingest without pandas: file -> returns matrix -> mu, Sigma)
from .ingest import align_prices, load_prices, read_csv_file, returns_matrix  # noqa: F401
from .metrics import (annual_return, annual_volatility, calc_asset_stats, cvar, max_drawdown, sharpe_ratio,  # noqa: F401
                      sortino_ratio, stats_table, var)
from .options import (calc_option_return, calc_options_series, calculate_breakeven, calculate_payoff,  # noqa: F401
                      calculate_profit_loss_percent, strategy_rows)
from .simulate import Context, simulate_paths, simulate_sweep  # noqa: F401
from .sweep import allocation, efficient_frontier, run_all_methods, run_sweep  # noqa: F401

__all__ = [
    "McpError", "build", "lib", "Context", "simulate_paths", "simulate_sweep", "run_sweep", "run_all_methods", "efficient_frontier",
    "allocation", "ingest_np", "read_csv_file", "align_prices", "load_prices", "returns_matrix", "calc_asset_stats", "stats_table",
    "sharpe_ratio", "sortino_ratio", "annual_volatility", "annual_return", "max_drawdown", "var", "cvar",
    "calc_option_return", "calc_options_series", "calculate_payoff", "calculate_breakeven",
    "calculate_profit_loss_percent", "strategy_rows",
]
