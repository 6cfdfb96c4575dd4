"""monte_carlo_portfolio_amd -- MI355X-native Monte Carlo portfolio path engine.

Host side of the hot path named by BASELINE.json: a thin ctypes layer over libmcport.so
(hand-written HIP for gfx950) with the reference's function surface (app.py) above it.
"""
from ._ffi import McpError, build, lib  # noqa: F401
from .simulate import Context, simulate_paths  # noqa: F401

__all__ = ["McpError", "build", "lib", "Context", "simulate_paths"]
