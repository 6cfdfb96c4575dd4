"""Deterministic synthetic inputs for the bench configs of BASELINE.json."""
from __future__ import annotations

import numpy as np

BENCH_SEED = 0x5EED5EED


def synthetic_market(n_assets: int, steps_per_year: int = 252, rng_seed: int = 20250614):
    """(mu_step [N] f64, cov_step [N,N] f64): random correlation matrix, annual vols 15..60 %,
    annual drifts 2..12 %, scaled to one step."""
    rng = np.random.default_rng(rng_seed)
    A = rng.standard_normal((n_assets, n_assets))
    C = A @ A.T / n_assets + 0.5 * np.eye(n_assets)
    d = 1.0 / np.sqrt(np.diag(C))
    corr = C * d[:, None] * d[None, :]
    vol = np.linspace(0.15, 0.60, n_assets)
    drift = np.linspace(0.02, 0.12, n_assets)
    cov = corr * vol[:, None] * vol[None, :] / steps_per_year
    mu = drift / steps_per_year
    return mu, cov


def equal_weights(n_assets: int) -> np.ndarray:
    return np.full(n_assets, 1.0 / n_assets)


def dirichlet_weights(n_assets: int, k: int, seed: int = 7) -> np.ndarray:
    """Same generator family the reference draws from (np.random.dirichlet, app.py:702)."""
    return np.random.RandomState(seed).dirichlet(np.ones(n_assets), size=k)
