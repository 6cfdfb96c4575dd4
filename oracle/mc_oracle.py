"""ctypes loader for oracle/libmcoracle.so (the C restatement of SPEC.md, see mc_oracle.c).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    if os.environ.get("MCO_LIB_PATH"):            # a sanitizer build of the oracle (tools/asan.sh)
        return os.environ["MCO_LIB_PATH"]
    so = os.path.join(_HERE, "libmcoracle.so")
    src = os.path.join(_HERE, "mc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libmcoracle.so"], check=True, capture_output=True)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.mco_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        L.mco_philox4x32_10.restype = None
        L.mco_normals_n.argtypes = [_u32p, _f32p, ctypes.c_uint64]
        L.mco_normals_n.restype = None
        L.mco_icdf_table.argtypes = [_f32p]
        L.mco_icdf_table.restype = None
        L.mco_step_normals.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, _f32p]
        L.mco_step_normals.restype = None
        L.mco_simulate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                   _f32p, _f32p, _f32p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                   _f32p, ctypes.c_int]
        L.mco_simulate.restype = ctypes.c_int
        L.mco_simulate_f64.argtypes = L.mco_simulate.argtypes[:11] + [np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS"),
                                                                      ctypes.c_int]
        L.mco_simulate_f64.restype = ctypes.c_int
        _LIB = L
    return _LIB


def philox4x32_10(ctr, key) -> np.ndarray:
    out = np.zeros(4, np.uint32)
    lib().mco_philox4x32_10(np.ascontiguousarray(ctr, np.uint32), np.ascontiguousarray(key, np.uint32), out)
    return out


def normals(x) -> np.ndarray:
    """N(0,1) draws of 32-bit words (SPEC.md section 3: inverse CDF)."""
    x = np.ascontiguousarray(x, np.uint32)
    z = np.empty(x.shape, np.float32)
    lib().mco_normals_n(x, z, x.size)
    return z


def icdf_table() -> np.ndarray:
    """[1056, 4] float32 coefficient table of SPEC.md section 3."""
    t = np.zeros((1056, 4), np.float32)
    lib().mco_icdf_table(t)
    return t


def step_normals(seed: int, path: int, step: int, n_assets: int) -> np.ndarray:
    z = np.zeros(4 * ((n_assets + 3) // 4), np.float32)
    lib().mco_step_normals(seed, path, step, n_assets, z)
    return z


def simulate(mu, chol, W, n_steps: int, n_paths: int, seed: int, path_begin: int = 0, v0: float = 1.0,
             compounding: str = "simple", n_threads: int | None = None, fold: bool = False) -> np.ndarray:
    """Terminal values [K, n_paths] float32 (V_T for 'simple'; sum of rho for 'log')."""
    mu = np.ascontiguousarray(mu, np.float32)
    chol = np.ascontiguousarray(chol, np.float32)
    W = np.ascontiguousarray(np.atleast_2d(W), np.float32)
    n = mu.shape[0]
    assert chol.shape == (n, n) and W.shape[1] == n
    K = W.shape[0]
    out = np.empty((K, n_paths), np.float32)
    if n_threads is None:
        n_threads = min(os.cpu_count() or 1, 64)
    if fold and K != 1:
        raise ValueError("fold needs exactly one portfolio")
    rc = lib().mco_simulate(n, n_steps, K, {"simple": 0, "log": 1}[compounding] | (2 if fold else 0), v0, mu, chol, W,
                            seed, path_begin, n_paths, out, n_threads)
    if rc != 0:
        raise ValueError(f"mco_simulate failed rc={rc}")
    return out


def simulate_f64(mu, chol, W, n_steps: int, n_paths: int, seed: int, path_begin: int = 0, v0: float = 1.0,
                 compounding: str = "simple", n_threads: int | None = None) -> np.ndarray:
    """Terminal values [K, n_paths] float64: the same Philox words, the same fp32 normals and the same binary32
    inputs as simulate(), recurrence in binary64 (mc_oracle.c, mco_simulate_f64).  Bounds the kernel's rounding."""
    mu = np.ascontiguousarray(mu, np.float32)
    chol = np.ascontiguousarray(chol, np.float32)
    W = np.ascontiguousarray(np.atleast_2d(W), np.float32)
    n = mu.shape[0]
    assert chol.shape == (n, n) and W.shape[1] == n
    out = np.empty((W.shape[0], n_paths), np.float64)
    if n_threads is None:
        n_threads = min(os.cpu_count() or 1, 64)
    rc = lib().mco_simulate_f64(n, n_steps, W.shape[0], {"simple": 0, "log": 1}[compounding], v0, mu, chol, W,
                                seed, path_begin, n_paths, out, n_threads)
    if rc != 0:
        raise ValueError(f"mco_simulate_f64 failed rc={rc}")
    return out
