"""ctypes binding of libmcport.so (C ABI: include/mcport.h).

The library is the product's only compute path: if it is missing or cannot be loaded this module
raises -- there is no NumPy/CPU fallback (a silent fallback would void every parity claim).
"""
from __future__ import annotations

import ctypes
import importlib.util
import os
import subprocess
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCP_LIB_PATH") or os.path.join(_PKG, "libmcport.so")     # MCP_LIB_PATH: a lab build (tools/kernel_lab.py)
CSRC = os.path.join(_PKG, "csrc")

MCP_ABI_VERSION = 3
MCP_MAX_ASSETS = 64
MCP_SELECT_BINS = 2048
MCP_COMPOUND = {"simple": 0, "log": 1}
MCP_FLAG_NATIVE_MATH = 1
MCP_FLAG_FOLD = 2
MCP_FLAG_SHARD_PORTFOLIOS = 4
MCP_E_ARG, MCP_E_NODEVICE, MCP_E_NOMEM, MCP_E_UNSUPPORTED, MCP_E_HIP, MCP_E_COMM = -1, -2, -3, -4, -5, -6
(WS_PARTIALS, WS_RECORD, WS_STATE, WS_HIST, WS_QUANT, WS_STATS, WS_BELOW, WS_PIVOT) = range(8)
WS_COUNT = 8
(EXCHANGE_UNSET, EXCHANGE_NONE, EXCHANGE_RCCL, EXCHANGE_KERNEL, EXCHANGE_P2P) = range(5)


class McpError(RuntimeError):
    """Raised for any negative return code of the C ABI; carries mcp_last_error()."""


class McpParams(ctypes.Structure):
    _fields_ = [
        ("n_assets", ctypes.c_int32), ("n_steps", ctypes.c_int32), ("n_portfolios", ctypes.c_int32),
        ("compounding", ctypes.c_int32), ("flags", ctypes.c_int32), ("reserved", ctypes.c_int32),
        ("v0", ctypes.c_double), ("alpha", ctypes.c_double), ("rf", ctypes.c_double),
    ]


class McpStats(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_uint64), ("n_tail", ctypes.c_uint64), ("mean", ctypes.c_double), ("m2", ctypes.c_double),
        ("std", ctypes.c_double), ("sharpe", ctypes.c_double), ("var", ctypes.c_double), ("cvar", ctypes.c_double),
        ("min", ctypes.c_double), ("max", ctypes.c_double), ("sum_tail", ctypes.c_double),
        ("x_lo", ctypes.c_double), ("x_hi", ctypes.c_double),
    ]


STATS_DTYPE = np.dtype([
    ("n", np.uint64), ("n_tail", np.uint64), ("mean", np.float64), ("m2", np.float64), ("std", np.float64),
    ("sharpe", np.float64), ("var", np.float64), ("cvar", np.float64), ("min", np.float64), ("max", np.float64),
    ("sum_tail", np.float64), ("x_lo", np.float64), ("x_hi", np.float64),
])
assert STATS_DTYPE.itemsize == ctypes.sizeof(McpStats)

RECORD_DTYPE = np.dtype([("n", np.float64), ("sum", np.float64), ("sumsq", np.float64), ("min", np.float64),
                         ("max", np.float64), ("below", np.float64), ("pivot", np.float64), ("pad", np.float64)])
# one moment partial of the path kernels' epilogue (csrc/mcp_stats_kernels.h: MomentPartial)
PARTIAL_DTYPE = np.dtype([("s1", np.float64), ("s2", np.float64), ("vmin", np.float32), ("vmax", np.float32), ("n", np.uint64)])
assert PARTIAL_DTYPE.itemsize == 32
QUANT_DTYPE = np.dtype([("x_lo", np.float64), ("x_hi", np.float64), ("var", np.float64), ("level2", np.float64),
                        ("n_tail", np.uint64), ("pad", np.uint64)])
RECORD_DOUBLES = RECORD_DTYPE.itemsize // 8

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_vp = ctypes.c_void_p
_u64 = ctypes.c_uint64
_int = ctypes.c_int
_PP = ctypes.POINTER(McpParams)

# every symbol include/mcport.h declares: (restype, argtypes)
SIGNATURES = {
    "mcp_abi_version": (_int, []),
    "mcp_device_count": (_int, []),
    "mcp_last_error": (ctypes.c_char_p, []),
    "mcp_ctx_create": (_int, [_int, ctypes.POINTER(_vp)]),
    "mcp_ctx_create_multi": (_int, [ctypes.POINTER(_int), _int, ctypes.POINTER(_vp)]),
    "mcp_ctx_device_count": (_int, [_vp]),
    "mcp_ctx_exchange_mode": (_int, [_vp]),
    "mcp_ctx_exchange_note": (ctypes.c_char_p, [_vp]),
    "mcp_ctx_set_terminal_budget": (_int, [_vp, ctypes.c_size_t]),
    "mcp_ctx_destroy": (None, [_vp]),
    "mcp_simulate": (_int, [_vp, _PP, _f32p, _f32p, _f32p, _u64, _u64, _u64, _vp, _vp]),
    "mcp_sweep_historical": (_int, [_vp, _int, _int, _int, _f64p, _f64p, _f64p, _f64p, ctypes.c_double, ctypes.c_double,
                                    _f64p, _f64p, _f64p, _f64p, _f64p]),
    "mcp_ws_bytes": (ctypes.c_size_t, [_int, _int, _u64]),
    "mcp_moment_slots": (_u64, [_int, _u64]),
    "mcp_pivots": (_int, [_PP, _f32p, _f32p, _f32p, _f64p]),
    "mcp_packed_len": (ctypes.c_size_t, [_int, _int]),
    "mcp_pack_params": (_int, [_int, _int, _f32p, _f32p, _f32p, _f32p, ctypes.c_size_t]),
    "mcp_launch_paths": (_int, [_PP, _vp, _vp, _u64, _u64, _u64, _vp, _u64, _vp, _vp, _vp]),
    "mcp_percentile_rank": (_int, [_u64, ctypes.c_double, ctypes.POINTER(_u64), ctypes.POINTER(_u64),
                                   ctypes.POINTER(ctypes.c_double)]),
    "mcp_launch_pass0": (_int, [_PP, _vp, _u64, _u64, _vp, _vp, _vp, _vp]),
    "mcp_launch_scan": (_int, [_PP, _int, _u64, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mcp_launch_hist": (_int, [_PP, _int, _vp, _u64, _u64, _vp, _vp, _vp, _vp, _vp]),
    "mcp_launch_final": (_int, [_PP, _u64, ctypes.c_double, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mcp_launch_stats": (_int, [_PP, _int, _vp, _vp, _vp, _vp]),
    "mcp_launch_sum_u64": (_int, [ctypes.POINTER(_vp), _int, ctypes.c_size_t, _vp]),
    "mcp_stream_create": (_int, [_int, _int, ctypes.POINTER(_vp)]),
    "mcp_stream_destroy": (_int, [_vp]),
    "mcp_launch_normals": (_int, [_vp, _u64, _vp, _vp]),
    "mcp_icdf_table": (_int, [_f32p, ctypes.c_size_t]),
    "mcp_float_to_key": (ctypes.c_uint32, [ctypes.c_float]),
    "mcp_key_to_float": (ctypes.c_float, [ctypes.c_uint32]),
    "mcp_terminal_to_x": (ctypes.c_double, [_PP, ctypes.c_float]),
}

_LIB = None


def build(force: bool = False, jobs: int = 8) -> str:
    """Compile libmcport.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, f"-j{jobs}"] + (["-B"] if force else [])
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libmcport.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    return LIB_PATH


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, the same as /opt/rocm's).  If libmcport.so pulled in the system copy first, a later
    `import torch` would load a second runtime and find no GPU; so when torch is installed its copy is
    loaded first and libmcport.so's DT_NEEDED binds to it by SONAME."""
    if "torch" in sys.modules:
        return                                  # torch already brought its runtime
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def preload_rccl() -> None:
    """Multi-device contexts load librccl at run time (dlopen("librccl.so.1") inside libmcport.so).  When torch is
    installed its bundled librccl (built against the HIP runtime this process already uses) is loaded first, so the
    library's dlopen resolves to it by SONAME."""
    if os.environ.get("MCP_RCCL_LIB"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
        if os.path.exists(cand):
            try:
                # default (RTLD_LOCAL) binding: the library's later dlopen("librccl.so.1") finds this copy by SONAME all the
                # same, and librccl's symbols stay out of the global scope -- promoted to RTLD_GLOBAL they interpose symbols of
                # a torch imported LATER in the same process, which then aborts at exit ("double free or corruption")
                ctypes.CDLL(cand)
            except OSError:
                pass


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _preload_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C {CSRC} -j8` (or __graft_entry__.build()). "
                "There is no CPU fallback for the path engine.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        got = L.mcp_abi_version()
        if got != MCP_ABI_VERSION:
            raise ImportError(f"libmcport.so ABI {got} != binding ABI {MCP_ABI_VERSION}; rebuild")
        _LIB = L
    return _LIB


def check(rc: int) -> int:
    if rc < 0:
        raise McpError(f"libmcport error {rc}: {lib().mcp_last_error().decode('utf-8', 'replace')}")
    return rc


def make_params(n_assets, n_steps, n_portfolios, compounding="simple", v0=1.0, alpha=0.95, rf=0.0,
                native_math=False, fold=False, shard_portfolios=False) -> McpParams:
    if compounding not in MCP_COMPOUND:
        raise ValueError(f"compounding must be 'simple' or 'log', got {compounding!r}")
    return McpParams(int(n_assets), int(n_steps), int(n_portfolios), MCP_COMPOUND[compounding],
                     (MCP_FLAG_NATIVE_MATH if native_math else 0) | (MCP_FLAG_FOLD if fold else 0)
                     | (MCP_FLAG_SHARD_PORTFOLIOS if shard_portfolios else 0), 0,
                     float(v0), float(alpha), float(rf))


def pack_params(mu: np.ndarray, chol: np.ndarray, W: np.ndarray) -> np.ndarray:
    n = mu.shape[0]
    k = W.shape[0]
    out = np.zeros(lib().mcp_packed_len(n, k), np.float32)
    check(lib().mcp_pack_params(n, k, mu, chol, W, out, out.size))
    return out


def pivots(prm: McpParams, mu: np.ndarray, chol: np.ndarray, W: np.ndarray) -> np.ndarray:
    """[K] shifts of the moments (include/mcport.h: mcp_pivots): the analytic mean of x per portfolio, pure host arithmetic."""
    out = np.zeros(W.shape[0], np.float64)
    check(lib().mcp_pivots(ctypes.byref(prm), mu, chol, W, out))
    return out


def percentile_rank(n_total: int, alpha: float):
    lo, hi, g = _u64(), _u64(), ctypes.c_double()
    check(lib().mcp_percentile_rank(n_total, alpha, ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(g)))
    return lo.value, hi.value, g.value
