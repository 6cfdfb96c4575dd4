"""Pure-NumPy restatement of SPEC.md ("MC-A") -- ORACLE, test infrastructure only (see oracle/__init__.py).

This is the "reference NumPy CPU loop" BASELINE.json's north_star speaks of.  The reference (app.py) holds no
path simulator, so what is restated here is SPEC.md, with the conventions the reference
does pin cited inline: fixed-weight portfolio return `returns_df @ ws` (app.py:710), compounding
`np.cumprod(1 + returns)` (app.py:253), statistics in binary64 (app.py:258-263, 711; oracle/ref_stats.py).

Three evaluations of the same draws (identical Philox words, identical binary32 inverse-CDF normals):

  simulate(..., dtype=np.float32, exact=True)    binary32 with emulated single-rounding fma in the spec's order:
                                                 bit-identical to oracle/mc_oracle.c and to the HIP kernels
                                                 (tests/test_np_oracle.py); slow (Python loop over steps and columns).
  simulate(..., dtype=np.float32, exact=False)   what a NumPy user would write: `Z @ L.T`, `r @ W.T`, running product,
                                                 in float32; same model, BLAS summation order -> agrees to ~1e-6.
  simulate(..., dtype=np.float64)                the same in float64 (NumPy's default precision, i.e. the reference's
                                                 arithmetic): the yardstick for the fp32 kernels' rounding drift, and
                                                 the NumPy CPU baseline bench.py times (`cpu_baseline_numpy`).

PARITY STATUS: "parity unpinned" against reference code for the path loop (there is none); the Philox layer is pinned
by the Random123 known-answer vectors, the statistics by goldens generated from the reference.
"""
from __future__ import annotations

import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)
ICDF_E_LO = 94
_TABLE = None


def icdf_table() -> np.ndarray:
    """[1056, 4] binary32 coefficients of SPEC.md section 3.1, parsed from the oracle's own copy of the table."""
    global _TABLE
    if _TABLE is None:
        vals = []
        for line in open(os.path.join(_HERE, "icdf_table.inc")):
            line = line.split("//")[0].strip().strip(",").replace("{", "").replace("}", "")
            if not line:
                continue
            vals.extend(float.fromhex(t.strip().rstrip("f")) for t in line.split(",") if t.strip())
        _TABLE = np.asarray(vals, np.float64).astype(np.float32).reshape(1056, 4)
    return _TABLE


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Vectorised Philox4x32-10 (SPEC.md section 2): uint32 arrays in, four uint32 arrays out."""
    c0, c1, c2, c3 = (np.asarray(c, np.uint64) for c in np.broadcast_arrays(c0, c1, c2, c3))
    for r in range(10):
        kk0 = np.uint64((k0 + r * W0) & 0xFFFFFFFF)
        kk1 = np.uint64((k1 + r * W1) & 0xFFFFFFFF)
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = (p1 >> _S32) ^ c1 ^ kk0, p1 & _MASK, (p0 >> _S32) ^ c3 ^ kk1, p0 & _MASK
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def _fma32(a, b, c):
    """Correctly rounded binary32 fma(a, b, c) on float32 arrays.  a*b is exact in binary64; the sum is formed with
    round-to-odd (TwoSum error term), so the final rounding to binary32 is a single rounding."""
    p = a.astype(np.float64) * b.astype(np.float64)
    c = np.asarray(c, np.float32).astype(np.float64)
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)                      # s + err == p + c exactly
    bits = s.view(np.int64)
    fix = (err != 0) & ((bits & 1) == 0) & np.isfinite(s)
    toward = np.where((err > 0) == (s > 0), 1, -1)       # one ulp away from / toward zero in the integer encoding
    bits = np.where(fix, bits + toward, bits)
    return bits.view(np.float64).astype(np.float32)


def normals(x) -> np.ndarray:
    """SPEC.md section 3: one 32-bit word -> one N(0,1) draw, table-driven inverse CDF, exact binary32 arithmetic."""
    x = np.asarray(x, np.uint32)
    tab = icdf_table()
    v = (x & np.uint32(0x7FFFFFFF)).astype(np.float32)                      # RNE to 24 bits
    u = _fma32(v, np.float32(2.0 ** -32), np.float32(2.0 ** -33))
    b = u.view(np.uint32) - np.uint32(ICDF_E_LO << 23)
    c = tab[b >> np.uint32(18)]
    dc = ((b & np.uint32(0x3FFFF)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.015625)   # 0x1.04p+0
    a = _fma32(c[..., 3], dc, c[..., 2])
    a = _fma32(a, dc, c[..., 1])
    a = _fma32(a, dc, c[..., 0])
    return ((a.view(np.uint32) & np.uint32(0x7FFFFFFF)) | (x & np.uint32(0x80000000))).view(np.float32)


def step_normals(seed: int, paths: np.ndarray, step: int, n_assets: int) -> np.ndarray:
    """z [n_paths, N4] of one step: z[:, m*nb + q] = normal m of Philox block q (SPEC.md section 2)."""
    nb = (n_assets + 3) // 4
    paths = np.asarray(paths, np.uint64)
    plo, phi = (paths & _MASK), (paths >> _S32)
    z = np.empty((paths.shape[0], 4 * nb), np.float32)
    for q in range(nb):
        xs = philox4x32_10(np.uint64(step * nb + q), np.uint64(0), plo, phi, seed & 0xFFFFFFFF, seed >> 32)
        for m in range(4):
            z[:, m * nb + q] = normals(xs[m])
    return z


def simulate(mu, chol, W, n_steps: int, n_paths: int, seed: int, path_begin: int = 0, v0: float = 1.0,
             compounding: str = "simple", dtype=np.float32, exact: bool = False, chunk: int = 65536) -> np.ndarray:
    """Terminal values [K, n_paths] of dtype (V_T for 'simple', sum of rho for 'log'), inputs as the kernels get them:
    mu [N], chol [N, N] lower, W [K, N], all binary32."""
    mu = np.asarray(mu, np.float32) + np.float32(0)
    L = np.tril(np.asarray(chol, np.float32))
    W = np.atleast_2d(np.asarray(W, np.float32))
    N, K = mu.shape[0], W.shape[0]
    dtype = np.dtype(dtype)
    if exact and dtype != np.float32:
        raise ValueError("exact=True is the binary32 spec order")
    out = np.empty((K, n_paths), dtype)
    muD, LD, WD = mu.astype(dtype), L.astype(dtype), W.astype(dtype)
    for lo in range(0, n_paths, chunk):
        hi = min(lo + chunk, n_paths)
        paths = np.arange(path_begin + lo, path_begin + hi, dtype=np.uint64)
        n = hi - lo
        V = np.full((n, K), 0.0 if compounding == "log" else np.float32(v0), dtype)
        for t in range(n_steps):
            z = step_normals(seed, paths, t, N)[:, :N]
            if exact:
                r = np.empty((n, N), np.float32)
                for i in range(N):                                   # r_i = mu_i + sum_j L_ij z_j, j ascending, fma
                    acc = np.full(n, mu[i], np.float32)
                    for j in range(i + 1):
                        acc = _fma32(np.full(n, L[i, j], np.float32), z[:, j], acc)
                    r[:, i] = acc
                for k in range(K):                                   # rho = sum_i w_i r_i, i ascending, fma
                    rho = np.zeros(n, np.float32)
                    for i in range(N):
                        rho = _fma32(np.full(n, W[k, i], np.float32), r[:, i], rho)
                    V[:, k] = V[:, k] + rho if compounding == "log" else _fma32(V[:, k], rho, V[:, k])
            else:
                r = muD + z.astype(dtype) @ LD.T                     # correlated draws (Cholesky factor of app.py:680's cov)
                rho = r @ WD.T                                       # returns @ w, app.py:710
                V = V + rho if compounding == "log" else V * (1 + rho)   # np.cumprod(1 + r) idiom, app.py:253
        out[:, lo:hi] = V.T
    return out
