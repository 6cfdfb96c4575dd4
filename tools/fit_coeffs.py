#!/usr/bin/env python3
"""Derive the fp32 polynomial coefficients frozen in SPEC.md / mcp_spec.h.

Build-authored (no reference counterpart): weighted least squares on dense Chebyshev nodes followed
by a few Remez-style reweighting sweeps, then rounding to float32 and an error scan with the
polynomial evaluated in float32 Horner/fma order (fma emulated in float64, which is exact for the
product and only double-rounds the sum).  Prints C hex-float literals.
"""
import numpy as np

def fit_weighted(xs, target, weight, deg, iters=40):
    """min max |weight*(target - P)|, Lawson-style iteratively reweighted LSQ."""
    V = np.vander(xs, deg + 1, increasing=True)
    lw = np.ones_like(xs)
    best = None
    for _ in range(iters):
        W = weight * np.sqrt(lw)
        c, *_ = np.linalg.lstsq(V * W[:, None], target * W, rcond=None)
        err = np.abs(weight * (target - V @ c))
        m = err.max()
        if best is None or m < best[0]:
            best = (m, c.copy())
        lw = lw * (err / m + 1e-3)
        lw /= lw.sum()
    return best

def f32(x):
    return np.float32(x)

def fma32(a, b, c):
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))

def hexf(x):
    return float(np.float32(x)).hex()

def main():
    n = 20001
    cheb = np.cos(np.pi * (np.arange(n) + 0.5) / n)

    # ---- log: t_m(f) = -2 log1p(f) = -2 f + f^2 Q(f) on [sqrt(.5)-1, sqrt(2)-1]
    lo, hi = np.sqrt(0.5) - 1.0, np.sqrt(2.0) - 1.0
    fs = 0.5 * (lo + hi) + 0.5 * (hi - lo) * cheb
    Q = np.where(np.abs(fs) > 1e-9, (-2.0 * np.log1p(fs) + 2.0 * fs) / np.where(fs == 0, 1, fs * fs), 1.0)
    for deg in (6, 7, 8, 9):
        m, c = fit_weighted(fs, Q, fs * fs, deg)
        c32 = c.astype(np.float32)
        # evaluate in float32
        ff = fs.astype(np.float32)
        ff = ff[(ff.astype(np.float64) >= lo) & (ff.astype(np.float64) <= hi)]
        q = np.full_like(ff, c32[-1])
        for k in range(deg - 1, -1, -1):
            q = fma32(q, ff, c32[k]).astype(np.float32)
        f2 = (ff * ff).astype(np.float32)
        t = fma32(ff, np.float32(-2.0), (f2 * q).astype(np.float32))
        exact = -2.0 * np.log1p(ff.astype(np.float64))
        abs_err = np.abs(t.astype(np.float64) - exact)
        rel_err = abs_err / np.maximum(np.abs(exact), 1e-300)
        print(f"LOG deg={deg}: fit max w-err={m:.3e}  fp32 eval: max abs={abs_err.max():.3e}  max rel(|t|>1e-3)={rel_err[np.abs(exact)>1e-3].max():.3e}")
        print("   coeffs:", ", ".join(hexf(x) + "f" for x in c32))

    # ---- sin/cos on [-pi/4, pi/4]
    A = np.pi / 4
    xs = A * cheb
    z = xs * xs
    # sin(a) = a + a^3 S(z):  S(z) = (sin(a)-a)/a^3
    S = np.where(np.abs(xs) > 1e-6, (np.sin(xs) - xs) / np.where(xs == 0, 1, xs ** 3), -1 / 6)
    C = np.where(np.abs(xs) > 1e-4, (np.cos(xs) - 1 + 0.5 * z) / np.where(xs == 0, 1, z * z), 1 / 24)
    for deg in (2, 3):
        zs = z
        Vs = np.vander(zs, deg + 1, increasing=True)
        m, c = fit_weighted(zs, S, np.abs(xs) ** 3, deg)
        c32 = c.astype(np.float32)
        a = xs.astype(np.float32); a2 = (a * a).astype(np.float32)
        p = np.full_like(a, c32[-1])
        for k in range(deg - 1, -1, -1):
            p = fma32(p, a2, c32[k]).astype(np.float32)
        s = fma32((a * a2).astype(np.float32), p, a)
        err = np.abs(s.astype(np.float64) - np.sin(a.astype(np.float64)))
        print(f"SIN deg(z)={deg}: fit={m:.3e} fp32 max abs err={err.max():.3e}")
        print("   coeffs:", ", ".join(hexf(x) + "f" for x in c32))
        m, c = fit_weighted(zs, C, z * z, deg)
        c32 = c.astype(np.float32)
        p = np.full_like(a, c32[-1])
        for k in range(deg - 1, -1, -1):
            p = fma32(p, a2, c32[k]).astype(np.float32)
        h = fma32(a2, np.float32(-0.5), np.float32(1.0))
        co = fma32((a2 * a2).astype(np.float32), p, h)
        err = np.abs(co.astype(np.float64) - np.cos(a.astype(np.float64)))
        print(f"COS deg(z)={deg}: fit={m:.3e} fp32 max abs err={err.max():.3e}")
        print("   coeffs:", ", ".join(hexf(x) + "f" for x in c32))

if __name__ == "__main__":
    main()
