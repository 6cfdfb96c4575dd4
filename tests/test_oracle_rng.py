"""Pins the oracle's RNG layer (CPU only).

Philox4x32-10: Random123 known-answer vectors (kat_vectors of the Random123 distribution, the
library rocRAND's engine cites at rocrand_philox4x32_10.h:287) and, when the ROCm headers and a
host compiler are present, rocRAND's own host-callable engine.
Box-Muller: against float64 math on the spec's u, and distribution moments.
"""
import os
import shutil
import subprocess
import textwrap

import numpy as np
import pytest

KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox_known_answers(oracle, ctr, key, want):
    assert [int(v) for v in oracle.philox4x32_10(ctr, key)] == want


def test_philox_matches_rocrand_host_engine(oracle, tmp_path):
    """rocrand_init(seed, subsequence=p, offset=0) + 4 draws per block == SPEC counter layout."""
    hdr = "/opt/rocm/include/rocrand/rocrand_philox4x32_10.h"
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not (os.path.exists(hdr) and os.path.exists(hipcc)):
        pytest.skip("rocRAND headers / hipcc not available")
    src = tmp_path / "rr.cpp"
    src.write_text(textwrap.dedent(r"""
        #include <hip/hip_runtime.h>
        #include <rocrand/rocrand_philox4x32_10.h>
        #include <cstdio>
        int main() {
          const unsigned long long seeds[2] = {0x5EED5EEDull, 0x0123456789abcdefull};
          const unsigned long long paths[3] = {0ull, 12345ull, 0x100000007ull};
          for (auto s : seeds) for (auto p : paths) {
            rocrand_device::philox4x32_10_engine e(s, p, 0);
            for (int b = 0; b < 3; b++) {
              unsigned x0 = e(), x1 = e(), x2 = e(), x3 = e();
              printf("%llu %llu %d %u %u %u %u\n", s, p, b, x0, x1, x2, x3);
            }
          }
          return 0;
        }"""))
    exe = tmp_path / "rr"
    r = subprocess.run([hipcc, "-O1", "--offload-arch=gfx950", str(src), "-o", str(exe)], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("could not build the rocRAND host probe: " + r.stderr[-300:])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    n = 0
    for line in out:
        if not line.strip():
            continue
        s, p, b, *xs = [int(t) for t in line.split()]
        got = oracle.philox4x32_10([b, 0, p & 0xffffffff, p >> 32], [s & 0xffffffff, s >> 32])
        assert [int(v) for v in got] == xs, line
        n += 1
    assert n == 18


def _math_reference(xa, xb):
    u = (xa.astype(np.float32) * np.float32(2.0 ** -32) + np.float32(2.0 ** -32)).astype(np.float64)
    s = np.sqrt(-2.0 * np.log(u))
    th = 2.0 * np.pi * xb.astype(np.float64) / 2.0 ** 32
    return s * np.sin(th), s * np.cos(th), s


def test_box_muller_accuracy_vs_float64_math(oracle):
    rng = np.random.default_rng(1)
    xa = rng.integers(0, 2 ** 32, 1_000_000, dtype=np.uint32)
    xb = rng.integers(0, 2 ** 32, 1_000_000, dtype=np.uint32)
    zs, zc = oracle.box_muller(xa, xb)
    rs, rc, s = _math_reference(xa, xb)
    scale = np.maximum(s, 1.0)
    assert np.max(np.abs(zs - rs) / scale) < 4e-7     # fp32-grade: ~3 ulp of max(|z|, 1)
    assert np.max(np.abs(zc - rc) / scale) < 4e-7


def test_box_muller_edges(oracle):
    e = np.array([0, 1, 2, 0xffffffff, 0xfffffffe, 0x7fffffff, 0x80000000, 0x3fffffff, 0x40000000,
                  0x1fffffff, 0x20000000, 0xdfffffff, 0xe0000000, 0x5fffffff, 0x60000000], np.uint32)
    A, B = [v.ravel().copy() for v in np.meshgrid(e, e)]
    zs, zc = oracle.box_muller(A, B)
    rs, rc, s = _math_reference(A, B)
    assert np.isfinite(zs).all() and np.isfinite(zc).all()
    assert np.max(np.abs(zs - rs)) < 1e-6 and np.max(np.abs(zc - rc)) < 1e-6
    # u == 1.0 (xa = 0xffffffff rounds up): radius exactly 0
    assert np.all(zs[A == 0xffffffff] == 0) and np.all(zc[A == 0xffffffff] == 0)


def test_normals_moments(oracle):
    z = np.concatenate([oracle.step_normals(0x5EED5EED, p, t, 16) for p in range(200) for t in range(50)])
    n = z.size
    assert abs(z.mean()) < 4 / np.sqrt(n)
    assert abs(z.std() - 1) < 4 / np.sqrt(2 * n)
    assert abs(((z - z.mean()) ** 4).mean() / z.var() ** 2 - 3) < 0.1


def test_normal_layout(oracle):
    """z[m*nb + q] = normal m of Philox block q; counter = (t*nb+q, 0, p_lo, p_hi), key = seed."""
    seed, p, t, N = 0x0123456789abcdef, (5 << 32) | 77, 9, 10
    nb = 3
    z = oracle.step_normals(seed, p, t, N)
    assert z.size == 12
    for q in range(nb):
        x = oracle.philox4x32_10([t * nb + q, 0, p & 0xffffffff, p >> 32], [seed & 0xffffffff, seed >> 32])
        s0, c0 = oracle.box_muller(x[0:1], x[1:2])
        s1, c1 = oracle.box_muller(x[2:3], x[3:4])
        assert z[0 * nb + q] == s0[0] and z[1 * nb + q] == c0[0]
        assert z[2 * nb + q] == s1[0] and z[3 * nb + q] == c1[0]


def test_simulate_python_restatement(oracle):
    """The C path loop against an independent pure-NumPy float32 restatement (small case)."""
    rng = np.random.default_rng(3)
    N, T, P, K = 6, 5, 7, 2
    mu = rng.normal(0, 1e-3, N).astype(np.float32)
    L = np.tril(rng.normal(0, 0.02, (N, N))).astype(np.float32)
    W = rng.dirichlet(np.ones(N), K).astype(np.float32)
    seed, pb = 99, 1234567
    got = oracle.simulate(mu, L, W, T, P, seed, path_begin=pb, v0=2.0)
    f32 = np.float32

    def fma(a, b, c):   # exact product in float64, single rounding to float32 (no double-rounding risk at these sizes is asserted by equality)
        return f32(np.float64(a) * np.float64(b) + np.float64(c))

    for p in range(P):
        V = [f32(2.0)] * K
        for t in range(T):
            z = oracle.step_normals(seed, pb + p, t, N)
            r = []
            for i in range(N):
                acc = mu[i]
                for j in range(i + 1):
                    acc = fma(L[i, j], z[j], acc)
                r.append(acc)
            for k in range(K):
                rho = f32(0)
                for i in range(N):
                    rho = fma(W[k, i], r[i], rho)
                V[k] = fma(V[k], rho, V[k])
        for k in range(K):
            assert V[k] == got[k, p]


def test_partition_invariance(oracle):
    mu = np.full(4, 1e-4, np.float32)
    L = np.tril(np.full((4, 4), 0.01, np.float32))
    w = np.full((1, 4), 0.25, np.float32)
    whole = oracle.simulate(mu, L, w, 10, 64, 5)
    parts = np.concatenate([oracle.simulate(mu, L, w, 10, 16, 5, path_begin=16 * g) for g in range(4)], axis=1)
    assert np.array_equal(whole, parts)
    assert np.array_equal(whole, oracle.simulate(mu, L, w, 10, 64, 5, n_threads=1))
