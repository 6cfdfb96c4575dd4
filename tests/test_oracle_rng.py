"""Pins the oracle's RNG layer (CPU only).

Philox4x32-10: Random123 known-answer vectors (kat_vectors of the Random123 distribution, the
library rocRAND's engine cites at rocrand_philox4x32_10.h:287) and, when the ROCm headers and a
host compiler are present, rocRAND's own host-callable engine.
Normal transform (inverse CDF): against float64 scipy.special.ndtri on the spec's u, symmetry, monotonicity, moments.
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


def _math_reference(x):
    """-/+ Phi^-1(u) in float64 on the spec's u (scipy.special.ndtri), sign from bit 31."""
    from scipy.special import ndtri
    u = ((x & np.uint32(0x7fffffff)).astype(np.float32) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)).astype(np.float64)
    return np.where(x >> np.uint32(31), ndtri(u), -ndtri(u)), u


def test_normal_transform_accuracy_vs_float64_math(oracle):
    rng = np.random.default_rng(1)
    x = rng.integers(0, 2 ** 32, 2_000_000, dtype=np.uint32)
    x[:100_000] = rng.integers(0, 1 << 12, 100_000).astype(np.uint32)                       # deepest positive tail
    x[100_000:200_000] = np.uint32(0xffffffff) - rng.integers(0, 1 << 12, 100_000).astype(np.uint32)   # centre, negative side
    z = oracle.normals(x)
    want, u = _math_reference(x)
    assert np.max(np.abs(z - want) / np.maximum(np.abs(want), 1.0)) < 2.5e-7       # fp32-grade everywhere, tails included


def test_normal_transform_edges(oracle):
    e = np.array([0, 1, 2, 0x7fffffff, 0x7ffffffe, 0x7fffff80, 0x7fffff7f, 0x40000000, 0x3fffffff, 0x20000000, 0x1fffffff,
                  0x00040000, 0x0003ffff, 0x00000100, 0x000000ff], np.uint32)
    x = np.concatenate([e, e | np.uint32(0x80000000)])
    z = oracle.normals(x)
    want, u = _math_reference(x)
    assert np.isfinite(z).all() and np.max(np.abs(z - want)) < 6e-7
    assert np.array_equal(z[:e.size], -z[e.size:])                         # exact symmetry: bit 31 is only the sign
    assert z[3] == 0.0 and z[0] == np.float32(6.337958)                    # u == 1/2 -> 0 ; u = 2^-33 -> the largest draw
    # monotone within the positive half: larger v (larger u) -> smaller magnitude
    v = np.sort(np.random.default_rng(2).integers(0, 2 ** 31, 200_000, dtype=np.uint32))
    assert np.all(np.diff(oracle.normals(v)) <= 2e-7)


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
        zz = oracle.normals(x)
        for m in range(4):
            assert z[m * nb + q] == zz[m]


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
