"""SPEC.md version 4 frozen as data (tests/golden/spec_vectors.npz, written by make_spec_vectors.py from the
oracle): the oracle must still reproduce it (CPU), and the HIP kernels must reproduce it (GPU)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
V = np.load(os.path.join(HERE, "golden", "spec_vectors.npz"))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_spec_vectors import SEED, problem          # noqa: E402


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


CASES = [t[len("terminal_"):] for t in V.files if t.startswith("terminal_")]


def test_oracle_reproduces_the_frozen_vectors(oracle):
    assert same_bits(oracle.icdf_table(), V["icdf_table"])
    got = np.stack([oracle.step_normals(SEED, 0, t, 16) for t in range(256)])
    assert same_bits(got, V["normals_path0"]) and got.size == 4096
    assert same_bits(oracle.normals(V["normal_x"]), V["normal_z"])
    for tag in CASES:
        N, K, T, P, pb, log = [int(x) for x in V[f"params_{tag}"]]
        mu, L, W = problem(N, K)
        assert same_bits(oracle.simulate(mu, L, W, T, P, SEED, path_begin=pb, compounding="log" if log else "simple"),
                         V[f"terminal_{tag}"]), tag


def test_table_properties(mcp_lib):
    """The table is DATA of the spec: the library's copy, the oracle's copy and the frozen copy are one and the same;
    entry (E, j) evaluated at its centre is -Phi^-1 of the bin's centre."""
    from scipy.special import ndtri
    from monte_carlo_portfolio_amd import _ffi
    T = V["icdf_table"]
    lib_t = np.zeros((1056, 4), np.float32)
    _ffi.check(mcp_lib.mcp_icdf_table(lib_t, lib_t.size))
    assert same_bits(lib_t, T)
    E, j = np.divmod(np.arange(1024), 32)
    centre = 2.0 ** (E + 94 - 127) * (1.0 + j / 32.0 + 1.0 / 64.0)
    assert np.max(np.abs(T[:1024, 0] - (-ndtri(centre))) / np.maximum(1.0, T[:1024, 0])) < 1e-7
    assert np.all(T[1024:] == 0)                              # u == 1/2 -> exactly 0; the rest of octave 126 is unreachable
    assert np.all(np.diff(T[:1024, 0]) < 0)                  # magnitude decreases as u grows


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_gpu_reproduces_the_frozen_terminal_values(gpu_ctx, tag):
    from monte_carlo_portfolio_amd import simulate_paths
    from monte_carlo_portfolio_amd import synthetic
    N, K, T, P, pb, log = [int(x) for x in V[f"params_{tag}"]]
    mu, cov = synthetic.synthetic_market(N)
    W = synthetic.equal_weights(N) if K == 1 else synthetic.dirichlet_weights(N, K)
    got = simulate_paths(mu, cov, W, n_steps=T, n_paths=P, seed=SEED, path_begin=pb, store=True,
                         compounding="log" if log else "simple")
    got = [got] if K == 1 else got
    for k in range(K):
        assert same_bits(got[k]["terminal"], V[f"terminal_{tag}"][k]), (tag, k)


@pytest.mark.gpu
def test_gpu_normals_reproduce_the_frozen_vectors(gpu_ctx):
    import ctypes
    import torch
    from monte_carlo_portfolio_amd import _ffi
    x = V["normal_x"]
    d_x = torch.from_numpy(x.view(np.int32)).cuda()
    z = torch.empty(x.size, dtype=torch.float32, device="cuda")
    _ffi.check(_ffi.lib().mcp_launch_normals(d_x.data_ptr(), x.size, z.data_ptr(),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    assert same_bits(z.cpu().numpy(), V["normal_z"])
