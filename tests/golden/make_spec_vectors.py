#!/usr/bin/env python3
"""Freeze SPEC.md (version 4) as data: vectors produced by the CPU oracle, committed so that any later
change of either implementation that moves a bit is caught on CPU (oracle) and on the GPU (kernels).

    python tests/golden/make_spec_vectors.py        # rewrites tests/golden/spec_vectors.npz

Contents: the inverse-CDF table, the first 4,096 normals of path 0 (16 assets x 256 steps, seed
0x5EED5EED), normals of fixed edge + random words, terminal values of three small problems.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from monte_carlo_portfolio_amd import synthetic          # noqa: E402  (pure NumPy helpers)
from oracle import mc_oracle                              # noqa: E402

SEED = synthetic.BENCH_SEED


def problem(N, K):
    mu, cov = synthetic.synthetic_market(N)
    W = synthetic.equal_weights(N)[None, :] if K == 1 else synthetic.dirichlet_weights(N, K)
    L = np.linalg.cholesky(cov)
    return mu.astype(np.float32), np.ascontiguousarray(L, np.float32), np.ascontiguousarray(W, np.float32)


def main():
    out = {}
    out["icdf_table"] = mc_oracle.icdf_table()
    out["normals_path0"] = np.stack([mc_oracle.step_normals(SEED, 0, t, 16) for t in range(256)])
    e = np.array([0, 1, 2, 0xffffffff, 0xfffffffe, 0xffffff80, 0xffffff7f, 0x7fffffff, 0x80000000, 0x3fffffff, 0x40000000,
                  0x0003ffff, 0x00040000, 0x1fffffff, 0x20000000, 0xbfffffff, 0xc0000000, 0x12345678, 0x9abcdef0], np.uint32)
    x = np.concatenate([e, np.random.default_rng(7).integers(0, 2 ** 32, 4096, dtype=np.uint32)])
    out["normal_x"], out["normal_z"] = x, mc_oracle.normals(x)
    for tag, (N, K, T, P, pb, comp) in {"n16": (16, 1, 252, 512, 0, "simple"), "n3k5": (3, 5, 40, 300, (1 << 32) - 100, "simple"),
                                        "n64log": (64, 2, 6, 128, 7, "log")}.items():
        mu, L, W = problem(N, K)
        out[f"terminal_{tag}"] = mc_oracle.simulate(mu, L, W, T, P, SEED, path_begin=pb, compounding=comp)
        out[f"params_{tag}"] = np.array([N, K, T, P, pb, comp == "log"], np.int64)
    np.savez_compressed(os.path.join(HERE, "spec_vectors.npz"), **out)
    print("wrote spec_vectors.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
