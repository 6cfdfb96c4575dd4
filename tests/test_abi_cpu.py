"""CPU-side checks of the C ABI: the library loads, exports every symbol include/mcport.h declares,
and its host-only helpers behave (no kernel is launched here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from monte_carlo_portfolio_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mcport.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcp_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_ffi.SIGNATURES)


def test_library_exports_every_declared_symbol(mcp_lib):
    raw = ctypes.CDLL(_ffi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(raw, name), name
    assert mcp_lib.mcp_abi_version() == _ffi.MCP_ABI_VERSION
    assert mcp_lib.mcp_device_count() >= 0


def test_struct_layouts(mcp_lib):
    assert ctypes.sizeof(_ffi.McpParams) == 48
    assert ctypes.sizeof(_ffi.McpStats) == 104 == _ffi.STATS_DTYPE.itemsize
    assert _ffi.RECORD_DTYPE.itemsize == 64 and _ffi.QUANT_DTYPE.itemsize == 48
    n = 1_000_000
    assert mcp_lib.mcp_ws_bytes(_ffi.WS_RECORD, 3, n) == 3 * 64
    assert mcp_lib.mcp_ws_bytes(_ffi.WS_QUANT, 3, n) == 3 * 48
    assert mcp_lib.mcp_ws_bytes(_ffi.WS_STATS, 3, n) == 3 * 104
    assert mcp_lib.mcp_ws_bytes(_ffi.WS_HIST, 2, n) == 2 * 2 * 2048 * 8
    assert mcp_lib.mcp_ws_bytes(_ffi.WS_STATE, 5, n) == 5 * 2 * 16
    assert mcp_lib.mcp_ws_bytes(_ffi.WS_PIVOT, 5, n) == 5 * 8
    assert mcp_lib.mcp_ws_bytes(_ffi.WS_BELOW, 2, n) == 2 * 2048 * 8 and mcp_lib.mcp_ws_bytes(_ffi.WS_BELOW, 10_000, n) == 10_000 * 1 * 8
    # moment partials (32 B each): one per workgroup of the one-lane-per-path kernels (256 paths, at most 8,192 workgroups) up to
    # 16 portfolios, one per 64-path wave tile of the MFMA sweep kernels from 17 on
    assert mcp_lib.mcp_moment_slots(1, n) == 3907 and mcp_lib.mcp_moment_slots(16, 10**8) == 8192 and mcp_lib.mcp_moment_slots(1, 0) == 1
    assert mcp_lib.mcp_moment_slots(17, n) == 15625 and mcp_lib.mcp_moment_slots(10_000, 131072) == 2048
    assert mcp_lib.mcp_ws_bytes(_ffi.WS_PARTIALS, 2, n) == 2 * 3907 * 32 and mcp_lib.mcp_ws_bytes(_ffi.WS_PARTIALS, 10_000, 131072) == 10_000 * 2048 * 32
    assert _ffi.PARTIAL_DTYPE.itemsize == 32
    assert mcp_lib.mcp_ws_bytes(_ffi.WS_COUNT, 5, n) == 0 and mcp_lib.mcp_ws_bytes(0, 0, n) == 0


def test_pack_params_layout(mcp_lib):
    rng = np.random.default_rng(0)
    N, K = 6, 3
    mu = rng.normal(size=N).astype(np.float32)
    mu[2] = -0.0
    L = rng.normal(size=(N, N)).astype(np.float32)      # upper part must be ignored
    W = rng.normal(size=(K, N)).astype(np.float32)
    p = _ffi.pack_params(mu, L, W)
    n4 = 8
    kpad = 8                                          # K < 17: whole 8-portfolio passes; K >= 17: whole 512-portfolio workgroups
    assert p.size == n4 + n4 * (n4 // 2 + 1) + kpad * n4 + 4 + n4 == mcp_lib.mcp_packed_len(N, K)
    assert mcp_lib.mcp_packed_len(N, 17) == n4 + n4 * (n4 // 2 + 1) + 512 * n4 + 4 + n4
    assert mcp_lib.mcp_packed_len(N, 513) == n4 + n4 * (n4 // 2 + 1) + 1024 * n4 + 4 + n4
    assert np.array_equal(p[:N], mu) and not np.signbit(p[2]) and np.all(p[N:n4] == 0)
    Lp = p[n4:n4 + n4 * (n4 // 2 + 1)]
    for m in range(n4 // 2):                      # row pairs (2m, 2m+1), columns interleaved
        for j in range(2 * m + 2):
            for h in range(2):
                i = 2 * m + h
                want = L[i, j] if (i < N and j <= i) else 0.0
                assert Lp[2 * m * (m + 1) + 2 * j + h] == want
    Wp = p[n4 + n4 * (n4 // 2 + 1):n4 + n4 * (n4 // 2 + 1) + kpad * n4].reshape(kpad, n4)
    F = p[n4 + n4 * (n4 // 2 + 1) + kpad * n4:]            # fold block of portfolio 0: [c, v_0..v_{n4-1}, pad]
    Lt = np.tril(L).astype(np.float64)
    assert F[0] == np.float32(np.dot(W[0].astype(np.float64), (mu + np.float32(0)).astype(np.float64)))
    np.testing.assert_allclose(F[1:1 + N], (Lt.T @ W[0].astype(np.float64)).astype(np.float32), rtol=2e-7)
    assert np.all(F[1 + N:] == 0)
    assert np.array_equal(Wp[:K, :N], W) and np.all(Wp[:K, N:] == 0) and np.all(Wp[K:] == 0)


def test_pivots_are_the_analytic_mean(mcp_lib):
    """mcp_pivots: c = (1 + w.mu)^T - 1 (simple) / expm1(T (w.mu + w'Sigma w / 2)) (log), binary64 from the binary32 inputs --
    the shift every rank uses for its moment sums."""
    rng = np.random.default_rng(3)
    N, K, T = 5, 4, 252
    mu = rng.normal(4e-4, 2e-4, N).astype(np.float32)
    A = rng.normal(size=(N, N)) * 0.01
    L = np.linalg.cholesky(A @ A.T + 1e-5 * np.eye(N)).astype(np.float32)
    W = rng.dirichlet(np.ones(N), K).astype(np.float32)
    m = W.astype(np.float64) @ mu.astype(np.float64)
    got = _ffi.pivots(_ffi.make_params(N, T, K), mu, L, W)
    np.testing.assert_allclose(got, np.expm1(T * np.log1p(m)), rtol=1e-14)
    np.testing.assert_allclose(got, (1 + m) ** T - 1, rtol=1e-11)
    s2 = np.einsum("ki,ij,kj->k", W.astype(np.float64), np.tril(L).astype(np.float64) @ np.tril(L).astype(np.float64).T, W.astype(np.float64))
    got = _ffi.pivots(_ffi.make_params(N, T, K, "log"), mu, L, W)
    np.testing.assert_allclose(got, np.expm1(T * (m + s2 / 2)), rtol=1e-12)
    assert np.all(_ffi.pivots(_ffi.make_params(N, 0, K), mu, L, W) == 0)
    wild = _ffi.pivots(_ffi.make_params(N, 10**6, 1), np.full(N, 5.0, np.float32), L, W[:1])      # overflows: falls back to 0
    assert wild[0] == 0.0


def test_pack_params_rejects_bad_shapes(mcp_lib):
    assert mcp_lib.mcp_packed_len(0, 1) == 0 and mcp_lib.mcp_packed_len(65, 1) == 0 and mcp_lib.mcp_packed_len(4, 0) == 0
    out = np.zeros(4, np.float32)
    z = np.zeros(16, np.float32)
    rc = mcp_lib.mcp_pack_params(4, 1, z[:4].copy(), z, z[:4].copy(), out, out.size)
    assert rc == -1 and b"too small" in mcp_lib.mcp_last_error()


@pytest.mark.parametrize("n", [1, 2, 13, 20, 21, 1000, 10_000, 1_000_000, 100_000_000, 999_999_937])
@pytest.mark.parametrize("alpha", [0.95, 0.99, 0.5, 0.9])
def test_percentile_rank_matches_numpy(n, alpha):
    """lo/hi/gamma against numpy's own virtual index for q = (1-alpha)*100 (app.py:259)."""
    from numpy.lib import _function_base_impl as fb
    q = np.true_divide((1 - alpha) * 100, 100)
    vi = fb._QuantileMethods["linear"]["get_virtual_index"](n, np.asarray(q))
    lo, hi, g = _ffi.percentile_rank(n, alpha)
    if vi >= n - 1:
        assert lo == hi == n - 1
    else:
        assert lo == int(np.floor(vi)) and hi == lo + 1 and g == float(vi - np.floor(vi))


def test_percentile_rank_reproduces_np_percentile():
    rng = np.random.default_rng(5)
    for alpha in (0.95, 0.99, 0.9, 0.5):
        for n in (1, 2, 13, 21, 1000, 4097, 100_003, 1_000_000):
            for rep in range(3 if n > 10_000 else 25):
                x = rng.normal(size=n).astype(np.float32).astype(np.float64)
                lo, hi, g = _ffi.percentile_rank(n, alpha)
                xs = np.partition(x, [lo, hi])
                a, b = xs[lo], xs[hi]
                d = b - a
                r = a + d * g if g < 0.5 else b - d * (1 - g)
                assert r == np.percentile(x, (1 - alpha) * 100), (alpha, n)


def test_key_transform_is_order_preserving(mcp_lib):
    v = np.array([-np.inf, -3.5, -1e-30, -0.0, 0.0, 1e-38, 0.5, 1.0, 1.0000001, 7e9, np.inf], np.float32)
    keys = [mcp_lib.mcp_float_to_key(float(x)) for x in v]
    assert keys == sorted(keys) and len(set(keys[:3] + keys[5:])) == len(keys[:3] + keys[5:])
    for x, k in zip(v, keys):
        assert np.float32(mcp_lib.mcp_key_to_float(k)).tobytes() == np.float32(x).tobytes()


def test_terminal_to_x(mcp_lib):
    prm = _ffi.make_params(4, 10, 1, "simple", v0=0.1)
    t = np.float32(0.1234)
    assert mcp_lib.mcp_terminal_to_x(ctypes.byref(prm), t) == float(np.float64(t) / np.float64(np.float32(0.1)) - 1.0)
    prm = _ffi.make_params(4, 10, 1, "log")
    assert mcp_lib.mcp_terminal_to_x(ctypes.byref(prm), t) == pytest.approx(float(np.expm1(np.float64(t))), rel=1e-15)


def test_argument_errors_are_reported_not_thrown(mcp_lib):
    prm = _ffi.make_params(4, 10, 1)
    prm.n_assets = 0
    assert mcp_lib.mcp_launch_paths(ctypes.byref(prm), None, None, 0, 0, 100, None, 100, None, None, None) == -1
    assert b"n_assets" in mcp_lib.mcp_last_error()
    prm = _ffi.make_params(4, 10, 1, alpha=0.95)
    assert mcp_lib.mcp_launch_paths(ctypes.byref(prm), None, None, 0, 0, 100, None, 100, None, None, None) == -1
    assert b"NULL device pointer" in mcp_lib.mcp_last_error()
    assert mcp_lib.mcp_launch_sum_u64(None, 2, 16, None) == -1 and mcp_lib.mcp_ctx_exchange_mode(None) == _ffi.EXCHANGE_UNSET
    assert mcp_lib.mcp_ctx_exchange_note(None) == b""
    prm.alpha = 1.5
    assert mcp_lib.mcp_launch_stats(ctypes.byref(prm), 1, None, None, None, None) == -1 and b"alpha" in mcp_lib.mcp_last_error()
    prm.alpha = 0.95
    assert mcp_lib.mcp_launch_scan(ctypes.byref(prm), 2, 10, 0, 1, None, None, None, None, None, None, None) == -1     # pass 2 is mcp_launch_final
    with pytest.raises(ValueError):
        _ffi.make_params(4, 10, 1, compounding="weird")


def test_product_fails_loudly_without_gpu(mcp_lib):
    if mcp_lib.mcp_device_count() > 0:
        pytest.skip("a GPU is visible")
    from monte_carlo_portfolio_amd import simulate_paths
    with pytest.raises(_ffi.McpError, match="no HIP device"):
        simulate_paths(np.zeros(3), np.eye(3) * 1e-4, np.ones(3) / 3, n_paths=8)


def test_not_positive_definite_is_value_error():
    from monte_carlo_portfolio_amd.simulate import cholesky_factor
    with pytest.raises(ValueError, match="positive definite"):
        cholesky_factor(np.ones((3, 3)))


def test_header_is_valid_c99_and_links_from_c(tmp_path, mcp_lib):
    """include/mcport.h must be consumable by a plain C compiler (the boundary is a C ABI), and a C program
    must link against libmcport.so and call the host-only entry points."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text(r'''
        #include <stdio.h>
        #include "mcport.h"
        int main(void) {
            mcp_params p = {4, 10, 1, MCP_COMPOUND_SIMPLE, 0, 0, 1.0, 0.95, 0.0};
            uint64_t lo, hi; double g;
            if (mcp_abi_version() != MCP_ABI_VERSION) return 1;
            if (mcp_percentile_rank(1000000, p.alpha, &lo, &hi, &g) != MCP_OK) return 2;
            if (mcp_packed_len(16, 1) != 16 + 16 * 9 + 8 * 16 + 20) return 3;
            if (mcp_launch_paths(&p, NULL, NULL, 0, 0, 10, NULL, 10, NULL, NULL, NULL) != MCP_E_ARG) return 4;
            if (mcp_ws_bytes(MCP_WS_PARTIALS, 1, 1000000) != 3907 * 32 || mcp_moment_slots(17, 128) != 2) return 5;
            printf("%llu %llu %.17g %s\n", (unsigned long long)lo, (unsigned long long)hi, g, mcp_last_error());
            return 0;
        }''')
    exe = tmp_path / "t"
    inc = os.path.join(ROOT, "include")
    libdir = os.path.dirname(_ffi.LIB_PATH)
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{inc}", str(src), "-o", str(exe),
                        f"-L{libdir}", "-lmcport", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    lo, hi, g = out.stdout.split()[:3]
    assert (int(lo), int(hi)) == (49999, 50000) and float(g) == _ffi.percentile_rank(1_000_000, 0.95)[2]
    assert "NULL device pointer" in out.stdout


def test_multi_device_context_argument_checks(mcp_lib):
    """mcp_ctx_create_multi: argument errors are codes, never crashes; without a GPU every request is MCP_E_NODEVICE."""
    h = ctypes.c_void_p()
    devs = (ctypes.c_int * 2)(0, 1)
    assert mcp_lib.mcp_ctx_create_multi(None, 2, ctypes.byref(h)) == _ffi.MCP_E_ARG
    assert mcp_lib.mcp_ctx_create_multi(devs, 0, ctypes.byref(h)) == _ffi.MCP_E_ARG
    assert mcp_lib.mcp_ctx_create_multi(devs, 2, None) == _ffi.MCP_E_ARG
    assert mcp_lib.mcp_ctx_device_count(None) == 0
    if mcp_lib.mcp_device_count() == 0:
        assert mcp_lib.mcp_ctx_create_multi(devs, 2, ctypes.byref(h)) == _ffi.MCP_E_NODEVICE and not h.value
        from monte_carlo_portfolio_amd import simulate_paths
        with pytest.raises(_ffi.McpError, match="no HIP device"):
            simulate_paths(np.zeros(3), np.eye(3) * 1e-4, np.ones(3) / 3, n_paths=8, devices=[0, 1])


def test_hip_errors_have_their_own_code():
    """A HIP runtime failure is MCP_E_HIP (-5) with the HIP error string, distinct from 'no device' (-2)."""
    text = open(os.path.join(ROOT, "include", "mcport.h")).read()
    assert "MCP_E_HIP = -5" in text and "MCP_E_COMM = -6" in text
    src = open(os.path.join(ROOT, "monte_carlo_portfolio_amd", "csrc", "mcp_api.cpp")).read()
    assert "return fail(MCP_E_HIP, \"%s: %s\", #expr" in src


def test_committed_issue_model_matches_the_kernel_at_head(tmp_path):
    """profiles/issue_model.json (what bench.py prices roofline.issue_model with) must be the instruction mix of the step loop
    the library is built from: regenerate it from the hipcc -S listing (no GPU needed) and compare counts and cycles."""
    import json
    import subprocess
    import sys
    out = tmp_path / "im.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "issue_model.py"), "--clock-hz", "2.303e9", "-o", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    fresh, committed = json.load(open(out)), json.load(open(os.path.join(ROOT, "profiles", "issue_model.json")))
    assert fresh["valu_insts_per_wave_step"] == committed["valu_insts_per_wave_step"] == 404
    assert fresh["cycles_per_wave_step"] == pytest.approx(committed["cycles_per_wave_step"], rel=1e-12)
    assert {(x["inst"], x["count"]) for x in fresh["rows"]} == {(x["inst"], x["count"]) for x in committed["rows"]}
    assert committed["clock_hz"] == pytest.approx(2.303e9) and "r03_clock" in committed["clock_source"]
