import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mcp_lib():
    """libmcport.so, built in-tree if missing (hipcc cross-compiles gfx950 without a GPU)."""
    from monte_carlo_portfolio_amd import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        _ffi.build()
    return _ffi.lib()


@pytest.fixture(scope="session")
def oracle():
    from oracle import mc_oracle
    mc_oracle.lib()
    return mc_oracle


@pytest.fixture(scope="session")
def gpu_ctx(mcp_lib):
    if mcp_lib.mcp_device_count() < 1:
        pytest.fail("this test is marked gpu but no HIP device is visible")
    from monte_carlo_portfolio_amd.simulate import default_context
    return default_context(0)
