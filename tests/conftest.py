import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    oracle_lib.load()
    return oracle_lib


@pytest.fixture(scope="session")
def product_lib():
    """libraymond_hip.so, built in-tree; building needs hipcc (cross-compiles without a GPU)."""
    from raymond_amd import lib

    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return lib.load()


@pytest.fixture(scope="session")
def gpu_ctx(product_lib):
    from raymond_amd import render

    ctx = render.Context(0)  # raises RMD_ERR_NO_DEVICE without an MI355X: -m gpu tests must fail loudly, not skip
    yield ctx
    ctx.close()
