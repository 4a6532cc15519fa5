import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    from _pkg import import_pkg
    return import_pkg()


@pytest.fixture(scope="session")
def oracle(pkg):
    """CPU oracle (test infrastructure only)."""
    from oracle.oracle_binding import Oracle, build
    build()
    return Oracle(pkg._abi, pkg.runtime.TABLES_PATH)


@pytest.fixture(scope="session")
def gpu(pkg):
    """The HIP product library on device 0. No fallback: a missing .so or device is a hard failure."""
    try:
        import torch  # noqa: F401  -- tests that also use torch need ITS bundled HIP/HSA runtime loaded before libmi355pt.so pulls in the system one
    except ImportError:
        pass
    lib = pkg.load_library()
    lib.init(0)
    return lib
