import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The traversal kernels walk four-wide BVH records by default (the production path); `bvh_nodes_visited` then counts records fetched, not the
# reference's node visits. Every GPU test therefore runs twice: "quad" (production: hits, films and every other counter against the oracle) and
# "exact" (pt_set_trace_exact(1): the two-wide walk, whose node counter is compared with the oracle's as well). `ckeys` drops the node counter from a
# list of counter names in production mode; child processes get the mode through PT_TRACE_EXACT (`trace_env`).
TRACE_EXACT = False
NODE_COUNTER = "bvh_nodes_visited"


def ckeys(keys):
    return tuple(k for k in keys if k != NODE_COUNTER or TRACE_EXACT)


def trace_env(env=None):
    e = dict(os.environ if env is None else env)
    e["PT_TRACE_EXACT"] = "1" if TRACE_EXACT else "0"
    return e


def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker("gpu") is not None:
        if "trace_mode" not in metafunc.fixturenames:
            metafunc.fixturenames.append("trace_mode")
        metafunc.parametrize("trace_mode", ["quad", "exact"], indirect=True)


@pytest.fixture
def trace_mode(request, gpu):
    global TRACE_EXACT
    TRACE_EXACT = request.param == "exact"
    gpu.set_trace_exact(TRACE_EXACT)
    yield request.param
    TRACE_EXACT = False
    gpu.set_trace_exact(False)


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
