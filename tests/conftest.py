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
NODE_COUNTER = "bvh_nodes_visited"


def trace_exact():
    """The walk the running GPU test instance uses. Kept in the environment, not in a module global: pytest imports this file as a plugin module of its own, the test
    files' `from conftest import ...` may get a second copy, and a global set by the fixture in one copy is not seen through the other (round 5 found `ckeys` never
    including the node counter in the exact instances for that reason)."""
    return os.environ.get("PT_TEST_TRACE_EXACT") == "1"


def ckeys(keys):
    return tuple(k for k in keys if k != NODE_COUNTER or trace_exact())


def trace_env(env=None):
    e = dict(os.environ if env is None else env)
    e["PT_TRACE_EXACT"] = "1" if trace_exact() else "0"
    return e


def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker("gpu") is not None:
        if "trace_mode" not in metafunc.fixturenames:
            metafunc.fixturenames.append("trace_mode")
        metafunc.parametrize("trace_mode", ["quad", "exact"], indirect=True)


# The mode is applied by the run-test hooks, from the instance's parameter: a fixture that the test function does not name is parametrised by the lines above but
# never EXECUTED by recent pytest versions (its name is not in the function's fixture closure), so until round 5 the "exact" instances of every test that did not
# list `trace_mode` among its arguments ran the production walk a second time.
_GPU = []   # the session's library object, once the `gpu` fixture has made it


def _apply_mode(exact):
    os.environ["PT_TEST_TRACE_EXACT"] = "1" if exact else "0"
    for lib in _GPU:
        lib.set_trace_exact(bool(exact))


def pytest_runtest_setup(item):
    cs = getattr(item, "callspec", None)
    _apply_mode(cs is not None and cs.params.get("trace_mode") == "exact")


def pytest_runtest_teardown(item):
    _apply_mode(False)


@pytest.fixture
def trace_mode(request):
    return request.param


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
    lib.set_trace_exact(trace_exact())   # (the first GPU test of the session: its setup hook ran before this object existed)
    _GPU.append(lib)
    return lib
