"""Import helper: the package directory is `pbrt-rust_amd/` (not a valid identifier)."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def import_pkg():
    name = "pbrt_rust_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(_ROOT, "pbrt-rust_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(path, "__init__.py"), submodule_search_locations=[path])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
