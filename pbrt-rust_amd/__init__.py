"""pbrt-rust_amd: MI355X-native wavefront path-tracing back end for pbrt-rust's render loop.

Python here is harness only (scene assembly mirror + ctypes binding over the C ABI in
include/mi355pt.h).  The product is csrc/ -> libmi355pt.so (hand-written HIP for gfx950).
The directory name contains a '-', so import it through `import_pkg()` in the repo-root
`_pkg.py` (module name `pbrt_rust_amd`).
"""
from . import _abi, host, scenes, frontend, textures, bssrdf  # noqa: F401
from . import runtime  # noqa: F401
from .runtime import Library, Scene, MultiScene, load_library, build_library  # noqa: F401
