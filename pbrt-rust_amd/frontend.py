"""ctypes binding of libmi355front.so, the compiled .pbrt scene-file front end (include/mi355front.h)."""
import ctypes as C
import os
import subprocess
from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
SAN = bool(os.environ.get("PT_SAN"))   # ASan/UBSan build of the front end (tools/san_cpu_tests.sh)
LIB_PATH = os.path.join(_HERE, "frontend", "libmi355front_san.so" if SAN else "libmi355front.so")
CLI_PATH = os.path.join(_HERE, "frontend", "mi355pbrt")


def build(verbose=False):
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "frontend")] + (["SAN=1"] if SAN else []), capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-3000:]); print(r.stderr[-3000:])
    if r.returncode != 0:
        raise RuntimeError("building libmi355front.so failed")
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.ptf_parse_file.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.ptf_parse_string.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.ptf_last_error.restype = C.c_char_p
        L.ptf_scene_desc.restype = C.POINTER(A.PtSceneDesc); L.ptf_scene_desc.argtypes = [C.c_void_p]
        L.ptf_render_params.restype = C.POINTER(A.PtRenderParams); L.ptf_render_params.argtypes = [C.c_void_p]
        L.ptf_output_filename.restype = C.c_char_p; L.ptf_output_filename.argtypes = [C.c_void_p]
        L.ptf_scene_destroy.argtypes = [C.c_void_p]
        L.ptf_write_pfm.argtypes = [C.c_char_p, C.c_int, C.c_int, A.fp]
        L.ptf_write_image.argtypes = [C.c_char_p, C.c_int, C.c_int, A.fp]
        L.ptf_read_image.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), A.fp, C.c_size_t]
        _lib = L
    return _lib


class FrontScene:
    """A parsed .pbrt scene; quacks like host.SceneData (desc()) so that runtime.Scene / the oracle binding accept it."""

    def __init__(self, path=None, text=None, base_dir=None):
        L = lib()
        self.h = C.c_void_p()
        if path is not None:
            st = L.ptf_parse_file(os.fsencode(path), C.byref(self.h))
        else:
            st = L.ptf_parse_string(text.encode(), os.fsencode(base_dir) if base_dir else None, C.byref(self.h))
        if st != A.PT_OK:
            raise ValueError(L.ptf_last_error().decode(errors="replace"))

    def desc(self):
        d = lib().ptf_scene_desc(self.h).contents
        d._owner = self   # the struct points into the C++ scene: keep it alive as long as the view is
        return d

    def render_params(self):
        rp = A.PtRenderParams()
        C.memmove(C.byref(rp), lib().ptf_render_params(self.h), C.sizeof(A.PtRenderParams))
        return rp

    def output_filename(self):
        return lib().ptf_output_filename(self.h).decode()

    def close(self):
        if self.h:
            lib().ptf_scene_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass


def read_image(path):
    """read_image (core/imageio.rs:18-40) through the compiled front end: (h, w, 3) float32, top row first."""
    import numpy as np
    L = lib(); w, h = C.c_int(), C.c_int()
    if L.ptf_read_image(os.fsencode(path), C.byref(w), C.byref(h), None, 0) != 0: raise ValueError(L.ptf_last_error().decode(errors="replace"))
    out = np.zeros((h.value, w.value, 3), np.float32)
    if L.ptf_read_image(os.fsencode(path), None, None, out.ctypes.data_as(A.fp), out.size) != 0: raise ValueError(L.ptf_last_error().decode(errors="replace"))
    return out


def write_image(path, rgb):
    """write_image (core/imageio.rs:42-60): rgb = (h, w, 3) float32, top row first; format by extension (exr, png, tga, pfm)."""
    import numpy as np
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    L = lib()
    if L.ptf_write_image(os.fsencode(path), rgb.shape[1], rgb.shape[0], rgb.ctypes.data_as(A.fp)) != 0: raise ValueError(L.ptf_last_error().decode(errors="replace"))
