"""ctypes binding of libmi355pt.so (the HIP product library).  Fails loudly when the
extension is missing: there is NO CPU fallback behind this module."""
import ctypes as C
import os
import subprocess
import numpy as np
from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_LIB_PATH") or os.path.join(_HERE, "csrc", "libmi355pt.so")   # PT_LIB_PATH: kernel-variant experiments
TABLES_PATH = os.path.join(_HERE, "data", "sobol_tables.bin")


def build_library(verbose=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:]); print(r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("building libmi355pt.so failed")
    return LIB_PATH


class PtError(RuntimeError):
    pass


class Library:
    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise PtError(f"{path} not found: build it with __graft_entry__.build() (no CPU fallback exists)")
        self.lib = A.bind(C.CDLL(path), strict=not os.environ.get("PT_LIB_PATH"))   # (PT_LIB_PATH: a kernel-variant library of an A/B run, possibly built from an older tree)

    def check(self, st, what=""):
        if st != A.PT_OK:
            raise PtError(f"{what} failed: status {st}: {self.lib.pt_last_error().decode(errors='replace')}")

    def init(self, device=0):
        self.check(self.lib.pt_init(device), "pt_init")

    def set_trace_exact(self, exact):
        """True: the two-wide BVH walk whose `bvh_nodes_visited` is the reference's counter; False (default): the four-wide production walk
        (same hits, films and other counters). Returns the previous setting."""
        return bool(self.lib.pt_set_trace_exact(1 if exact else 0))


_lib = None


def load_library():
    global _lib
    if _lib is None:
        _lib = Library()
    return _lib


def _fptr(a):
    return a.ctypes.data_as(A.fp)


class Scene:
    """pt_scene handle + the render / parity entry points."""

    def __init__(self, lib, scene_data):
        self.L = lib
        self.data = scene_data
        self.h = C.c_void_p()
        d = scene_data.desc()
        lib.check(lib.lib.pt_scene_create(C.byref(d), C.byref(self.h)), "pt_scene_create")

    def close(self):
        if self.h:
            self.L.lib.pt_scene_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass

    def bvh(self):
        nn, npr = C.c_uint32(), C.c_uint32()
        self.L.check(self.L.lib.pt_scene_bvh_info(self.h, C.byref(nn), C.byref(npr)))
        nodes = (A.PtBVHNode * nn.value)(); ordered = np.zeros(npr.value, dtype=np.uint32)
        self.L.check(self.L.lib.pt_scene_bvh_read(self.h, nodes, ordered.ctypes.data_as(A.u32p)))
        return nodes, ordered

    def pass_size(self, rp):
        """Samples per pixel per wavefront pass pt_render would use for `rp` now (the library's choice from the free memory when rp.spp_per_pass == 0)."""
        s = C.c_uint32()
        self.L.check(self.L.lib.pt_pass_size(self.h, C.byref(rp), C.byref(s)), "pt_pass_size")
        return int(s.value)

    def render(self, rp, film=None, device_ptr=None):
        """Returns the un-normalised film (H, W, 4) = XYZ sums + weight sum."""
        cb = rp.cropped_pixel_bounds
        w, h = cb[2] - cb[0], cb[3] - cb[1]
        if device_ptr is not None:
            st = self.L.lib.pt_render(self.h, C.byref(rp), C.c_void_p(device_ptr), 1)
            self.L.check(st, "pt_render"); return None
        if film is None:
            film = np.zeros((h, w, 4), dtype=np.float32)
        st = self.L.lib.pt_render(self.h, C.byref(rp), film.ctypes.data_as(C.c_void_p), 0)
        self.L.check(st, "pt_render")
        return film

    def resolve(self, film, scale=1.0):
        film = np.ascontiguousarray(film, dtype=np.float32)
        out = np.zeros(film.shape[:-1] + (3,), dtype=np.float32)
        self.L.check(self.L.lib.pt_film_resolve(_fptr(film), film.size // 4, scale, _fptr(out)))
        return out

    def counters(self):
        c = A.PtCounters()
        self.L.check(self.L.lib.pt_get_counters(self.h, C.byref(c)))
        return c.as_dict()

    def kernel_stats(self):
        arr = (A.PtKernelStat * 32)(); n = C.c_uint32()
        self.L.check(self.L.lib.pt_get_kernel_stats(self.h, arr, 32, C.byref(n)))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms, items=arr[i].items, bvh_nodes=arr[i].bvh_nodes, triangle_tests=arr[i].triangle_tests, kernel=arr[i].kernel.decode()) for i in range(n.value)]

    def trace_closest(self, o, d, tmax):
        o, d, tmax = (np.ascontiguousarray(x, dtype=np.float32) for x in (o, d, tmax))
        n = len(tmax)
        prim = np.zeros(n, np.uint32); t = np.zeros(n, np.float32); b = np.zeros((n, 3), np.float32)
        self.L.check(self.L.lib.pt_trace_closest(self.h, n, _fptr(o), _fptr(d), _fptr(tmax), prim.ctypes.data_as(A.u32p), _fptr(t), _fptr(b)))
        return prim, t, b

    def trace_any(self, o, d, tmax):
        o, d, tmax = (np.ascontiguousarray(x, dtype=np.float32) for x in (o, d, tmax))
        n = len(tmax)
        hit = np.zeros(n, np.uint8)
        self.L.check(self.L.lib.pt_trace_any(self.h, n, _fptr(o), _fptr(d), _fptr(tmax), hit.ctypes.data_as(A.u8p)))
        return hit


class MultiScene:
    """pt_multi_scene: the scene replicated on several devices of THIS process (one host thread + stream per replica inside
    pt_multi_render); a device ordinal may repeat. The film comes back summed, on the first device or in a host array."""

    def __init__(self, lib, scene_data, devices):
        self.L = lib; self.data = scene_data; self.devices = list(devices)
        self.h = C.c_void_p()
        d = scene_data.desc()
        devs = (C.c_int * len(self.devices))(*self.devices)
        lib.check(lib.lib.pt_multi_scene_create(C.byref(d), devs, len(self.devices), C.byref(self.h)), "pt_multi_scene_create")

    def close(self):
        if self.h:
            self.L.lib.pt_multi_scene_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass

    def render(self, rp, film=None, device_ptr=None):
        cb = rp.cropped_pixel_bounds
        w, h = cb[2] - cb[0], cb[3] - cb[1]
        if device_ptr is not None:
            self.L.check(self.L.lib.pt_multi_render(self.h, C.byref(rp), C.c_void_p(device_ptr), 1), "pt_multi_render"); return None
        if film is None:
            film = np.zeros((h, w, 4), dtype=np.float32)
        self.L.check(self.L.lib.pt_multi_render(self.h, C.byref(rp), film.ctypes.data_as(C.c_void_p), 0), "pt_multi_render")
        return film

    def counters(self):
        c = A.PtCounters()
        self.L.check(self.L.lib.pt_multi_get_counters(self.h, C.byref(c)))
        return c.as_dict()

    def resolve(self, film, scale=1.0):
        return Scene.resolve(self, film, scale)

    def timing(self):
        """Last render: merge_ms and, per replica, the wall time of its pt_render and of its peer copy."""
        n = len(self.devices)
        merge = C.c_double(); r = (C.c_double * n)(); c = (C.c_double * n)()
        self.L.check(self.L.lib.pt_multi_get_timing(self.h, C.byref(merge), r, c, n))
        return dict(merge_ms=merge.value, render_ms=list(r), copy_ms=list(c))

    def create_timing(self):
        """pt_multi_scene_create: wall time of the call and, per replica, of its own scene creation (replicas 1.. are created concurrently), in ms."""
        n = len(self.devices)
        wall = C.c_double(); r = (C.c_double * n)()
        self.L.check(self.L.lib.pt_multi_get_create_timing(self.h, C.byref(wall), r, n))
        return dict(wall_ms=wall.value, replica_ms=list(r))

    def peer_access(self):
        """Per replica: "same device", "peer access" (device-to-device copies) or "staged through the host" -- how its film reaches the first device."""
        n = len(self.devices)
        p = (C.c_int * n)()
        self.L.check(self.L.lib.pt_multi_get_peer_access(self.h, p, n))
        return [("same device", "peer access", "staged through the host")[v] for v in p]

    def kernel_stats(self, replica=0):
        arr = (A.PtKernelStat * 32)(); n = C.c_uint32()
        self.L.check(self.L.lib.pt_multi_get_kernel_stats(self.h, replica, arr, 32, C.byref(n)))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms, items=arr[i].items, bvh_nodes=arr[i].bvh_nodes, triangle_tests=arr[i].triangle_tests, kernel=arr[i].kernel.decode()) for i in range(n.value)]


def tile_shard(lib, rank, world, replica, n_replicas):
    r, w = C.c_uint32(), C.c_uint32()
    lib.lib.pt_multi_tile_shard(rank, world, replica, n_replicas, C.byref(r), C.byref(w))
    return r.value, w.value
