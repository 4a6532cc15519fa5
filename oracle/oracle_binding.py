"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SAN = bool(os.environ.get("PT_SAN"))   # ASan/UBSan build of the oracle (tools/san_cpu_tests.sh preloads the sanitizer runtimes)
LIB_PATH = os.path.join(_HERE, "liboracle_san.so" if SAN else "liboracle.so")


def build(verbose=False):
    r = subprocess.run(["make", "-C", _HERE] + (["SAN=1"] if SAN else []), capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-3000:]); print(r.stderr[-3000:])
    if r.returncode != 0:
        raise RuntimeError("building liboracle.so failed")
    return LIB_PATH


class Oracle:
    def __init__(self, abi, tables_path):
        if not os.path.exists(LIB_PATH):
            build()
        self.A = abi
        A = abi
        lib = C.CDLL(LIB_PATH)
        table = {k.replace("pt_", "orc_", 1): v for k, v in A.ENTRY_POINTS.items()
                 if k in ("pt_scene_create", "pt_scene_destroy", "pt_scene_bvh_info", "pt_scene_bvh_read", "pt_get_counters",
                          "pt_trace_closest", "pt_trace_any", "pt_sobol_samples", "pt_halton_samples", "pt_camera_rays")}
        for name, (res, args) in table.items():
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
        lib.orc_render.restype = C.c_int
        lib.orc_render.argtypes = [C.c_void_p, C.POINTER(A.PtRenderParams), A.fp, C.c_int]
        lib.orc_film_resolve.argtypes = [A.fp, C.c_uint32, C.c_float, A.fp]
        lib.orc_last_render_seconds.restype = C.c_double; lib.orc_last_render_seconds.argtypes = [C.c_void_p]
        lib.orc_load_tables.argtypes = [C.c_char_p]
        f = C.c_float
        lib.orc_sobol_sample_float.restype = f; lib.orc_sobol_sample_float.argtypes = [C.c_uint64, C.c_int, C.c_uint32]
        lib.orc_radical_inverse.restype = f; lib.orc_radical_inverse.argtypes = [C.c_int, C.c_uint64]
        lib.orc_radical_inverse_any.restype = f; lib.orc_radical_inverse_any.argtypes = [C.c_uint32, C.c_uint64]
        lib.orc_halton_permutation.restype = C.c_uint32; lib.orc_halton_permutation.argtypes = [C.c_uint32, C.POINTER(C.c_uint16)]
        for n in ("orc_next_float_up", "orc_next_float_down", "orc_dm_sin", "orc_dm_cos", "orc_dm_acos", "orc_dm_log"):
            getattr(lib, n).restype = f; getattr(lib, n).argtypes = [f]
        lib.orc_dm_atan2.restype = f; lib.orc_dm_atan2.argtypes = [f, f]
        lib.orc_find_interval.argtypes = [C.c_int, A.fp, f]
        lib.orc_rng_u32_stream.argtypes = [C.c_uint64, C.c_int, C.c_uint32, A.u32p, A.fp]
        lib.orc_dist1d_sample_discrete.argtypes = [A.fp, C.c_int, f, A.fp, A.fp]
        lib.orc_dist1d_discrete_pdf.restype = f; lib.orc_dist1d_discrete_pdf.argtypes = [A.fp, C.c_int, C.c_int]
        lib.orc_dist1d_sample_continuous.restype = f; lib.orc_dist1d_sample_continuous.argtypes = [A.fp, C.c_int, f, A.fp, C.POINTER(C.c_int)]
        lib.orc_tri_intersect.argtypes = [C.c_void_p, C.c_uint32, A.fp, A.fp, f, A.fp, A.fp, A.fp, A.fp, A.fp]
        lib.orc_tri_intersect_p.argtypes = [C.c_void_p, C.c_uint32, A.fp, A.fp, f]
        lib.orc_offset_ray_origin.argtypes = [A.fp] * 5
        tp = C.POINTER(A.PtBSSRDFTable)
        lib.orc_bssrdf_sr.argtypes = [tp, A.fp, A.fp, f, C.c_uint32, A.fp, A.fp, A.fp]
        lib.orc_bssrdf_sample_sr.argtypes = [tp, A.fp, A.fp, f, C.c_int, C.c_uint32, A.fp, A.fp]
        lib.orc_catmull_rom_weights.argtypes = [C.c_int, A.fp, f, C.POINTER(C.c_int), A.fp]
        lib.orc_bssrdf_sw.restype = f; lib.orc_bssrdf_sw.argtypes = [f, f]
        lib.orc_light_sample_li.argtypes = [C.c_void_p, C.c_uint32, A.fp, A.fp, A.fp, C.c_uint32, A.fp, A.fp, A.fp, A.fp]
        lib.orc_light_pdf_li.argtypes = [C.c_void_p, C.c_uint32, A.fp, A.fp, A.fp, C.c_uint32, A.fp, A.fp]
        if lib.orc_load_tables(tables_path.encode()) != 0:
            raise RuntimeError("oracle: cannot load " + tables_path)
        self.lib = lib

    def scene(self, scene_data):
        return OracleScene(self, scene_data)


def _fp(A, a):
    return a.ctypes.data_as(A.fp)


class OracleScene:
    def __init__(self, orc, scene_data):
        self.O = orc; self.A = orc.A; self.data = scene_data
        self.h = C.c_void_p()
        d = scene_data.desc()
        st = orc.lib.orc_scene_create(C.byref(d), C.byref(self.h))
        assert st == 0, st

    def close(self):
        if self.h:
            self.O.lib.orc_scene_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass

    def bvh(self):
        A = self.A
        nn, npr = C.c_uint32(), C.c_uint32()
        self.O.lib.orc_scene_bvh_info(self.h, C.byref(nn), C.byref(npr))
        nodes = (A.PtBVHNode * nn.value)(); ordered = np.zeros(npr.value, dtype=np.uint32)
        self.O.lib.orc_scene_bvh_read(self.h, nodes, ordered.ctypes.data_as(A.u32p))
        return nodes, ordered

    def render(self, rp, nthreads=1):
        cb = rp.cropped_pixel_bounds
        w, h = cb[2] - cb[0], cb[3] - cb[1]
        film = np.zeros((h, w, 4), dtype=np.float32)
        st = self.O.lib.orc_render(self.h, C.byref(rp), _fp(self.A, film), nthreads)
        assert st == 0, st
        return film

    def seconds(self):
        return self.O.lib.orc_last_render_seconds(self.h)

    def resolve(self, film, scale=1.0):
        out = np.zeros(film.shape[:-1] + (3,), dtype=np.float32)
        self.O.lib.orc_film_resolve(_fp(self.A, film), film.size // 4, scale, _fp(self.A, out))
        return out

    def counters(self):
        c = self.A.PtCounters()
        self.O.lib.orc_get_counters(self.h, C.byref(c))
        return c.as_dict()

    def trace_closest(self, o, d, tmax):
        A = self.A
        o, d, tmax = (np.ascontiguousarray(x, dtype=np.float32) for x in (o, d, tmax))
        n = len(tmax)
        prim = np.zeros(n, np.uint32); t = np.zeros(n, np.float32); b = np.zeros((n, 3), np.float32)
        self.O.lib.orc_trace_closest(self.h, n, _fp(A, o), _fp(A, d), _fp(A, tmax), prim.ctypes.data_as(A.u32p), _fp(A, t), _fp(A, b))
        return prim, t, b

    def trace_any(self, o, d, tmax):
        A = self.A
        o, d, tmax = (np.ascontiguousarray(x, dtype=np.float32) for x in (o, d, tmax))
        n = len(tmax)
        hit = np.zeros(n, np.uint8)
        self.O.lib.orc_trace_any(self.h, n, _fp(A, o), _fp(A, d), _fp(A, tmax), hit.ctypes.data_as(A.u8p))
        return hit
