"""Debug helper for tests/test_fuzz_parity.py: python tools/fuzz_debug.py <seed> [key=value ...] (integ keys: maxdepth, rrthreshold, strategy, kind)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _pkg import import_pkg
pkg = import_pkg()
from oracle.oracle_binding import Oracle
import test_fuzz_parity as T
seed = int(sys.argv[1])
b = T.random_scene(pkg, seed)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k == "spp": b.spp = int(v)
    elif k == "sampler": b.sampler = v
    elif k in ("maxdepth",): b.integ[k] = int(v)
    elif k in ("rrthreshold",): b.integ[k] = float(v)
    else: b.integ[k] = v
A = pkg._abi
if "nolens" in os.environ: b.cam.update(lensradius=0.0)
if "boxfilter" in os.environ: b.filter.update(kind="box", radius=(0.5, 0.5))
if "nolights" in os.environ:
    keep = [int(x) for x in os.environ["nolights"].split(",")]
    print("light types", [l.type for l in b.lights])
if "onlylight" in os.environ:
    k = int(os.environ["onlylight"])
    for i, l in enumerate(b.lights):
        if i != k: l.L = (__import__("ctypes").c_float * 3)(0.0, 0.0, 0.0)
    if k != 0 and b.env is not None: b.env["texels"][:] = 0
if "allmatte" in os.environ:
    for m in b.materials: m.type = A.PT_MAT_MATTE
if "noiface" in os.environ:
    b.prim_med_in = [np.full_like(a, A.PT_NONE if b.camera_medium is None else b.camera_medium) for a in b.prim_med_in]
    b.prim_med_out = [np.full_like(a, A.PT_NONE if b.camera_medium is None else b.camera_medium) for a in b.prim_med_out]
if "notex" in os.environ:
    for m in b.materials:
        for k in range(16): m.tex[k] = -1
sd, rp = b.world_end()
lib = pkg.load_library(); lib.init(0)
orc = Oracle(pkg._abi, pkg.runtime.TABLES_PATH)
g = pkg.Scene(lib, sd); o = orc.scene(sd)
film, ref = g.render(rp), o.render(rp, nthreads=8)
gc, oc = g.counters(), o.counters()
print("gpu hist", gc["path_length_hist"][:9]); print("orc hist", oc["path_length_hist"][:9])
for k in ("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests", "film_splats"):
    print(k, gc[k], oc[k], "OK" if gc[k] == oc[k] else "DIFF")
d = np.abs(film[..., :3] - ref[..., :3]).max(axis=2)
rel = d / (np.abs(ref[..., :3]).max(axis=2) + 1e-6)
bad = np.argwhere(rel > 1e-4)
print("pixels off:", len(bad), bad[:10].tolist(), "max rel", float(rel.max()))
