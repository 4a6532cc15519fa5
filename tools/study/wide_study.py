"""CPU study for VERDICT r5 item 2: records / box tests / triangle tests / leaf visits per ray of the production walk (reference tree, two levels per record, reference
order) against SAH-collapsed wide trees with multi-triangle leaves and nearest-first order, on C2-like rays (camera, diffuse bounce, shadow). No GPU.
usage: python tools/study/wide_study.py [mesh_n=733] [n_camera_rays=200000]"""
import ctypes as C, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from _pkg import import_pkg
pkg = import_pkg(); A = pkg._abi
from oracle.oracle_binding import Oracle
out = "/tmp/study/libwide.so"; os.makedirs("/tmp/study", exist_ok=True)
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tools", "study", "wide_study.cpp")])
L = C.CDLL(out)
mesh_n = int(sys.argv[1]) if len(sys.argv) > 1 else 733
n_cam = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
t0 = time.time()
b = pkg.scenes.ganesha_scale(n=mesh_n, xres=1920, yres=1080, spp=1)
sd, rp = b.world_end()
orc = Oracle(A, pkg.runtime.TABLES_PATH); sc = orc.scene(sd)
nodes, ordered = sc.bvh()
print(f"scene: {len(sd.idx)} triangles, {len(nodes)} reference nodes, built in {time.time() - t0:.1f} s")
Pv = np.ascontiguousarray(sd.P, np.float32); Iv = np.ascontiguousarray(sd.idx, np.uint32)
shape = np.asarray(sd.prim_shape, np.uint32)
assert (shape >> 30 == 0).all()
tri_of_prim = (shape & 0x3fffffff).astype(np.uint32)
idx_by_prim = np.ascontiguousarray(Iv.reshape(-1, 3)[tri_of_prim], np.uint32)      # the study indexes triangles by primitive
L.study_set(nodes, ordered.ctypes.data_as(A.u32p), Pv.ctypes.data_as(A.fp), idx_by_prim.ctypes.data_as(A.u32p))
ray_t = np.dtype([("o", np.float32, 3), ("d", np.float32, 3), ("tmax", np.float32), ("any", np.uint32)])
rng = np.random.default_rng(1)
cs = np.concatenate([rng.random((n_cam, 2)) * [1920, 1080], rng.random((n_cam, 3))], axis=1).astype(np.float32)
o = np.zeros((n_cam, 3), np.float32); d = np.zeros((n_cam, 3), np.float32)
assert orc.lib.orc_camera_rays(C.byref(rp), n_cam, cs.ctypes.data_as(A.fp), o.ctypes.data_as(A.fp), d.ctypes.data_as(A.fp)) == 0
cam = np.zeros(n_cam, ray_t); cam["o"] = o; cam["d"] = d; cam["tmax"] = np.inf
L.study_closest.restype = C.c_float
th = np.array([L.study_closest(cam[i:i + 1].ctypes.data_as(C.c_void_p)) for i in range(n_cam)], np.float32)
hit = np.isfinite(th)
ph = o[hit] + d[hit] * th[hit][:, None]
nrm = np.where((ph[:, 1] < -1.15)[:, None], np.array([0, 1, 0], np.float32), ph / np.linalg.norm(ph, axis=1, keepdims=True)).astype(np.float32)
ph = ph + nrm * 1e-3
def cosine_dirs(n):
    u = rng.random((len(n), 2)); r = np.sqrt(u[:, 0]); phi = 2 * np.pi * u[:, 1]
    l = np.stack([r * np.cos(phi), r * np.sin(phi), np.sqrt(1 - u[:, 0])], axis=1)
    a = np.where(np.abs(n[:, :1]) > 0.9, np.array([[0, 1, 0]]), np.array([[1, 0, 0]]))
    t = np.cross(n, a); t /= np.linalg.norm(t, axis=1, keepdims=True); bt = np.cross(n, t)
    return (t * l[:, :1] + bt * l[:, 1:2] + n * l[:, 2:3]).astype(np.float32)
bounce = np.zeros(len(ph), ray_t); bounce["o"] = ph; bounce["d"] = cosine_dirs(nrm); bounce["tmax"] = np.inf
lp = np.stack([rng.random(len(ph)) * 2 - 1, np.full(len(ph), 4.0), rng.random(len(ph)) * 2 - 1], axis=1).astype(np.float32)
shadow = np.zeros(len(ph), ray_t); shadow["o"] = ph; shadow["d"] = lp - ph; shadow["tmax"] = 0.9999; shadow["any"] = 1
class Stats(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("rays", "records", "box_tests", "tri_tests", "leaf_visits", "pushes", "max_stack", "hits", "near_ties")]
def run(which, rays):
    s = Stats(); L.study_walk(which, rays.ctypes.data_as(C.c_void_p), len(rays), C.byref(s)); return s
kinds = (("camera", cam), ("bounce", bounce), ("shadow", shadow))
def report(label, which):
    tot = dict(rays=0, records=0, box=0, tri=0, leaf=0, push=0)
    for name, rays in kinds:
        s = run(which, rays)
        print(f"  {label:26s} {name:7s} records/ray {s.records / s.rays:6.2f}  box tests {s.box_tests / s.rays:6.2f}  tri tests {s.tri_tests / s.rays:5.2f}  leaf visits {s.leaf_visits / s.rays:5.2f}  pushes {s.pushes / s.rays:5.2f}  max stack {int(s.max_stack):3d}  hit {s.hits / s.rays:.3f}  near ties {int(s.near_ties)}")
        w = 1.0 if name != "camera" else 0.5   # a C2 step: ~1 camera : 2 bounce (ext + mis) : 1.3 shadow rays; camera rays are their own launch
        tot["rays"] += s.rays * w; tot["records"] += s.records * w; tot["tri"] += s.tri_tests * w; tot["leaf"] += s.leaf_visits * w
    print(f"  {label:26s} lane-steps per ray (records + triangle tests) {(tot['records'] + tot['tri']) / tot['rays']:.2f}")
print("production walk (reference tree, two levels per record, reference order):"); report("ref4", 0)
for width, leaf_max, ci in ((4, 1, 0.6), (4, 2, 0.6), (4, 4, 0.6), (4, 4, 0.3), (8, 1, 0.6), (8, 4, 0.6), (8, 4, 0.3), (8, 8, 0.3)):
    info = (C.c_double * 4)(); L.study_build_wide(len(nodes), width, leaf_max, C.c_float(ci), info)
    print(f"wide tree width {width}, leaves <= {leaf_max} triangles (ci {ci}): {int(info[0])} nodes (fill {info[2]:.2f}), {int(info[1])} leaves ({info[3]:.2f} triangles each)")
    report(f"wide{width} leaf<={leaf_max} ci{ci}", 1)
