"""CPU study behind round 5's traversal work (tools/study/orc_study.h): renders a coarse frame of a config with a STUDY build of the oracle
(oracle/*.cpp compiled with -DORC_STUDY into /tmp, never into oracle/liboracle.so) and prints what leaf visits and instance entries look like.
usage: python tools/study/c4_study.py [C4|C2|C3|C5] [xres yres spp]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from _pkg import import_pkg
pkg = import_pkg()
import oracle.oracle_binding as ob

out = "/tmp/study/liboracle_study.so"
os.makedirs("/tmp/study", exist_ok=True)
srcs = [os.path.join(ROOT, "oracle", f) for f in ("ref_render.cpp", "ref_shading.cpp", "ref_sphere.cpp", "ref_kats.cpp", "ref_kats_shapes.cpp")]
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-DORC_STUDY", "-shared", "-o", out] + srcs)
ob.LIB_PATH = out
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
xres, yres, spp = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (96, 54, 4)
builder = pkg.scenes.CONFIG_SCENES[cfg][0]
b = builder(xres=xres, yres=yres, spp=spp)
sd, rp = b.world_end()
orc = ob.Oracle(pkg._abi, pkg.runtime.TABLES_PATH)
sc = orc.scene(sd)
orc.lib.orc_study_reset()
sc.render(rp, nthreads=8)
v = (C.c_uint64 * 64)(); orc.lib.orc_study_read(v); v = list(v)
c = sc.counters()
rays = c["intersect_tests"] + c["shadow_tests"]
print(f"{cfg} {xres}x{yres}x{spp}spp: rays {rays}, nodes/ray {c['bvh_nodes_visited'] / rays:.1f}, tris/ray {c['triangle_tests'] / rays:.2f}")
ent = v[0]
if ent:
    print(f"instance entries {ent} ({ent / rays:.2f} per ray); object root test rejects {v[1]} = {100 * v[1] / ent:.1f} %")
    for name, k in (("box of the instance", 2), ("+ xz diagonals", 4), ("oriented box, bf16 rows", 6)):
        print(f"  gate [{name:24s}] rejects {v[k]} = {100 * v[k] / ent:.1f} % of the entries = {100 * v[k] / max(v[1], 1):.1f} % of the root test's; rejects the root test admits: {v[k + 1]}")
    print(f"  gate [world boxes of the object's four root slots] rejects {v[8]} = {100 * v[8] / ent:.1f} % of the entries ({v[10]} the root test rejects too, {v[9]} it admits -- no leaf reached either way)")
    print("top-level leaf visits by packets:", {n: v[16 + n] for n in range(8) if v[16 + n]}, " by instances in the leaf:", {n: v[40 + n] for n in range(8) if v[40 + n]})
ol = {n: v[24 + n] for n in range(8) if v[24 + n]}
tot = sum(ol.values()); pk = sum(n * k for n, k in ol.items())
print(f"{'object' if ent else 'tree'} leaf visits by packets: {ol}; {pk / max(tot, 1):.2f} packets per visit; visits per ray {tot / rays:.2f}")
print(f"closest-hit leaf visits {v[33]}, with more than one packet hit under the entry t_max: {v[32]} = {100 * v[32] / max(v[33], 1):.2f} %")
