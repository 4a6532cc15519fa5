import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from _pkg import import_pkg
pkg = import_pkg()
from oracle.oracle_binding import Oracle
lib = pkg.load_library(); lib.init(0)
orc = Oracle(pkg._abi, pkg.runtime.TABLES_PATH)
S = pkg.scenes
def run(name, mod):
    b = S.foggy_room(xres=48, yres=32, spp=4)
    mod(b)
    sd, rp = b.world_end()
    g = pkg.Scene(lib, sd); o = orc.scene(sd)
    film, ref = g.render(rp), o.render(rp, nthreads=8)
    gc, oc = g.counters(), o.counters()
    d = np.abs(film[..., :3] - ref[..., :3]).max(axis=2); rel = d / (np.abs(ref[..., :3]).max(axis=2) + 1e-6)
    print(name, "hist", "OK" if gc["path_length_hist"] == oc["path_length_hist"] else "DIFF", "isect", "OK" if gc["intersect_tests"] == oc["intersect_tests"] else "DIFF", "pixels off", int((rel > 1e-4).sum()))
run("base", lambda b: None)
run("image env", lambda b: b.light_source("infinite", L=(0.5, 0.5, 0.5), texels=S.sky_env(16, 8), scale=0.7))
run("spot", lambda b: b.light_source("spot", from_=(2.0, 3.0, 1.0), to=(0.0, 0.0, 0.0), I=(30, 30, 30), coneangle=30.0, conedeltaangle=5.0))
run("distant", lambda b: b.light_source("distant", from_=(2.0, 4.0, 1.0), to=(0.0, 0.0, 0.0), L=(1.0, 1.0, 1.0)))
def sph_light(b):
    b.attribute_begin(); b.area_light_source(L=(20, 20, 20)); b.translate(-1.0, 2.5, 0.5); b.sphere(radius=0.3); b.attribute_end()
run("sphere light", sph_light)
def partial(b):
    b.attribute_begin(); b.material("matte"); b.translate(-0.3, 0.3, 1.5); b.rotate(40, 1, 1, 0); b.sphere(radius=0.4, zmin=-0.2, zmax=0.3, phimax=250.0); b.attribute_end()
run("partial sphere", partial)
def rev(b):
    b.attribute_begin(); b.reverse_orientation = True; b.medium_interface("juice", "fog"); b.material("glass"); b.translate(-0.3, 0.3, 1.5); b.sphere(radius=0.4); b.attribute_end()
run("reversed glass with interface", rev)
run("uniform strategy", lambda b: b.integ.update(strategy="uniform"))
run("hlbvh", lambda b: setattr(b, "split_method", "hlbvh"))
run("gaussian", lambda b: b.filter.update(kind="gaussian", radius=(1.1, 0.6)))
def substrate(b):
    b.attribute_begin(); b.material("substrate"); b.translate(-0.3, 0.3, 1.5); b.sphere(radius=0.4); b.attribute_end()
run("substrate", substrate)
