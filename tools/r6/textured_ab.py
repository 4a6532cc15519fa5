"""A/B for ADVICE r5 item 4: a textured scene whose materials are metal / plastic / uber -- the lobe-set classes (round 6: textured scenes keep them) against the general
kernels (PT_SHADE_SPECIALISE=0). usage: python tools/r6/textured_ab.py  (prints ms per render and the kernels used)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from _pkg import import_pkg
pkg = import_pkg()
lib = pkg.load_library(); lib.init(0)
b = pkg.scenes.country_kitchen_s3(xres=1920, yres=1080, spp=32, wall_n=140, box_n=24, obj_n=56)
# one checkerboard on a small extra quad: the scene is "textured" (MODE 2 kernels) while its other materials keep constant parameters
import numpy as np
b.texture("chk", "color", "checkerboard", uscale=8.0, vscale=8.0, tex1=(0.8, 0.8, 0.8), tex2=(0.1, 0.1, 0.1))
b.material("matte", Kd="chk")
P, I = pkg.scenes.quad((-0.5, 0.01, 1.0), (-0.5, 0.01, 2.0), (0.5, 0.01, 2.0), (0.5, 0.01, 1.0))
b.trianglemesh(P, I, UV=np.array([[0, 0], [0, 1], [1, 1], [1, 0]], dtype=np.float32))
sd, rp = b.world_end()
sc = pkg.Scene(lib, sd)
for rep in range(3):
    t = time.time(); sc.render(rp); dt = time.time() - t
    ks = {k["name"]: (round(k["total_ms"], 1), k["kernel"]) for k in sc.kernel_stats() if k["name"].startswith("shade")}
    print(f"render {dt * 1e3:.1f} ms", ks, flush=True)
