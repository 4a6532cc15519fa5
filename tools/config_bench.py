#!/usr/bin/env python3
"""Throughput of synthetic analogues of the BASELINE configs other than the headline one (C1, C3, C4, C5) at their named
resolution, with a reduced spp (the per-sample cost does not depend on spp). One JSON line per config:
    python tools/config_bench.py [--spp 64] [--configs C1,C3,C4,C5]
Not part of bench.py's contract (that is the C2 headline); used for DESIGN.md section 7."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from _pkg import import_pkg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--configs", default="C1,C3,C4,C5,TEX,VOL,DISNEY")
    args = ap.parse_args()
    pkg = import_pkg()
    lib = pkg.load_library(); lib.init(0)
    S = pkg.scenes
    builders = {
        "C1": lambda: S.spheres_c1(xres=400, yres=400, spp=args.spp),
        "C3": lambda: S.material_zoo(n=600, xres=1920, yres=1080, spp=args.spp),
        "C4": lambda: S.instanced_garden(n_inst=4000, plant_n=60, xres=1920, yres=1080, spp=args.spp),
        "C5": lambda: S.subsurface_c5(n=500, xres=1920, yres=1080, spp=args.spp),
        "TEX": lambda: S.textured(xres=1920, yres=1080, spp=args.spp),
        "VOL": lambda: S.foggy_room(xres=1920, yres=1080, spp=args.spp),        # volpath: world-filling fog + a medium inside glass
        "DISNEY": lambda: S.disney_spheres(xres=1920, yres=1080, spp=args.spp),
    }
    for name in args.configs.split(","):
        t0 = time.time()
        sd, rp = builders[name]().world_end()
        rp.profile = 1
        scene = pkg.Scene(lib, sd)
        t_setup = time.time() - t0
        scene.render(rp)                      # warm-up (allocations, light grid)
        t0 = time.time(); film = scene.render(rp); dt = time.time() - t0
        c = scene.counters()
        ks = {k["name"]: round(k["total_ms"], 1) for k in scene.kernel_stats() if k["total_ms"] > 0.5}
        n_samples = c["camera_rays"]
        print(json.dumps(dict(config=name, triangles=int(sd.desc().n_triangles), spheres=int(sd.desc().n_spheres), instances=int(sd.desc().n_instances),
                              spp=args.spp, msamples_per_s=round(n_samples / dt / 1e6, 2), ms=round(dt * 1e3, 1), setup_s=round(t_setup, 1),
                              rays_per_sample=round((c["intersect_tests"] + c["shadow_tests"]) / max(1, n_samples), 2),
                              nodes_per_ray=round(c["bvh_nodes_visited"] / max(1, c["intersect_tests"] + c["shadow_tests"]), 1), kernels_ms=ks)), flush=True)
        del scene


if __name__ == "__main__":
    main()
