#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz: small end-to-end fixtures (inputs = a named scene of pbrt_rust_amd.scenes at a fixed
size; outputs = the film, the work counters, Sobol' samples and camera rays) produced by the CPU ORACLE.

The reference itself cannot be run in this environment (no Rust toolchain), so these vectors pin the oracle against
drift and give the GPU tests a committed target that does not need the oracle at run time; they are NOT reference
outputs ("parity unpinned" end to end).  Usage: python tools/make_golden.py"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from _pkg import import_pkg

CASES = {   # name -> (scene function, kwargs); kept tiny: each film is 40x24x4 float32
    "ganesha_small": ("ganesha_scale", dict(n=16, xres=40, yres=24, spp=4)),
    "ganesha_normals_uniform": ("ganesha_scale", dict(n=12, xres=40, yres=24, spp=4, with_normals=True, strategy="uniform")),
    "material_zoo": ("material_zoo", dict(n=10, xres=40, yres=24, spp=4)),
    "spheres_c1": ("spheres_c1", dict(xres=32, yres=32, spp=4)),
    "instanced_garden": ("instanced_garden", dict(xres=40, yres=24, spp=4)),
    "subsurface_c5": ("subsurface_c5", dict(n=10, xres=40, yres=24, spp=4)),
    "sphere_lights": ("sphere_lights", dict(xres=40, yres=24, spp=4)),
    "textured_bump_noise": ("textured", dict(xres=40, yres=24, spp=4, bump=True, noise=True)),
    "alpha_foliage": ("alpha_foliage", dict(xres=40, yres=24, spp=4)),
    "translucent_panels": ("translucent_panels", dict(xres=40, yres=24, spp=4)),
    "mix_materials_textured": ("mix_materials", dict(xres=40, yres=24, spp=4, textured=True)),
    "disney_spheres": ("disney_spheres", dict(xres=40, yres=24, spp=4)),
    "foggy_room_volpath": ("foggy_room", dict(xres=40, yres=24, spp=4)),
    "ganesha_halton_hlbvh": ("ganesha_halton_hlbvh", dict(n=16, xres=40, yres=24, spp=4)),
    # round 2
    "smoke_room_grid_medium": ("smoke_room", dict(xres=40, yres=24, spp=4, sampler="halton")),
    "subsurface_sheets_long_chains": ("subsurface_sheets", dict(xres=40, yres=24, spp=4)),
    "country_kitchen_s3_mini": ("country_kitchen_s3", dict(xres=40, yres=24, spp=4, wall_n=10, box_n=4, obj_n=8)),
    "ecosystem_s4_mini": ("ecosystem_s4", dict(xres=40, yres=24, spp=4, n_inst=40, terrain_n=16, plant_scale=0.07, env_size=(32, 16))),
    "dragon_s5_mini": ("dragon_s5", dict(xres=40, yres=24, spp=4, n=24, env_size=(32, 16))),
}
COUNTERS = ("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests",
            "zero_radiance_paths_num", "zero_radiance_paths_den", "path_length_hist", "film_splats")


def main():
    pkg = import_pkg()
    from oracle.oracle_binding import Oracle, build
    build()
    orc = Oracle(pkg._abi, pkg.runtime.TABLES_PATH)
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    only = set(sys.argv[1:])     # optional: the names to (re)generate
    for name, (fn, kw) in CASES.items():
        if only and name not in only: continue
        sd, rp = getattr(pkg.scenes, fn)(**kw).world_end()
        s = orc.scene(sd)
        film = s.render(rp, nthreads=1)
        c = s.counters()
        np.savez_compressed(os.path.join(out, name + ".npz"), film=film.astype(np.float32),
                            meta=np.frombuffer(json.dumps(dict(scene=fn, kwargs=kw, counters={k: c[k] for k in COUNTERS})).encode(), dtype=np.uint8))
        print(name, film.shape, float(film[..., :3].mean()))


if __name__ == "__main__":
    main()
