#!/usr/bin/env python3
"""What makes an eighth of the C2 job slower than an eighth of the time (round 5)? The same number of pixels rendered four ways:
every 8th 16x16 tile (pt_render's `tile % world == rank`), a vertical strip and a horizontal band of an eighth of the frame (crop window, every tile of it), and --
when the library supports grouped sharding (PtRenderParams.tile_group) -- groups of consecutive tiles. One JSON line each with the per-kernel milliseconds."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _pkg import import_pkg
pkg = import_pkg()
import torch
torch.cuda.init()
lib = pkg.load_library(); lib.init(0)
S = pkg.scenes


def run(label, crop=(0.0, 1.0, 0.0, 1.0), rank=0, world=1, group=None):
    b = S.ganesha_scale(n=1466, xres=1920, yres=1080, spp=256)
    b.film["crop"] = crop
    sd, rp = b.world_end()
    rp.tile_rank, rp.tile_world = rank, world
    if group is not None:
        if not hasattr(rp, "tile_group"): return
        rp.tile_group = group
    rp.profile = 1
    scene = pkg.Scene(lib, sd)
    scene.render(rp)
    t0 = time.perf_counter(); scene.render(rp); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    ks = {k["name"]: round(k["total_ms"], 2) for k in scene.kernel_stats() if k["total_ms"] > 0.3 and ":" not in k["name"]}
    print(json.dumps(dict(shape=label, ms=round(dt, 2), camera_rays=scene.counters()["camera_rays"], kernels=ks)), flush=True)
    scene.close() if hasattr(scene, "close") else None


run("every 8th tile (rank 3 of 8)", rank=3, world=8)
run("vertical strip, an eighth of the width (centre)", crop=(0.4375, 0.5625, 0.0, 1.0))
run("horizontal band, an eighth of the height (centre)", crop=(0.0, 1.0, 0.4375, 0.5625))
for g in (4, 8, 15, 30, 120):
    run(f"groups of {g} consecutive tiles (rank 3 of 8)", rank=3, world=8, group=g)
run("whole frame", )
