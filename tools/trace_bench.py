#!/usr/bin/env python3
"""Micro-benchmark of the traversal kernel alone on recorded ray sets of the headline scene:
 (a) coherent camera rays (tile order), (b) incoherent rays leaving the camera hit points in random directions.
Scheduling knobs come from the environment (PT_TRACE_REFILL_MIN / PT_TRACE_LEAF_QUORUM)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _pkg import import_pkg
pkg = import_pkg()
n_mesh = int(os.environ.get("MESH_N", "1466"))
lib = pkg.load_library(); lib.init(0)
b = pkg.scenes.ganesha_scale(n=n_mesh, xres=1920, yres=1080, spp=1)
sd, rp = b.world_end()
scene = pkg.Scene(lib, sd)
A = pkg._abi
import ctypes as C
# camera rays in tile order, 2 per pixel
xs, ys = np.meshgrid(np.arange(1920), np.arange(1080))
tile = (ys // 16) * 120 + xs // 16
order = np.lexsort((xs.ravel() % 16, ys.ravel() % 16, tile.ravel()))
px = np.stack([xs.ravel()[order], ys.ravel()[order]], axis=1).astype(np.float32)
rng = np.random.default_rng(0)
cs = np.concatenate([np.repeat(px, 2, axis=0) + rng.random((2 * len(px), 2), dtype=np.float32), rng.random((2 * len(px), 3), dtype=np.float32)], axis=1).astype(np.float32)
n = len(cs)
o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
assert lib.lib.pt_camera_rays(C.byref(rp), n, cs.ctypes.data_as(A.fp), o.ctypes.data_as(A.fp), d.ctypes.data_as(A.fp)) == 0
tmax = np.full(n, np.inf, np.float32)

def run(name, o, d, tmax, any_hit=False, reps=3):
    best = None
    for _ in range(reps):
        if any_hit: scene.trace_any(o, d, tmax)
        else: res = scene.trace_closest(o, d, tmax)
        ks = scene.kernel_stats()[0]; c = scene.counters()
        if best is None or ks["total_ms"] < best[0]:
            best = (ks["total_ms"], c)
    ms, c = best
    ab = 32 * c["bvh_nodes_visited"] + 48 * c["triangle_tests"] + 44 * len(tmax)
    print(json.dumps(dict(case=name, rays=len(tmax), ms=round(ms, 3), mrays_s=round(len(tmax) / ms / 1e3, 1), nodes_per_ray=round(c["bvh_nodes_visited"] / len(tmax), 1),
                          tris_per_ray=round(c["triangle_tests"] / len(tmax), 2), algo_GBs=round(ab / ms / 1e6, 1))))
    return None if any_hit else res

prim, t, bb = run("camera_closest", o, d, tmax)
hit = prim != 0xFFFFFFFF
p = o[hit] + d[hit] * t[hit][:, None]
v = rng.normal(size=(len(p), 3)).astype(np.float32); v /= np.linalg.norm(v, axis=1, keepdims=True)
p2 = (p + 1e-3 * v).astype(np.float32)
run("bounce_closest", p2, v, np.full(len(p2), np.inf, np.float32))
run("bounce_any", p2, v, np.full(len(p2), 3.0, np.float32), any_hit=True)
perm = rng.permutation(len(p2))
run("bounce_closest_shuffled", p2[perm], v[perm], np.full(len(p2), np.inf, np.float32))
# Bound study: (1) every ray identical -> all 64 lanes of a step fetch ONE record: the issue/ALU ceiling of the loop;
# (2) groups of 64 identical rays -> each wave step touches one record but waves differ: adds the cache-capacity effect
# without per-lane address divergence; (3) groups of 8.
m = int(os.environ.get('STUDY_RAYS', '16000000'))   # large enough that ramp-up and drain of the persistent waves do not matter
sel = rng.integers(0, len(p2), size=m)
inf = np.full(m, np.inf, np.float32)
one = np.repeat(sel[:1], m)
run("identical_rays", p2[one], v[one], inf)
g64 = np.repeat(sel[: m // 64], 64)[:m]
run("groups_of_64", p2[g64], v[g64], inf[: len(g64)])
g8 = np.repeat(sel[: m // 8], 8)[:m]
run("groups_of_8", p2[g8], v[g8], inf[: len(g8)])
run("independent_same_count", p2[sel], v[sel], inf)
# (4) K distinct rays tiled over the launch: every wave step has 64 different addresses (full per-lane divergence) but the
# footprint of the whole launch is K rays' worth of records -- separates the address-processing cost from cache misses.
for K in (64, 4096, 262144):
    t = np.tile(sel[:K], m // K)
    run(f"tiled_{K}_distinct", p2[t], v[t], inf[: len(t)])
