"""A scene that is larger than 2^25 packets / 4 GB of traversal records BY ITSELF (round 5: the production walk's 31-bit references, k_trace<.., 2> for pools beyond 4 GB):
a 2 n^2-triangle height field (n = 5700: 65 M triangles, ~3.1 GB of packets + ~2.6 GB of four-wide records) is built by the library's host SAH builder, a ray set is
traced closest-hit and any-hit on the GPU, and the oracle -- which ADOPTS the library's tree, so that its own single-threaded build of 65 M primitives is not what the
run waits for -- traces the same rays on the CPU. Hits (primitive, t, barycentrics), occlusion flags and triangle-test counters must be identical.
usage: python tools/big_scene_parity.py [n=5700] [rays=200000]   (prints one JSON line; needs ~25 GB of host memory at n = 5700)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _pkg import import_pkg
pkg = import_pkg()
import torch
torch.cuda.init()
from oracle.oracle_binding import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5700
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
t0 = time.time()
g = np.linspace(-20.0, 20.0, n + 1, dtype=np.float32)
gx, gz = np.meshgrid(g, g, indexing="xy")
gy = (np.sin(gx * np.float32(0.9)) * np.cos(gz * np.float32(1.3)) + np.float32(0.15) * np.sin(gx * np.float32(7.0) + gz * np.float32(5.0))).astype(np.float32)
P = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1).astype(np.float32); del gx, gy, gz
i0 = (np.arange(n, dtype=np.uint32)[None, :] + np.uint32(n + 1) * np.arange(n, dtype=np.uint32)[:, None]).ravel()
I = np.empty((2 * n * n, 3), dtype=np.uint32)
I[0::2, 0] = i0; I[0::2, 1] = i0 + np.uint32(n + 1); I[0::2, 2] = i0 + np.uint32(1)
I[1::2, 0] = i0 + np.uint32(1); I[1::2, 1] = i0 + np.uint32(n + 1); I[1::2, 2] = i0 + np.uint32(n + 2); del i0
b = pkg.scenes.SceneBuilder()
b.film.update(xres=64, yres=48); b.spp = 1
b.look_at((0.0, 30.0, 40.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=40.0)
b.world_begin(); b.light_source("infinite", L=(1.0, 1.0, 1.0)); b.material("matte", Kd=(0.5, 0.5, 0.5)); b.trianglemesh(P, I)
sd, rp = b.world_end()
t_gen = time.time() - t0
lib = pkg.load_library(); lib.init(0)
t0 = time.time(); scene = pkg.Scene(lib, sd); t_build = time.time() - t0
rng = np.random.default_rng(5)
o = np.stack([rng.uniform(-22, 22, n_rays), rng.uniform(1.5, 12.0, n_rays), rng.uniform(-22, 22, n_rays)], axis=1).astype(np.float32)
tgt = np.stack([rng.uniform(-20, 20, n_rays), rng.uniform(-1.0, 1.0, n_rays), rng.uniform(-20, 20, n_rays)], axis=1).astype(np.float32)
d = (tgt - o).astype(np.float32); d[::7] *= np.float32(-1.0)   # (a seventh points at the sky)
tmax = np.full(n_rays, np.inf, np.float32); tm2 = np.full(n_rays, 0.6, np.float32)
t0 = time.time()
gp, gt, gb = scene.trace_closest(o, d, tmax); c1 = scene.counters()
gh = scene.trace_any(o, d, tm2); c2 = scene.counters()
t_gpu = time.time() - t0
ks = [k["kernel"] for k in scene.kernel_stats() if k["launches"] and k["kernel"].startswith("k_trace")]
t0 = time.time()
nodes, ordered = scene.bvh(); sd.set_bvh(nodes, ordered)
orc = Oracle(pkg._abi, pkg.runtime.TABLES_PATH).scene(sd)
t_orc_build = time.time() - t0
t0 = time.time()
op, ot, ob = orc.trace_closest(o, d, tmax); o1 = orc.counters()
oh = orc.trace_any(o, d, tm2); o2 = orc.counters()
t_orc = time.time() - t0
ok = bool(np.array_equal(gp, op) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32)) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32)) and np.array_equal(gh, oh)
          and c1["triangle_tests"] == o1["triangle_tests"] and c2["triangle_tests"] == o2["triangle_tests"])
print(json.dumps(dict(n=n, triangles=int(2 * n * n), bvh_nodes=len(nodes), rays=n_rays, identical=ok, hit_fraction=round(float((op != 0xFFFFFFFF).mean()), 4), occluded_fraction=round(float(oh.mean()), 4),
                      distinct_primitives_hit=int(len(np.unique(op))), max_primitive_hit=int(op[op != 0xFFFFFFFF].max()), triangle_tests=[int(c1["triangle_tests"]), int(c2["triangle_tests"])],
                      trace_kernels=sorted(set(ks)), seconds=dict(scene_gen=round(t_gen, 1), gpu_build_upload=round(t_build, 1), gpu_trace=round(t_gpu, 2), oracle_adopt=round(t_orc_build, 1), oracle_trace=round(t_orc, 1)))))
sys.exit(0 if ok else 1)
