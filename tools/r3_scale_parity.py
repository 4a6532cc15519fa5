#!/usr/bin/env python3
"""Round-3 features at a size beyond the suite's: GPU vs oracle (exact counters, identical weights, film differences) on media in material-less
shells, subsurface materials under volpath, and the first-touch light grid with 20 000 lights. Run on the GPU box:
    python tools/r3_scale_parity.py > gpurun_out/<tag>/scale_parity.jsonl"""
import json, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from _pkg import import_pkg
pkg = import_pkg()
import torch  # noqa: F401
from oracle.oracle_binding import Oracle, build
build()
lib = pkg.load_library(); lib.init(0)
orc = Oracle(pkg._abi, pkg.runtime.TABLES_PATH)
COUNTERS = ("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests", "path_length_hist", "film_splats",
            "zero_radiance_paths_num", "zero_radiance_paths_den", "sanitized_nan", "sanitized_negative", "sanitized_infinite", "reference_asserts")
def _sss_shell(grid, sampler="sobol"):
    import numpy as np
    b = pkg.scenes.subsurface_in_fog(n=48, xres=960, yres=540, spp=16, sampler=sampler,
                                     fog_density=np.random.default_rng(11).uniform(0.1, 1.0, (4, 3, 5)).astype(np.float32) if grid else None)
    b.attribute_begin(); b.material("none"); b.medium_interface("juice", "fog"); b.translate(0.0, 2.0, 0.0); b.sphere(radius=0.6); b.attribute_end()
    return b


CASES = [
    ("shell_media grid", lambda: pkg.scenes.shell_media(xres=960, yres=540, spp=32, grid=True)),
    ("shell_media homogeneous halton", lambda: pkg.scenes.shell_media(xres=960, yres=540, spp=32, grid=False, sampler="halton")),
    ("subsurface_in_fog", lambda: pkg.scenes.subsurface_in_fog(n=48, xres=960, yres=540, spp=32)),
    ("subsurface next to a shell, fog = grid medium (k_bssrdf stage B)", lambda: _sss_shell(True)),
    ("subsurface next to a shell, homogeneous fog, halton (k_bssrdf stage B)", lambda: _sss_shell(False, "halton")),
    ("emissive_field 20000 lights (first-touch light grid)", lambda: pkg.scenes.emissive_field(n_lights=20000, xres=96, yres=64, spp=4, maxdepth=3)),
]
for name, make in CASES:
    sd, rp = make().world_end()
    t0 = time.time(); g = pkg.Scene(lib, sd); film = g.render(rp); t_gpu = time.time() - t0
    o = orc.scene(sd)
    t0 = time.time(); ref = o.render(rp, nthreads=os.cpu_count()); t_cpu = time.time() - t0
    gc, oc = g.counters(), o.counters()
    bad = [k for k in COUNTERS if gc[k] != oc[k]]
    rel = np.abs(film[..., :3] - ref[..., :3]) / np.maximum(np.abs(ref[..., :3]), 1e-3)
    a, b = g.resolve(film), o.resolve(ref)
    stats = {k["name"]: k["launches"] for k in g.kernel_stats() if k["launches"]}
    print(json.dumps(dict(scene=name, samples=int(gc["camera_rays"]), rays=int(gc["intersect_tests"] + gc["shadow_tests"]), counters_equal=not bad, counters_differing=bad,
                          weights_identical=bool(np.array_equal(film[..., 3], ref[..., 3])), max_rel_diff_film=float(rel.max()), linf_normalised=float(np.abs(a - b).max()),
                          path_length_hist=gc["path_length_hist"], launches=stats, gpu_s=round(t_gpu, 2), oracle_s=round(t_cpu, 1))), flush=True)
