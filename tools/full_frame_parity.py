#!/usr/bin/env python3
"""Whole-frame parity at BASELINE size (run on the GPU box, whose 256 hardware threads make the oracle fast enough):
the full 1920x1080 frame of a config rendered by the HIP path and by the CPU oracle at a reduced sample count --
exact work counters, identical weights, relative and normalised-L-infinity differences of the films.
The suite's gates (tests/test_gpu_parity.py, tests/test_configs.py) do the same on 256x256 crops.

    python tools/full_frame_parity.py C2:64 C3:16 C4:8 C5:32  > gpurun_out/<tag>/full_frame_parity.jsonl
"""
import json, os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from _pkg import import_pkg
pkg = import_pkg()
import torch  # noqa: F401  (its HIP runtime before the library's)
from oracle.oracle_binding import Oracle, build
build()
lib = pkg.load_library(); lib.init(0)
orc = Oracle(pkg._abi, pkg.runtime.TABLES_PATH)
COUNTERS = ("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests", "path_length_hist", "film_splats",
            "zero_radiance_paths_num", "zero_radiance_paths_den", "sanitized_nan", "sanitized_negative", "sanitized_infinite", "reference_asserts")
for spec in sys.argv[1:]:
    cfg, spp = spec.split(":"); spp = int(spp)
    builder, _, desc = pkg.scenes.CONFIG_SCENES[cfg]
    kw = dict(xres=1920, yres=1080, spp=spp)
    sd, rp = builder(**kw).world_end()
    t0 = time.time(); g = pkg.Scene(lib, sd); film = g.render(rp); t_gpu = time.time() - t0
    nodes, ordered = g.bvh(); sd.set_bvh(nodes, ordered)      # the oracle adopts the library's tree (identical to its own: test_bvh_identical_to_oracle)
    o = orc.scene(sd)
    t0 = time.time(); ref = o.render(rp, nthreads=os.cpu_count()); t_cpu = time.time() - t0
    gc, oc = g.counters(), o.counters()
    bad = [k for k in COUNTERS if gc[k] != oc[k]]
    w_equal = bool(np.array_equal(film[..., 3], ref[..., 3]))
    rel = np.abs(film[..., :3] - ref[..., :3]) / np.maximum(np.abs(ref[..., :3]), 1e-3)
    a, b = g.resolve(film), o.resolve(ref)
    print(json.dumps(dict(config=cfg, workload=desc, spp=spp, samples=int(gc["camera_rays"]), counters_equal=not bad, counters_differing=bad, weights_identical=w_equal,
                          max_rel_diff_film=float(rel.max()), linf_normalised=float(np.abs(a - b).max()), pixels_over_1e_5=int((np.abs(a - b).max(axis=2) > 1e-5).sum()),
                          rays=int(gc["intersect_tests"] + gc["shadow_tests"]), gpu_render_s=round(t_gpu, 2), oracle_render_s=round(t_cpu, 1), oracle_threads=os.cpu_count())), flush=True)
