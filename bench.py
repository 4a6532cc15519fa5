#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json: Msamples/s at 1920x1080).

A "step" is one full render (pt_render) of the named workload: S2 / config C2, the Ganesha-scale
synthetic scene (4,298,312-triangle displaced sphere, matte, quad area light + constant environment),
1920x1080 x 256 spp, PathIntegrator maxdepth 5, Sobol sampler, box filter, spatial light sampling.
Scene generation, BVH build and upload are outside the timed region; the film
hand-off (and, for N > 1, the RCCL film reduction) is inside it.

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL). The 16x16 sample tiles of
integrator.rs:276-283 are dealt round-robin to ranks (tile_rank/tile_world in PtRenderParams), every
rank renders all spp of its tiles into a device film, and the films are summed onto rank 0 with one
dist.reduce -- total work is fixed, so scaling is "strong".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel per step (headline config: 256)")
    ap.add_argument("--mesh-n", type=int, default=1466, help="displaced-sphere grid (1466 -> 4,298,312 triangles)")
    ap.add_argument("--xres", type=int, default=1920)
    ap.add_argument("--yres", type=int, default=1080)
    ap.add_argument("--spp-per-pass", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample (0 disables)")
    ap.add_argument("--dump-image", default="")
    ap.add_argument("--sim-world", type=int, default=0, help="single-GPU study: render rank 0's shard of an N-rank job (value is then this rank's share only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import numpy as np
    import torch
    from _pkg import import_pkg
    pkg = import_pkg()
    lib = pkg.load_library()          # raises if libmi355pt.so is missing: there is no CPU fallback
    lib.init(local_rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    t_gen = time.time()
    b = pkg.scenes.ganesha_scale(n=args.mesh_n, xres=args.xres, yres=args.yres, spp=args.spp)
    sd, rp = b.world_end()
    n_tris = int(len(sd.idx))
    t_gen = time.time() - t_gen
    t_up = time.time()
    scene = pkg.Scene(lib, sd)        # host SAH build + upload + packet build
    t_up = time.time() - t_up
    rp.tile_rank, rp.tile_world = rank, world
    if args.sim_world > 1 and world == 1: rp.tile_rank, rp.tile_world = 0, args.sim_world
    rp.spp_per_pass = args.spp_per_pass
    rp.profile = 1                    # HIP events around every launch, on the render stream
    cb = rp.cropped_pixel_bounds
    W, H = cb[2] - cb[0], cb[3] - cb[1]
    film = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    pb = rp.pixel_bounds
    n_samples = (pb[2] - pb[0]) * (pb[3] - pb[1]) * args.spp
    n_slots = (-(-(rp.sample_bounds[2] - rp.sample_bounds[0]) // 16)) * (-(-(rp.sample_bounds[3] - rp.sample_bounds[1]) // 16)) * 256 // world
    eff_spp_per_pass = min(args.spp, args.spp_per_pass if args.spp_per_pass else max(1, (1 << 26) // max(1, n_slots)))

    def step():
        film.zero_()
        torch.cuda.synchronize()
        scene.render(rp, device_ptr=film.data_ptr())
        if dist is not None:
            dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)  # merge_film_tile across ranks (RCCL over xGMI)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    kstats = {}
    counters = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for ks in scene.kernel_stats():
            a = kstats.setdefault(ks["name"], dict(launches=0, total_ms=0.0, items=0, bvh_nodes=0, triangle_tests=0))
            for k in a:
                a[k] += ks[k]
        counters = scene.counters()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = n_samples * args.steps / elapsed / 1e6
    # --- roofline of the dominant kernel: algorithmic bytes / HIP-event time
    # trace kernels: 32 B per BVH node visited + 48 B per triangle packet tested + 44 B per ray (pid 4, ray 24, hit record 16)
    def algo_bytes(name, s):
        if name in ("extend", "extend_mis", "shadow", "extend_camera"):
            return 32 * s["bvh_nodes"] + 48 * s["triangle_tests"] + 44 * s["items"]
        if name.startswith("shade_"):
            return s["bvh_nodes"]  # path-state + queue bytes counted in-kernel (PtKernelStat.bvh_nodes for shade kernels)
        return None
    # Group the per-launch-kind statistics by kernel symbol (what rocprofv3 --stats reports) and take the
    # symbol with the largest total time as the dominant kernel.
    SYMBOL = {"extend_camera": "k_trace<false, 0>", "extend": "k_trace<false, 0>", "extend_mis": "k_trace<false, 0>",
              "shadow": "k_trace<true, 0>", "shade_matte": "k_shade<1, 0, true>", "shade_1lobe": "k_shade<1, 0, false>",
              "shade_2lobe": "k_shade<2, 0, false>", "shade_uber": "k_shade<5, 0, false>", "shade_miss": "k_shade_miss<false>"}  # names as rocprofv3 prints them (no spheres in S2)
    groups = {}
    for n, v in kstats.items():
        ab = algo_bytes(n, v)
        if ab is None:
            continue
        g = groups.setdefault(SYMBOL.get(n, n), dict(ms=0.0, launches=0, bytes=0, kinds=[]))
        g["ms"] += v["total_ms"]; g["launches"] += v["launches"]; g["bytes"] += ab; g["kinds"].append(n)
    roofline = None
    if groups:
        name, g = max(groups.items(), key=lambda kv: kv[1]["ms"])
        achieved = g["bytes"] / (g["ms"] * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath)).get(name)
                if rec and rec.get("spp_per_pass") == eff_spp_per_pass and rec.get("workload") == [args.mesh_n, args.xres, args.yres]:
                    traffic, traffic_src = rec.get("hbm_bytes_per_launch"), rec.get("source")
            except Exception:
                traffic = None
        roofline = dict(bound="hbm", kernel=name, launch_kinds=g["kinds"], achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_src,
                        launches=g["launches"], avg_launch_ms=round(g["ms"] / max(1, g["launches"]), 4),
                        algorithmic_bytes_per_launch=int(g["bytes"] / max(1, g["launches"])))
    def gbs(n, v):
        ab = algo_bytes(n, v)
        return {} if ab is None else {"algo_GBs": round(ab / max(1e-9, v["total_ms"]) / 1e6, 1)}
    kernels = {n: dict(ms=round(v["total_ms"] / args.steps, 3), launches=v["launches"] // args.steps, **gbs(n, v),
                       **({"Mrays_s": round(v["items"] / max(1e-9, v["total_ms"]) / 1e3, 1), "nodes_per_ray": round(v["bvh_nodes"] / max(1, v["items"]), 1)} if n in ("extend", "extend_mis", "shadow", "extend_camera") else {}))
               for n, v in kstats.items()}

    cpu_baseline = None
    if args.cpu_seconds > 0 and world == 1:
        cpu_baseline = run_cpu_baseline(pkg, sd, b, args)

    if args.dump_image:
        from tools.imgio import write_png
        write_png(args.dump_image, scene.resolve(film.cpu().numpy()))

    out = {
        "metric": "Msamples/s (camera rays x spp) at 1920x1080",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"S2 Ganesha-scale: {n_tris}-triangle displaced sphere (matte) + ground + quad area light + constant env, "
                               f"{args.xres}x{args.yres}x{args.spp}spp, path maxdepth 5, sobol, box filter, spatial light sampling",
                   "triangles": n_tris, "spp": args.spp, "spp_per_pass": eff_spp_per_pass, "resolution": [args.xres, args.yres],
                   "parallelism": f"16x16 sample tiles round-robin over {world} GPU(s); RCCL film reduce" if world > 1 else "1 GPU"},
        "roofline": roofline, "cpu_baseline": cpu_baseline,
        "kernels_ms_per_step": kernels,
        "rays_per_sample": round((counters["intersect_tests"] + counters["shadow_tests"]) / max(1, counters["camera_rays"]), 3) if counters else None,
        "setup_s": {"scene_gen": round(t_gen, 1), "bvh_build_upload": round(t_up, 1)},
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def run_cpu_baseline(pkg, sd, builder, args):
    """The CPU oracle (C++ restatement of the reference path, tile-parallel std::thread like the reference's rayon
    loop) timed on this box's host cores on a bounded pixel window of the same workload."""
    import numpy as np
    from oracle.oracle_binding import Oracle
    orc = Oracle(pkg._abi, pkg.runtime.TABLES_PATH)
    oscene = orc.scene(sd)
    cores = os.cpu_count() or 1
    xres, yres = args.xres, args.yres

    def window(wpx, hpx):
        x0, y0 = (xres - wpx) // 2 // 16 * 16, (yres - hpx) // 2 // 16 * 16
        builder.integ["pixelbounds"] = (x0, x0 + wpx, y0, y0 + hpx)
        rp = builder.render_params()
        builder.integ["pixelbounds"] = None
        return rp
    # calibrate on a 64x32 window, then size the sample for ~cpu_seconds
    rp = window(64, 32)
    oscene.render(rp, nthreads=cores)
    rate = 64 * 32 * args.spp / max(1e-6, oscene.seconds())
    target_px = rate * args.cpu_seconds / args.spp
    h = int(max(32, min(yres // 16 * 16, (target_px / 16 * 9) ** 0.5 // 16 * 16)))
    w = int(max(64, min(xres // 16 * 16, (target_px / max(1, h)) // 16 * 16)))
    rp = window(w, h)
    oscene.render(rp, nthreads=cores)
    secs = oscene.seconds()
    msps = w * h * args.spp / secs / 1e6
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return dict(value=round(msps, 4), unit="Msamples/s", cores=cores, kind="port",
                sample=f"centre {w}x{h}-pixel window of the same scene at {args.spp} spp ({w * h * args.spp} samples, {secs:.1f} s), "
                       f"oracle = C++ restatement of pbrt-rust's path, one std::thread per core over 16x16 tiles",
                cpu=model)


if __name__ == "__main__":
    main()
