#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json: Msamples/s at 1920x1080).

A "step" is one full render (pt_render) of the named workload. Default: S2 / config C2, the Ganesha-scale
synthetic scene (4,298,312-triangle displaced sphere, matte, quad area light + constant environment),
1920x1080 x 256 spp, PathIntegrator maxdepth 5, Sobol sampler, box filter, spatial light sampling.
`--config C3|C4|C5` selects the other BASELINE configs built to SURVEY 8(d)'s S3 / S4 / S5 specification
(pbrt-rust_amd/scenes.py: country_kitchen_s3, ecosystem_s4, dragon_s5) at their named spp (override: --spp).
Scene generation, BVH build and upload are outside the timed region. The film stays on the device
(pt_render's film_is_device path): the 33 MB read-back of SURVEY 8(d)'s definition (0.6 ms over PCIe) is not in `value`;
for N > 1 the RCCL film reduction is inside the timed region.

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL). The 16x16 sample tiles of
integrator.rs:276-283 are dealt round-robin to ranks (tile_rank/tile_world in PtRenderParams), every
rank renders all spp of its tiles into a device film, and the films are summed onto rank 0 with one
dist.reduce -- total work is fixed, so scaling is "strong". `--in-process` instead drives all N devices from ONE
process through pt_multi_render (the C ABI's own multi-device path, include/mi355pt.h).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
TRACE_KINDS = ("trace", "extend", "extend_mis", "shadow", "extend_camera", "extend_probe")   # "trace" = the mixed launch: continuation + MIS + shadow rays of one wavefront iteration


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2", choices=["C2", "C3", "C4", "C5"], help="BASELINE config (default: the headline C2)")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per step (default: the config's named spp; C2: 256)")
    ap.add_argument("--mesh-n", type=int, default=1466, help="C2 / C5: displaced-sphere grid (1466 -> 4,298,312 triangles)")
    ap.add_argument("--xres", type=int, default=1920)
    ap.add_argument("--yres", type=int, default=1080)
    ap.add_argument("--spp-per-pass", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample (0 disables)")
    ap.add_argument("--dump-image", default="")
    ap.add_argument("--sim-world", type=int, default=0, help="single-GPU study: render rank 0's shard of an N-rank job (value is then this rank's share only)")
    ap.add_argument("--in-process", action="store_true", help="N > 1 without torchrun: one process drives --gpus devices through pt_multi_render (include/mi355pt.h)")
    ap.add_argument("--devices", default="", help="--in-process: explicit device ordinals, e.g. 0,1,2,3 (an ordinal may repeat: replicas share the device; default 0..gpus-1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    devices = [int(x) for x in args.devices.split(",") if x != ""] or list(range(args.gpus))
    in_process = args.in_process and world == 1 and len(devices) > 1
    if in_process:
        args.gpus = len(set(devices))

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

    builder_fn, named_spp, workload_desc = pkg.scenes.CONFIG_SCENES[args.config]
    spp = args.spp if args.spp > 0 else named_spp
    t_gen = time.time()
    kw = dict(xres=args.xres, yres=args.yres, spp=spp)
    if args.config in ("C2", "C5"):
        kw["n"] = args.mesh_n
    b = builder_fn(**kw)
    sd, rp = b.world_end()
    d = sd.desc()
    n_tris, n_inst = int(d.n_triangles), int(d.n_instances)
    t_gen = time.time() - t_gen
    t_up = time.time()
    scene = pkg.Scene(lib, sd)        # host SAH build + upload + packet build
    t_up = time.time() - t_up
    rp.tile_rank, rp.tile_world = rank, world
    if args.sim_world > 1 and world == 1: rp.tile_rank, rp.tile_world = 0, args.sim_world
    rp.spp_per_pass = args.spp_per_pass
    rp.profile = int(os.environ.get("PT_BENCH_PROFILE", "1"))   # 1: HIP events around every launch, on the render stream; 2: + exact per-class launch sizes
    cb = rp.cropped_pixel_bounds
    W, H = cb[2] - cb[0], cb[3] - cb[1]
    film = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    pb = rp.pixel_bounds
    n_samples = (pb[2] - pb[0]) * (pb[3] - pb[1]) * spp
    n_slots = (-(-(rp.sample_bounds[2] - rp.sample_bounds[0]) // 16)) * (-(-(rp.sample_bounds[3] - rp.sample_bounds[1]) // 16)) * 256 // max(1, world if not in_process else len(devices))
    eff_spp_per_pass = None   # what the library chose (0 = as many samples per pass as the free memory holds, up to 2^28 paths): from the number of k_generate launches below
    multi = None
    if in_process:
        multi = pkg.MultiScene(lib, sd, devices)   # scene replicated on every device, one host thread + stream each

    def step():
        film.zero_()
        torch.cuda.synchronize()
        if multi is not None:
            multi.render(rp, device_ptr=film.data_ptr())          # tiles % n_devices, films summed onto device 0 inside the call
        else:
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
        for ks in (multi.kernel_stats() if multi is not None else scene.kernel_stats()):
            a = kstats.setdefault(ks["name"], dict(launches=0, total_ms=0.0, items=0, bvh_nodes=0, triangle_tests=0, kernel=ks.get("kernel", "")))
            for k in ("launches", "total_ms", "items", "bvh_nodes", "triangle_tests"):
                a[k] += ks[k]
        counters = multi.counters() if multi is not None else scene.counters()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if kstats.get("generate", {}).get("launches"):
        n_pass = max(1, kstats["generate"]["launches"] // args.steps)
        eff_spp_per_pass = -(-spp // n_pass)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    n_gpus = args.gpus if in_process else world
    value = n_samples * args.steps / elapsed / 1e6
    # --- roofline of the dominant kernel: ALGORITHMIC bytes / HIP-event time, next to the HBM bytes the
    # PMC counters saw for the same kernel (profiles/pmc_traffic.json, collected by tools/profile_gpu.sh in separate passes).
    # trace kernels: 32 B per BVH node visited + 48 B per shape packet tested + 44 B per ray (pid 4, ray 24, hit record 16)
    def algo_bytes(name, s):
        if name in TRACE_KINDS:
            return 32 * s["bvh_nodes"] + 48 * s["triangle_tests"] + 44 * s["items"]
        if name.startswith("shade_") or name == "bssrdf":
            return s["bvh_nodes"]  # path-state + mesh + queue bytes counted in-kernel (PtKernelStat.bvh_nodes for shade kernels)
        return None
    # Group the per-launch-kind statistics by kernel symbol (what rocprofv3 --stats reports; PtKernelStat.kernel) and take
    # the symbol with the largest total time as the dominant kernel.
    groups = {}
    for n, v in kstats.items():
        ab = algo_bytes(n, v)
        if ab is None or v["launches"] == 0:
            continue
        g = groups.setdefault(v["kernel"] or n, dict(ms=0.0, launches=0, bytes=0, kinds=[]))
        g["ms"] += v["total_ms"]; g["launches"] += v["launches"]; g["bytes"] += ab; g["kinds"].append(n)
    roofline = None
    if groups:
        name, g = max(groups.items(), key=lambda kv: kv[1]["ms"])
        avg_ms = g["ms"] / max(1, g["launches"])
        achieved = g["bytes"] / (g["ms"] * 1e-3) / 1e9
        traffic = traffic_src = hbm_achieved = hbm_frac = l2_hit = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                data = json.load(open(tpath))
                rec = data.get(args.config, data if args.config == "C2" else {}).get(name)
                if rec and rec.get("spp_per_pass") == eff_spp_per_pass and rec.get("workload") == [args.mesh_n, args.xres, args.yres]:
                    traffic, traffic_src, l2_hit = rec.get("hbm_bytes_per_launch"), rec.get("source"), rec.get("l2_hit_rate")
            except Exception:
                traffic = None
        if traffic:
            hbm_achieved = traffic / (avg_ms * 1e-3) / 1e9
            hbm_frac = hbm_achieved / HBM_PEAK_GBS
        roofline = dict(bound="hbm", limiter="instruction issue (VALU ~80 % of SIMD cycles, as many scalar as vector instructions) with the dependent gathers close behind: the L2 misses are ~70 % of the fabric's measured random-gather ceiling (54 G requests/s, profiles/r2_gather_calibration.json), HBM bytes at hbm_frac (DESIGN.md section 7)", kernel=name, launch_kinds=g["kinds"],
                        achieved=round(achieved, 2), achieved_kind="algorithmic bytes (cache-inclusive) / launch time", peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_src,
                        hbm_achieved=None if hbm_achieved is None else round(hbm_achieved, 2), hbm_frac=None if hbm_frac is None else round(hbm_frac, 5),
                        l2_hit_rate=l2_hit, launches=g["launches"], avg_launch_ms=round(avg_ms, 4),
                        algorithmic_bytes_per_launch=int(g["bytes"] / max(1, g["launches"])))
    def gbs(n, v):
        ab = algo_bytes(n, v)
        return {} if ab is None else {"algo_GBs": round(ab / max(1e-9, v["total_ms"]) / 1e6, 1)}
    kernels = {n: dict(ms=round(v["total_ms"] / args.steps, 3), launches=v["launches"] // args.steps, kernel=v["kernel"], **gbs(n, v),
                       **({"Mitems": round(v["items"] / args.steps / 1e6, 2), "ns_per_item": round(v["total_ms"] * 1e6 / v["items"], 3)} if v["items"] and n not in TRACE_KINDS else {}),
                       **({"Mrays_s": round(v["items"] / max(1e-9, v["total_ms"]) / 1e3, 1), "nodes_per_ray": round(v["bvh_nodes"] / max(1, v["items"]), 1)} if n in TRACE_KINDS else {}))
               for n, v in kstats.items() if v["launches"]}
    # what the mixed traversal launches did per ray kind (counters only: the kinds share the launches' time)
    trace_kinds = {n.split(":", 1)[1]: dict(Mrays_per_step=round(v["items"] / args.steps / 1e6, 2), nodes_per_ray=round(v["bvh_nodes"] / max(1, v["items"]), 1), tris_per_ray=round(v["triangle_tests"] / max(1, v["items"]), 2))
                   for n, v in kstats.items() if n.startswith("trace:")}

    cpu_baseline = None
    if args.cpu_seconds > 0 and world == 1:
        cpu_baseline = run_cpu_baseline(pkg, sd, rp, spp, args)

    if args.dump_image:
        from tools.imgio import write_png
        write_png(args.dump_image, scene.resolve(film.cpu().numpy()))

    out = {
        "metric": "Msamples/s (camera rays x spp) at 1920x1080",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {workload_desc}; {n_tris} triangles" + (f", {n_inst} instances" if n_inst else "") +
                               f", {args.xres}x{args.yres}x{spp}spp, path maxdepth 5, sobol, box filter, spatial light sampling",
                   "name": args.config, "triangles": n_tris, "instances": n_inst, "spp": spp, "spp_per_pass": eff_spp_per_pass, "resolution": [args.xres, args.yres],
                   "film": "stays on the device (no read-back in the timed region; 33 MB = 0.6 ms over PCIe)",
                   "parallelism": (f"16x16 sample tiles round-robin over {n_gpus} GPU(s); " + ("one process, pt_multi_render (peer film sum)" if in_process else "RCCL film reduce")) if n_gpus > 1 else "1 GPU"},
        "roofline": roofline, "cpu_baseline": cpu_baseline,
        "kernels_ms_per_step": kernels, **({"trace_kinds": trace_kinds} if trace_kinds else {}),
        "rays_per_sample": round((counters["intersect_tests"] + counters["shadow_tests"]) / max(1, counters["camera_rays"]), 3) if counters else None,
        "nodes_per_ray": round(counters["bvh_nodes_visited"] / max(1, counters["intersect_tests"] + counters["shadow_tests"]), 2) if counters else None,
        "setup_s": {"scene_gen": round(t_gen, 1), "bvh_build_upload": round(t_up, 1)},
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def run_cpu_baseline(pkg, sd, rp_full, spp, args):
    """The CPU oracle (C++ restatement of the reference path, tile-parallel std::thread like the reference's rayon loop) timed on
    this box's host cores on a bounded sample of the SAME workload: every K-th 16x16 tile of the whole frame (tile_rank /
    tile_world sharding of integrator.rs:276-283's tile list with a stride coprime to the tiles per row), so sky, ground and
    dense geometry enter in the proportion they have in the frame."""
    import copy
    from oracle.oracle_binding import Oracle
    orc = Oracle(pkg._abi, pkg.runtime.TABLES_PATH)
    oscene = orc.scene(sd)
    cores = os.cpu_count() or 1
    sb = rp_full.sample_bounds
    ntx, nty = -(-(sb[2] - sb[0]) // 16), -(-(sb[3] - sb[1]) // 16)
    ntiles = ntx * nty

    def coprime_stride(k):
        from math import gcd
        k = max(1, min(int(k), ntiles))
        while k > 1 and gcd(k, ntx) != 1:
            k += 1
        return k

    cpu_spp = min(spp, 64)   # bounded sample: the per-sample cost does not depend on how many samples a pixel gets (each has its own Sobol' index)

    def render(stride):
        rp = copy.copy(rp_full)
        rp.tile_rank, rp.tile_world, rp.profile, rp.spp = 0, stride, 0, cpu_spp
        oscene.render(rp, nthreads=cores)
        n = oscene.counters()["camera_rays"]
        return n, oscene.seconds()
    # calibrate on ~ one tile per core, then size the sample for ~cpu_seconds
    n, secs = render(coprime_stride(ntiles // max(1, cores)))
    rate = n / max(1e-6, secs)
    want_tiles = max(cores, int(rate * args.cpu_seconds / (256 * cpu_spp)))
    stride = coprime_stride(max(1, ntiles // want_tiles))
    n, secs = render(stride)
    msps = n / secs / 1e6
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return dict(value=round(msps, 4), unit="Msamples/s", cores=cores, kind="port",
                sample=f"every {stride}th 16x16 tile of the whole {args.xres}x{args.yres} frame ({-(-ntiles // stride)} of {ntiles} tiles) at {cpu_spp} spp "
                       f"({n} samples, {secs:.1f} s), oracle = C++ restatement of pbrt-rust's path, one std::thread per core over 16x16 tiles",
                cpu=model)


if __name__ == "__main__":
    main()
