"""One process, several devices (include/mi355pt.h: pt_multi_*).

CPU: the tile-shard arithmetic behind pt_multi_render is a pure host function of the library (no GPU needed to call it): replicas
partition the caller's own shard exactly, and nest inside a multi-process launch. GPU: three replicas of one scene on the single
device of the test box (ordinals may repeat) render through one host thread + stream each, the films are merged on the device --
the result must equal the plain single-scene render (weights bit for bit, radiance to float summation order) with the counters
summing to the same totals."""
import ctypes as C
import json
import numpy as np
import pytest
from conftest import ckeys


def test_replica_tile_shards_partition_the_callers_shard(pkg):
    lib = pkg.load_library()      # dlopen only: pt_multi_tile_shard is host arithmetic
    ntiles = 8160                 # 1920x1080 in 16x16 tiles
    tiles = np.arange(ntiles)
    for world in (1, 2, 8):
        for rank in range(world):
            mine = tiles[tiles % world == rank]
            for n in (1, 2, 3, 8):
                owned = []
                for i in range(n):
                    r, w = pkg.runtime.tile_shard(lib, rank, world, i, n)
                    assert w == world * n and r < w
                    owned.append(tiles[tiles % w == r])
                allo = np.concatenate(owned)
                assert len(allo) == len(mine) and np.array_equal(np.sort(allo), mine)          # exact partition of the caller's shard
                assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1            # balanced to one tile


def test_library_reports_no_devices_without_a_gpu_instead_of_crashing(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box")
    lib = pkg.load_library()
    n = C.c_int(-1)
    assert lib.lib.pt_device_count(C.byref(n)) == 0 and n.value == 0


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["ganesha", "garden"])
def test_three_replicas_on_one_device_equal_the_plain_render(pkg, gpu, scene):
    if scene == "ganesha":
        sd, rp = pkg.scenes.ganesha_scale(n=48, xres=160, yres=96, spp=8).world_end()
    else:
        sd, rp = pkg.scenes.instanced_garden(xres=128, yres=80, spp=8).world_end()
    single = pkg.Scene(gpu, sd)
    ref = single.render(rp)
    rc = single.counters()
    multi = pkg.MultiScene(gpu, sd, [0, 0, 0])
    film = multi.render(rp)
    mc = multi.counters()
    for k in ckeys(("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "path_length_hist", "film_splats")):
        assert mc[k] == rc[k], (k, mc[k], rc[k])
    assert np.array_equal(film[..., 3], ref[..., 3])
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)
    # the device-pointer form adds into a film on the first device
    import torch
    dfilm = torch.zeros(film.shape, dtype=torch.float32, device="cuda:0")
    multi.render(rp, device_ptr=dfilm.data_ptr())
    torch.cuda.synchronize()
    # (weights bit for bit; radiance within the reordering of float atomics: a Sobol' sample that sits exactly on a pixel corner -- every
    #  pixel's sample 0 -- splats onto four pixels, three of them through atomics whose order against the owner's additions is not defined)
    assert np.array_equal(dfilm.cpu().numpy()[..., 3], film[..., 3])
    np.testing.assert_allclose(dfilm.cpu().numpy()[..., :3], film[..., :3], rtol=2e-6, atol=1e-7)
    # nested inside a 2-process launch: rank 1 of 2, split over the three replicas
    rp.tile_rank, rp.tile_world = 1, 2
    a = multi.render(rp); b = single.render(rp)
    assert np.array_equal(a[..., 3], b[..., 3])
    np.testing.assert_allclose(a[..., :3], b[..., :3], rtol=2e-6, atol=1e-7)
    assert len(multi.kernel_stats(2)) > 0
    tm = multi.timing()
    assert len(tm["render_ms"]) == 3 and all(x > 0 for x in tm["render_ms"]) and tm["merge_ms"] > 0 and tm["copy_ms"] == [0.0, 0.0, 0.0]   # replicas on the first device are summed in place
    assert multi.peer_access() == ["same device"] * 3   # how each replica's film reaches the first device is reported, not guessed from the timing


@pytest.mark.gpu
def test_eight_replicas_and_two_film_sizes_on_one_multiscene(pkg, gpu):
    """The 8-way merge (one sum kernel over eight films, integrator.rs:392-396) and a pt_multi_scene rendered at a small, then a larger,
    then the small film size again (the landing / replica buffers follow the film size instead of keeping the first one)."""
    b = pkg.scenes.ganesha_scale(n=40, xres=96, yres=64, spp=4)
    sd, rp_small = b.world_end()
    _, rp_big = pkg.scenes.ganesha_scale(n=40, xres=208, yres=144, spp=4).world_end()
    single = pkg.Scene(gpu, sd)
    multi = pkg.MultiScene(gpu, sd, [0] * 8)
    ct = multi.create_timing()   # replicas 1..7 adopt replica 0's tree and are created concurrently (one host thread each): the call takes less than the eight creations end to end
    assert len(ct["replica_ms"]) == 8 and all(t > 0 for t in ct["replica_ms"]) and 0 < ct["wall_ms"] <= sum(ct["replica_ms"]) * 1.05
    for rp in (rp_small, rp_big, rp_small):
        ref = single.render(rp); rc = single.counters()
        film = multi.render(rp); mc = multi.counters()
        for k in ckeys(("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "path_length_hist", "film_splats")):
            assert mc[k] == rc[k], (k, mc[k], rc[k])
        assert film.shape == ref.shape and np.array_equal(film[..., 3], ref[..., 3])
        np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
def test_a_replica_that_cannot_be_created_fails_the_call_and_leaks_nothing(pkg, gpu):
    """Round 6: replicas 1.. are created concurrently on host threads of their own; one that fails (here: a device ordinal that does not exist) must fail
    pt_multi_scene_create with ITS error text handed back to the caller's thread, tear the finished replicas down, and leave the library usable."""
    sd, rp = pkg.scenes.ganesha_scale(n=16, xres=64, yres=48, spp=2).world_end()
    with pytest.raises(Exception, match=r"replica 2 \(device 99\).*out of range"):
        pkg.MultiScene(gpu, sd, [0, 0, 99, 0])
    ref = pkg.Scene(gpu, sd).render(rp)
    film = pkg.MultiScene(gpu, sd, [0, 0]).render(rp)
    assert np.array_equal(film[..., 3], ref[..., 3])


@pytest.mark.gpu
def test_distinct_devices_equal_the_plain_render(pkg, gpu):
    """Needs two or more visible GPUs (skipped on the one-GPU test box): peer access, hipMemcpyPeerAsync into the landing buffers and
    the per-thread device binding, on really different devices."""
    n = C.c_int(0)
    assert gpu.lib.pt_device_count(C.byref(n)) == 0
    if n.value < 2:
        pytest.skip(f"{n.value} device(s) visible: the distinct-device path needs two")
    devs = list(range(min(n.value, 8)))
    sd, rp = pkg.scenes.ganesha_scale(n=48, xres=160, yres=96, spp=8).world_end()
    single = pkg.Scene(gpu, sd)
    ref = single.render(rp); rc = single.counters()
    multi = pkg.MultiScene(gpu, sd, devs)
    film = multi.render(rp); mc = multi.counters()
    for k in ckeys(("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "path_length_hist", "film_splats")):
        assert mc[k] == rc[k], (k, mc[k], rc[k])
    assert np.array_equal(film[..., 3], ref[..., 3])
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)
    tm = multi.timing()
    assert all(c > 0 for c in tm["copy_ms"][1:])
    assert multi.peer_access()[0] == "same device" and all(p in ("peer access", "staged through the host") for p in multi.peer_access()[1:])


def test_bench_takes_the_one_process_form_for_gpus_n_without_a_launcher():
    """`python bench.py --gpus N` with no WORLD_SIZE must drive N devices itself (pt_multi_render) instead of silently rendering
    on one. Without that many devices it says so and stops -- on this CPU box: before anything is built or rendered."""
    import os, subprocess, sys
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 3:
        pytest.skip("a multi-GPU box: the command would really run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--cpu-seconds", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "asks for device ordinals [0, 1, 2]" in (r.stderr + r.stdout), (r.stdout[-500:], r.stderr[-500:])


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,xres,yres,spp", [(2, 256, 144, 8), (3, 1920, 1080, 2)])
def test_launcher_form_of_bench_runs_with_several_ranks_on_one_gpu(pkg, gpu, tmp_path, trace_mode, ranks, xres, yres, spp):
    """the launcher form of bench.py (one process per rank under `python -m torch.distributed.run`, WORLD_SIZE set, tiles
    sharded `tile % world == rank`, films reduced onto rank 0 -- what the driver's SCALE run starts with nccl on eight GPUs) executed before a multi-GPU
    node sees it: fresh child processes (started before anything in them touches the GPU), all on device 0, backend gloo with the film staged through the
    host. Two ranks at a small film, and THREE at the full 1920x1080 film of the BASELINE configs (an odd world size: 8160 tiles = 3 x 2720). The pool lets six processes
    hold a card: the test's own, the launcher and the ranks -- five ranks got the run killed by the box's process guard, four sit exactly at the limit, three leave
    room for whatever harness the suite runs under; the driver's SCALE run has eight ranks on eight cards -- every rank sizes its workspace from the device's free memory while the others do the same, rank 0's JSON line carries n_gpus and every
    rank's busy time, the 33 MB film is reduced at its real size. The reduced film must equal the single render: weights bit for bit, radiance to float
    summation order. Shape matched: core/integrator.rs:294-296 (tiles fanned out), :392-396 (merge_film_tile)."""
    import os, socket, subprocess, sys
    from conftest import trace_env
    if trace_mode == "exact" and ranks > 2:
        pytest.skip("the launch form does not depend on the walk: the three-rank rehearsal runs in the production instance only")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    env = trace_env({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")})
    out = tmp_path / "film.npy"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", str(ranks), "--devices", ",".join(["0"] * ranks), "--dist-backend", "gloo", "--xres", str(xres), "--yres", str(yres),
           "--spp", str(spp), "--mesh-n", "64", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--other-configs", "off", "--projection", "off", "--dump-film", str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.strip().split("\n") if l.startswith("{")][-1])
    assert line["n_gpus"] == ranks and line["multi_gpu"]["form"].startswith("one process per GPU") and len(line["multi_gpu"]["per_rank_kernel_busy_ms"]) == ranks
    assert all(b > 0 for b in line["multi_gpu"]["per_rank_kernel_busy_ms"]) and line["scaling"] == "strong" and line["config"]["resolution"] == [xres, yres]
    merged = np.load(out)
    sd, rp = pkg.scenes.ganesha_scale(n=64, xres=xres, yres=yres, spp=spp).world_end()
    ref = pkg.Scene(gpu, sd).render(rp)
    assert merged.shape == ref.shape
    assert np.array_equal(merged[..., 3], ref[..., 3])
    np.testing.assert_allclose(merged[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)
