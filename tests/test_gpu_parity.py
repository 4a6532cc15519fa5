"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Integer/index work must be bit-exact; radiance is compared with the tolerance stated per test."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np
import pytest
from conftest import ckeys, trace_env

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _small_scene(pkg, **kw):
    args = dict(n=24, xres=64, yres=48, spp=4)
    args.update(kw)
    return pkg.scenes.ganesha_scale(**args).world_end()


@pytest.mark.parametrize("nd,max_sample", [(48, 4096), (70, 1 << 12), (1024, 1 << 30), (19, 1 << 30)])
def test_sobol_samples_bit_exact(pkg, gpu, oracle, nd, max_sample):
    """pt_sobol_samples runs the shade kernels' Sampler: eight-dimension windows from the LDS nibble tables (dimensions < 56), from their HBM
    copy, the last dimensions one by one; sample numbers up to 2^30 put index bits above 32 and above 40 (the bit-by-bit tail) to work."""
    A = pkg._abi
    rng = np.random.default_rng(1)
    n = 4096 if nd <= 70 else 512
    sb = (C.c_int32 * 4)(0, 0, 1920, 1080)
    xy = np.stack([rng.integers(0, 1920, n), rng.integers(0, 1080, n)], axis=1).astype(np.int32)
    sn = rng.integers(0, max_sample, n).astype(np.uint32)
    sn[:4] = [0, max_sample - 1, max_sample >> 1, 1]
    outs = []
    for fn in (gpu.lib.pt_sobol_samples, oracle.lib.orc_sobol_samples):
        out = np.zeros((n, nd), np.float32); idx = np.zeros(n, np.uint64)
        st = fn(sb, n, xy.ctypes.data_as(A.i32p), sn.ctypes.data_as(A.u32p), nd, out.ctypes.data_as(A.fp), idx.ctypes.data_as(A.u64p))
        assert st == 0
        outs.append((out, idx))
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))


def test_camera_rays_bit_exact(pkg, gpu, oracle):
    A = pkg._abi
    b = pkg.scenes.ganesha_scale(n=4, xres=1920, yres=1080, spp=1)
    b.cam.update(lensradius=0.05, focaldistance=4.0)
    rp = b.render_params()
    rng = np.random.default_rng(2)
    n = 10000
    cs = np.concatenate([rng.random((n, 2)) * [1920, 1080], rng.random((n, 3))], axis=1).astype(np.float32)
    res = []
    for fn in (gpu.lib.pt_camera_rays, oracle.lib.orc_camera_rays):
        o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32)
        assert fn(C.byref(rp), n, cs.ctypes.data_as(A.fp), o.ctypes.data_as(A.fp), d.ctypes.data_as(A.fp)) == 0
        res.append((o, d))
    assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
    assert np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))


def test_bvh_identical_to_oracle(pkg, gpu, oracle):
    sd, rp = _small_scene(pkg, n=40)
    g = pkg.Scene(gpu, sd); o = oracle.scene(sd)
    gn, go = g.bvh(); on, oo = o.bvh()
    assert len(gn) == len(on)
    assert bytes(gn) == bytes(on)
    assert np.array_equal(go, oo)


def _random_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = (rng.random((n, 3)) * 6 - 3).astype(np.float32); o[:, 1] = np.abs(o[:, 1]) + 0.1
    t = (rng.random((n, 3)) * 2 - 1).astype(np.float32) * 0.8
    d = t - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def test_trace_closest_and_any_bit_exact(pkg, gpu, oracle):
    sd, rp = _small_scene(pkg, n=64)
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    o, d = _random_rays(200000, 3)
    tmax = np.full(len(o), np.inf, np.float32)
    gp, gt, gb = g.trace_closest(o, d, tmax); gc = g.counters()
    op, ot, ob = orc.trace_closest(o, d, tmax); oc = orc.counters()
    assert np.array_equal(gp, op)
    assert np.array_equal(gt.view(np.uint32), ot.view(np.uint32))
    assert np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
    assert (gp != 0xFFFFFFFF).mean() > 0.2
    for k in ckeys(("bvh_nodes_visited", "triangle_tests", "intersect_tests")):
        assert gc[k] == oc[k], k
    tm2 = np.full(len(o), 3.0, np.float32)
    gh = g.trace_any(o, d, tm2); gc = g.counters()
    oh = orc.trace_any(o, d, tm2); oc = orc.counters()
    assert np.array_equal(gh, oh)
    for k in ckeys(("bvh_nodes_visited", "triangle_tests", "shadow_tests")):
        assert gc[k] == oc[k], k


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 511, 512, 513, 1000, 4097])
def test_trace_ragged_ray_counts_and_rays_outside_the_bounds(pkg, gpu, oracle, n):
    """The traversal kernel's refill block hands queue entries out of a 64-entry window of a 512-entry bite (kern_trace.h): counts around those sizes. A third of the rays start
    outside the scene's bounds and point away, a third start outside and point at it -- the four-wide walk enters the root record without the root's own slab test (bvh.rs:725-727)
    and must still return the reference's hits, misses and triangle-test count."""
    sd, rp = _small_scene(pkg, n=24)
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    rng = np.random.default_rng(1000 + n)
    o, d = _random_rays(n, 77 + n)
    far = (rng.random((n, 3)) * 2 - 1).astype(np.float32); far /= np.maximum(np.linalg.norm(far, axis=1, keepdims=True), 1e-6); far *= 50.0
    kind = rng.integers(0, 3, n)
    o = np.where((kind > 0)[:, None], far, o).astype(np.float32)
    away = far / 50.0
    d = np.where((kind == 1)[:, None], away, np.where((kind == 2)[:, None], -away + (rng.random((n, 3)).astype(np.float32) - 0.5) * 0.05, d)).astype(np.float32)
    tmax = np.full(n, np.inf, np.float32)
    gp, gt, gb = g.trace_closest(o, d, tmax); gc = g.counters()
    op, ot, ob = orc.trace_closest(o, d, tmax); oc = orc.counters()
    assert np.array_equal(gp, op)
    assert np.array_equal(gt.view(np.uint32), ot.view(np.uint32)) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
    for k in ckeys(("bvh_nodes_visited", "triangle_tests", "intersect_tests")):
        assert gc[k] == oc[k], k
    gh = g.trace_any(o, d, tmax); gc = g.counters()
    oh = orc.trace_any(o, d, tmax); oc = orc.counters()
    assert np.array_equal(gh, oh)
    for k in ckeys(("bvh_nodes_visited", "triangle_tests", "shadow_tests")):
        assert gc[k] == oc[k], k


@pytest.mark.parametrize("kw", [dict(), dict(env=False), dict(with_normals=True), dict(strategy="uniform"), dict(maxdepth=1)])
def test_film_matches_oracle(pkg, gpu, oracle, kw):
    sd, rp = _small_scene(pkg, **kw)
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film = g.render(rp)
    ref = orc.render(rp, nthreads=1)
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "zero_radiance_paths_num",
              "zero_radiance_paths_den", "path_length_hist", "film_splats")):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    # identical sample radiances; the only difference allowed is float summation order of filter splats
    assert np.array_equal(film[..., 3], ref[..., 3])
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)
    rgb, rrgb = g.resolve(film), orc.resolve(ref)
    assert np.abs(rgb - rrgb).max() < 1e-5  # north_star gate is 1e-3


def _compare_render(pkg, gpu, oracle, sd, rp, rtol=2e-6, atol=1e-7):
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film = g.render(rp)
    ref = orc.render(rp, nthreads=4)
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "zero_radiance_paths_num",
              "zero_radiance_paths_den", "path_length_hist", "film_splats", "sanitized_nan", "sanitized_negative", "sanitized_infinite", "reference_asserts")):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    assert np.array_equal(film[..., 3], ref[..., 3])
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=rtol, atol=atol)
    assert np.abs(g.resolve(film) - orc.resolve(ref)).max() < 1e-4
    return film, ref


def test_material_zoo_matches_oracle(pkg, gpu, oracle):
    """Config C3 material set: matte/Oren-Nayar, plastic, metal, specular + rough glass, mirror, uber, substrate,
    two area lights (one two-sided) and a constant environment; every shade-queue class is exercised."""
    sd, rp = pkg.scenes.material_zoo(n=16, xres=96, yres=64, spp=8).world_end()
    _compare_render(pkg, gpu, oracle, sd, rp)


def test_thin_lens_gaussian_filter_and_crop(pkg, gpu, oracle):
    b = pkg.scenes.ganesha_scale(n=20, xres=72, yres=40, spp=4)
    b.cam.update(lensradius=0.08, focaldistance=5.0)
    b.filter.update(kind="gaussian", radius=(2.0, 2.0), alpha=2.0)
    b.film.update(crop=(0.1, 0.9, 0.2, 1.0))
    sd, rp = b.world_end()
    # gaussian splats overlap between neighbouring samples: float atomics reorder the sums (SURVEY "Hard parts")
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film, ref = g.render(rp), orc.render(rp, nthreads=1)
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(("camera_rays", "bvh_nodes_visited", "triangle_tests", "film_splats", "path_length_hist")):
        assert gc[k] == oc[k], k
    np.testing.assert_allclose(film, ref, rtol=2e-5, atol=1e-6)


def test_point_and_distant_lights_and_empty_light_list(pkg, gpu, oracle):
    b = pkg.scenes.ganesha_scale(n=16, xres=48, yres=32, spp=4, env=False)
    b.light_source("distant", L=(2.0, 2.0, 1.5), from_=(0, 10, 0), to=(0.3, 0, 0.1))
    b.light_source("point", I=(30.0, 10.0, 10.0), from_=(2.0, 3.0, 2.0))
    sd, rp = b.world_end()
    _compare_render(pkg, gpu, oracle, sd, rp)
    # a scene with no lights at all: uniform_sample_onelight consumes no dimensions (integrator.rs:85-86)
    b2 = pkg.host.SceneBuilder()
    b2.film.update(xres=32, yres=32); b2.spp = 2
    b2.look_at((0, 1, 4), (0, 0, 0), (0, 1, 0)); b2.camera(fov=40.0); b2.world_begin()
    P, I, N = pkg.scenes.displaced_sphere(8)
    b2.trianglemesh(P, I)
    sd2, rp2 = b2.world_end()
    film, ref = _compare_render(pkg, gpu, oracle, sd2, rp2)
    assert film[..., :3].max() == 0.0


def test_prebuilt_bvh_is_adopted(pkg, gpu, oracle):
    """The Rust host passes its own BVHAccel (nodes + ordered primitives); here the oracle's tree plays that role."""
    sd, rp = _small_scene(pkg, n=16)
    nodes, ordered = oracle.scene(sd).bvh()
    sd.set_bvh(nodes, ordered)
    g = pkg.Scene(gpu, sd)
    gn, go = g.bvh()
    assert bytes(gn) == bytes(nodes) and np.array_equal(go, ordered)
    _compare_render(pkg, gpu, oracle, sd, rp)


def test_adopted_trees_are_validated(pkg, gpu, oracle):
    """an adopted tree is the caller's data. (1) An interior node whose second child does not lie behind its first child's subtree (a back edge: the flattened
    tree is in pre-order, bvh.rs:662-703) would send the depth-first walks round in circles: refused with PT_ERR_INVALID_ARG. (2) A tree whose child boxes are NOT nested
    in their parents' (here: every leaf box blown up to the root's) is legal for the reference, which tests every box it meets; the four-wide walk
    skips the boxes of collapsed children, so such a scene is walked two-wide, box by box, in both modes -- hits, counters and film == the oracle on the same tree."""
    import copy
    sd, rp = _small_scene(pkg, n=12)
    nodes, ordered = oracle.scene(sd).bvh()
    interior = [i for i in range(len(nodes)) if nodes[i].n_prims == 0]
    bad = type(nodes)(); C.memmove(bad, nodes, C.sizeof(nodes))
    i = interior[len(interior) // 2]
    bad[i].offset = i                      # back edge
    sd_bad = copy.copy(sd); sd_bad.set_bvh(bad, ordered)
    with pytest.raises(Exception, match="malformed BVH node"):
        pkg.Scene(gpu, sd_bad)
    loose = type(nodes)(); C.memmove(loose, nodes, C.sizeof(nodes))
    root = nodes[0]
    # not nested: every LEAF box blown up to the root's box (the interior boxes below the root stay tight, so their children stick out of them)
    for i in range(len(loose)):
        if loose[i].n_prims:
            for k in range(3):
                loose[i].bmin[k] = root.bmin[k]; loose[i].bmax[k] = root.bmax[k]
    sd2 = copy.copy(sd); sd2.set_bvh(loose, ordered)
    g = pkg.Scene(gpu, sd2); orc = oracle.scene(sd2)
    o, d = _random_rays(20000, 9)
    tmax = np.full(len(o), np.inf, np.float32)
    gp, gt, gb = g.trace_closest(o, d, tmax); gc = g.counters()
    op, ot, ob = orc.trace_closest(o, d, tmax); oc = orc.counters()
    assert np.array_equal(gp, op) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32)) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
    for k in ("bvh_nodes_visited", "triangle_tests", "intersect_tests"):   # the node counter in BOTH modes: this scene has no production walk
        assert gc[k] == oc[k], k
    assert any(k["kernel"].endswith(", 0>") for k in g.kernel_stats() if k["kernel"].startswith("k_trace"))


def test_tile_sharding_sums_to_full_render(pkg, gpu):
    sd, rp = _small_scene(pkg, n=16, xres=80, yres=48)
    g = pkg.Scene(gpu, sd)
    full = g.render(rp)
    acc = np.zeros_like(full)
    for r in range(3):
        rp.tile_rank, rp.tile_world = r, 3
        g.render(rp, film=acc)
    rp.tile_rank, rp.tile_world = 0, 1
    # box filter: disjoint pixels. Weights bit for bit; radiance too except where a sample sits exactly on a pixel corner and reaches its
    # neighbours through float atomics, whose order against the owner's additions is not defined (see test_multi_device.py)
    assert np.array_equal(acc[..., 3], full[..., 3]) and (acc == full).mean() > 0.99
    np.testing.assert_allclose(acc[..., :3], full[..., :3], rtol=2e-6, atol=1e-7)


def test_tile_sharding_with_halton_and_volpath(pkg, gpu):
    """The multi-GPU partition (16x16 sample tiles, tile_index % world == rank) under the Halton sampler and the volumetric
    integrator: a sample's radiance depends only on (pixel, sample number), so the shards sum to the single render."""
    b = pkg.scenes.foggy_room(xres=80, yres=48, spp=4); b.sampler = "halton"
    sd, rp = b.world_end()
    g = pkg.Scene(gpu, sd)
    full = g.render(rp)
    acc = np.zeros_like(full)
    for r in range(4):
        rp.tile_rank, rp.tile_world = r, 4
        g.render(rp, film=acc)
    # Halton's first dimension is not clamped below 1 (pbrt_macros:101): a sample can land in the neighbouring pixel, possibly of
    # another shard, so those pixels are float sums in a different order
    assert np.array_equal(acc[..., 3], full[..., 3])
    np.testing.assert_allclose(acc[..., :3], full[..., :3], rtol=2e-6, atol=1e-7)
    assert (acc[..., :3] == full[..., :3]).mean() > 0.99


def test_full_size_properties(pkg, gpu):
    """BASELINE size (1920x1080, 4.3 M triangles) at 2 spp: size-independent properties instead of an oracle run:
    weight sum == spp in every pixel (box filter), film is finite and non-negative in Y, rendering twice is bit-identical
    for the weights and equal within float-atomic reordering for radiance, and passes of different size agree."""
    sd, rp = pkg.scenes.ganesha_scale(n=1466, xres=1920, yres=1080, spp=2).world_end()
    g = pkg.Scene(gpu, sd)
    a = g.render(rp)
    c = g.counters()
    assert c["camera_rays"] == 1920 * 1080 * 2 and sum(c["path_length_hist"]) == c["camera_rays"]
    spill = c["film_splats"] - c["camera_rays"]
    assert 0 <= spill < c["camera_rays"] * 1e-3
    w = a[..., 3]
    assert np.isfinite(a).all() and (np.abs(w - 2.0) <= 1.0).all() and abs(float(w.sum()) - c["film_splats"]) < 1.0
    rp.spp_per_pass = 1
    b = g.render(rp)
    assert np.array_equal(a[..., 3], b[..., 3])
    np.testing.assert_allclose(a[..., :3], b[..., :3], rtol=1e-6, atol=1e-7)


def test_full_size_parity_gate_on_a_crop(pkg, gpu, oracle):
    """SURVEY 8d parity gate at BASELINE size: S2 (4.3 M triangles, 1920x1080 film), 16 spp, the centre 256x256 crop rendered by the
    HIP path and by the oracle (which adopts the library's tree -- identical to its own, see test_bvh_identical_to_oracle -- to skip a
    second 4.3 M-triangle build): per-pixel L-infinity of the normalised film < 1e-3, and the exact counter equalities."""
    b = pkg.scenes.ganesha_scale(n=1466, xres=1920, yres=1080, spp=16)
    b.film.update(crop=(832 / 1920, 1088 / 1920, 412 / 1080, 668 / 1080))
    sd, rp = b.world_end()
    g = pkg.Scene(gpu, sd)
    film = g.render(rp)
    assert film.shape[:2] == (256, 256)
    nodes, ordered = g.bvh()
    sd.set_bvh(nodes, ordered)
    orc = oracle.scene(sd)
    ref = orc.render(rp, nthreads=32)
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "path_length_hist", "film_splats")):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    assert np.array_equal(film[..., 3], ref[..., 3])
    assert np.abs(g.resolve(film) - orc.resolve(ref)).max() < 1e-3   # the gate; in practice the films agree to 2e-6 relative
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)


def test_c2_whole_frame_gate_at_baseline_resolution(pkg, gpu, oracle, trace_mode):
    """the comparison of tools/full_frame_parity.py under the driver's eyes -- the HEADLINE scene (S2, 4 298 316 triangles) over
    the WHOLE 1920x1080 frame at 16 spp (33.2 M samples, ~90 M rays), HIP path against the CPU oracle: every work counter equal (the production walk's
    node counter aside; the exact walk's node counter too), weights identical, normalised L-infinity < 1e-3 (the north-star gate; in practice 4e-7).
    One oracle render serves both walks, so the "exact" instance of this test is a skip."""
    import os
    if trace_mode == "exact":
        pytest.skip("both walks are compared inside the production-mode instance (one oracle render of the whole frame)")
    sd, rp = pkg.scenes.ganesha_scale(n=1466, xres=1920, yres=1080, spp=16).world_end()
    g = pkg.Scene(gpu, sd)
    film = g.render(rp); gc = g.counters()
    gpu.set_trace_exact(True)
    try:
        film_x = g.render(rp); gx = g.counters()
    finally:
        gpu.set_trace_exact(False)
    assert film.shape[:2] == (1080, 1920)
    nodes, ordered = g.bvh()
    sd.set_bvh(nodes, ordered)      # the oracle adopts the library's tree (identical to its own: test_bvh_identical_to_oracle)
    orc = oracle.scene(sd)
    ref = orc.render(rp, nthreads=os.cpu_count())
    oc = orc.counters()
    keys = ("camera_rays", "intersect_tests", "shadow_tests", "triangle_tests", "path_length_hist", "film_splats", "zero_radiance_paths_num", "zero_radiance_paths_den",
            "sanitized_nan", "sanitized_negative", "sanitized_infinite", "reference_asserts")
    for k in keys:
        assert gc[k] == oc[k], (k, gc[k], oc[k])
        assert gx[k] == oc[k], (k, gx[k], oc[k])
    assert gx["bvh_nodes_visited"] == oc["bvh_nodes_visited"]
    assert gc["camera_rays"] == 1920 * 1080 * 16
    for f in (film, film_x):
        assert np.array_equal(f[..., 3], ref[..., 3])
        assert np.abs(g.resolve(f) - orc.resolve(ref)).max() < 1e-3
        np.testing.assert_allclose(f[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)


def test_spheres_c1_matches_oracle(pkg, gpu, oracle):
    """Config C1: analytic spheres (mirror, glass, partial plastic sphere) + distant + area light."""
    sd, rp = pkg.scenes.spheres_c1(xres=96, yres=96, spp=8).world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    gn, go = g.bvh(); on, oo = orc.bvh()
    assert bytes(gn) == bytes(on) and np.array_equal(go, oo)
    film, ref = g.render(rp), orc.render(rp, nthreads=4)
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests", "path_length_hist", "film_splats")):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    np.testing.assert_allclose(film, ref, rtol=2e-6, atol=1e-7)


def test_spot_light_and_power_distribution(pkg, gpu, oracle):
    b = pkg.scenes.ganesha_scale(n=16, xres=48, yres=32, spp=4, strategy="power")
    b.light_source("spot", I=(40.0, 40.0, 30.0), from_=(1.0, 3.0, 2.0), to=(0.0, 0.0, 0.0), coneangle=25.0, conedeltaangle=8.0)
    b.light_source("point", I=(5.0, 5.0, 9.0), from_=(-2.0, 2.0, 1.0))
    sd, rp = b.world_end()
    _compare_render(pkg, gpu, oracle, sd, rp)


def test_instancing_matches_oracle(pkg, gpu, oracle):
    """Row a12 / config C4 in miniature: ObjectInstances (multi-primitive objects with their own BVH, a single-triangle
    object without one, an identity-transform instance) over a ground mesh; two-level traversal, bit-exact counters."""
    sd, rp = pkg.scenes.instanced_garden(xres=96, yres=64, spp=8).world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    gn, go = g.bvh(); on, oo = orc.bvh()
    assert bytes(gn) == bytes(on) and np.array_equal(go, oo)
    _compare_render(pkg, gpu, oracle, sd, rp)
    # closest-hit records through instances
    o, d = _random_rays(50000, 11)
    o[:, 1] += 2.0
    tmax = np.full(len(o), np.inf, np.float32)
    gp, gt, gb = g.trace_closest(o, d, tmax); op, ot, ob = orc.trace_closest(o, d, tmax)
    assert np.array_equal(gp, op) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32)) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
    assert np.array_equal(g.trace_any(o, d, np.full(len(o), 5.0, np.float32)), orc.trace_any(o, d, np.full(len(o), 5.0, np.float32)))


@pytest.mark.parametrize("rough", [False, True])
def test_subsurface_matches_oracle(pkg, gpu, oracle, rough):
    """Row a23 / config C5: `subsurface` (named medium, scaled) on a triangle mesh and `kdsubsurface` on a sphere shape.
    path.rs:177-204: probe-ray chains (bssrdf.rs:367-395), Sp / pdf_sp, NEE + BSDF sampling through the adapter lobe.
    Every lane of k_trace<.., PROBE> walks a whole chain once and keeps the last 8 matching intersections, so all work counters
    (Scene::intersect calls, nodes, triangle and sphere tests) equal the oracle's."""
    sd, rp = pkg.scenes.subsurface_c5(n=16, xres=96, yres=64, spp=8, rough=rough).world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film, ref = g.render(rp), orc.render(rp, nthreads=4)
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(("camera_rays", "shadow_tests", "path_length_hist", "film_splats", "zero_radiance_paths_num", "zero_radiance_paths_den",
              "sanitized_nan", "sanitized_negative", "sanitized_infinite", "intersect_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests")):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    assert np.array_equal(film[..., 3], ref[..., 3])
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)
    assert np.abs(g.resolve(film) - orc.resolve(ref)).max() < 1e-4


@pytest.mark.parametrize("rough", [False, True])
def test_subsurface_with_sigma_textures_matches_oracle(pkg, gpu, oracle, rough):
    """subsurface.rs:100-101: sigma_a / sigma_s are textures evaluated on the outgoing interaction; the BSSRDF built there (its sigma_t and
    albedo) is the one the probe chain, Sp / pdf_sp and the adapter lobe at the exit point use. The evaluated coefficients travel with the
    path (BssSoA.sa_* / sc_*); a 3-D and a planar checkerboard drive them here."""
    sd, rp = pkg.scenes.subsurface_c5(n=16, xres=96, yres=64, spp=8, rough=rough, textured_sigma=True).world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film, ref = g.render(rp), orc.render(rp, nthreads=4)
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(("camera_rays", "shadow_tests", "path_length_hist", "film_splats", "zero_radiance_paths_num", "zero_radiance_paths_den",
              "sanitized_nan", "sanitized_negative", "sanitized_infinite", "intersect_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests")):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    assert np.array_equal(film[..., 3], ref[..., 3])
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)
    plain, _ = pkg.scenes.subsurface_c5(n=16, xres=96, yres=64, spp=8, rough=rough).world_end()
    assert not np.allclose(film[..., :3], pkg.Scene(gpu, plain).render(rp)[..., :3], rtol=1e-3)   # (the textures do change the image)


def test_kdsubsurface_with_textures_matches_oracle(pkg, gpu, oracle):
    """kdsubsurface.rs:96-99 with textured `Kd` / `mfp`: subsurface_from_diffuse (invert_catmull_rom over the table's effective albedo,
    interpolation.rs:265-330) runs at every hit -- in k_shade<5, 2> on the device, in the oracle's compute_scattering_functions -- and the
    resulting coefficients travel with the path like the textured sigma_a / sigma_s of `subsurface`."""
    sd, rp = pkg.scenes.subsurface_c5(n=16, xres=96, yres=64, spp=8, textured_kd=True).world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film, ref = g.render(rp), orc.render(rp, nthreads=4)
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(("camera_rays", "shadow_tests", "path_length_hist", "film_splats", "zero_radiance_paths_num", "zero_radiance_paths_den",
              "sanitized_nan", "sanitized_negative", "sanitized_infinite", "intersect_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests")):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    assert np.array_equal(film[..., 3], ref[..., 3])
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)


def test_long_probe_chains_fall_back_to_an_uncounted_rewalk(pkg, gpu, oracle):
    """A stack of 40 thin sheets of one subsurface material: probe chains along the normal cross up to 40 matching surfaces, far more
    than the 8-entry ring of k_trace<.., PROBE>, so chains whose selected intersection has left the ring are walked a second
    time -- with the rewalk's work left out of the counters, which must still equal the oracle's single walk."""
    b = pkg.scenes.subsurface_sheets(xres=64, yres=48, spp=8)
    sd, rp = b.world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film, ref = g.render(rp), orc.render(rp, nthreads=4)
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(("camera_rays", "shadow_tests", "path_length_hist", "film_splats", "intersect_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests")):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    assert oc["intersect_tests"] > 5 * oc["camera_rays"]       # the chains really are long (40 matches where a probe crosses the stack)
    assert np.array_equal(film[..., 3], ref[..., 3])
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("shape", [(16, 8), (8, 8), (32, 4), (2, 16)])
def test_image_environment_map_matches_oracle(pkg, gpu, oracle, shape):
    """Row a21 InfiniteAreaLight with an image map: level-0 bilinear `le`, Distribution2D importance sampling / pdf, and the
    spatial light grid built from it. (32, 4) and (2, 16): aspects beyond 2:1, whose importance image comes from a coarser level."""
    tex = pkg.scenes.sky_env(*shape)
    b = pkg.scenes.ganesha_scale(n=16, xres=64, yres=40, spp=8, env=False)
    b.rotate(-90.0, 1.0, 0.0, 0.0)
    b.light_source("infinite", texels=tex, L=(0.5, 0.5, 0.5), scale=2.0)
    sd, rp = b.world_end()
    _compare_render(pkg, gpu, oracle, sd, rp)


def test_sphere_area_lights_match_oracle(pkg, gpu, oracle):
    """Rows a14/a21: DiffuseAreaLight on sphere shapes -- cone sampling (two-sided and the one-sided zero-normal quirk),
    the inside-the-sphere branch with shape_pdfwi (negative pdfs squared by the power heuristic), spatial light grid."""
    sd, rp = pkg.scenes.sphere_lights(xres=96, yres=64, spp=8).world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film, ref = g.render(rp), orc.render(rp, nthreads=4)
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests", "path_length_hist",
              "film_splats", "zero_radiance_paths_num", "zero_radiance_paths_den")):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    np.testing.assert_allclose(film, ref, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("trilinear,bump,noise", [(False, False, False), (True, False, False), (False, True, False), (True, False, True)])
def test_textures_match_oracle(pkg, gpu, oracle, trilinear, bump, noise):
    """image maps (EWA / trilinear MIPMap; uv, planar, spherical mappings; repeat and black wrap; RGB and float
    memory), checkerboards (2-D closed form / point sampled over uv, planar and cylindrical mappings; 3-D), scale, mix,
    bilerp, uv -- camera-ray differentials at the first vertex, zero-width lookups afterwards."""
    # bump=True adds `bumpmap` float textures (image, closed-form checkerboard, mix) on a sphere, a mesh with shading
    # normals (dndu/dndv path) and the floor: bump() + set_shading_geometry (material.rs:46-87, interaction.rs:228-249)
    # noise=True adds the Perlin-noise textures: marble, fbm (as Oren-Nayar sigma), wrinkled, windy (as a bump map), dots
    sd, rp = pkg.scenes.textured(xres=96, yres=64, spp=8, trilinear=trilinear, bump=bump, noise=noise).world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film, ref = g.render(rp), orc.render(rp, nthreads=4)
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests", "path_length_hist",
              "film_splats", "zero_radiance_paths_num", "zero_radiance_paths_den")):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    np.testing.assert_allclose(film, ref, rtol=2e-6, atol=1e-7)


def test_textures_with_thin_lens(pkg, gpu, oracle):
    """Lens branch of generate_ray_differential (perspective.rs:147-165): the auxiliary rays start on the lens sample."""
    b = pkg.scenes.textured(xres=64, yres=40, spp=4)
    b.cam.update(lensradius=0.05, focaldistance=7.0)
    sd, rp = b.world_end()
    _compare_render(pkg, gpu, oracle, sd, rp)


@pytest.mark.parametrize("instanced", [True, False])
def test_alpha_masks_match_oracle(pkg, gpu, oracle, instanced):
    """Alpha masks in Triangle::intersect / intersect_p (triangle.rs:275-285,497-545): cut-out cards (checkerboard and
    image float textures), `shadowalpha`, constant-zero alpha, inside object instances and at top level; the traversal
    kernels evaluate the mask texture at candidate hits. Closest-hit / any-hit records are compared as well."""
    sd, rp = pkg.scenes.alpha_foliage(xres=96, yres=64, spp=8, instanced=instanced).world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    _compare_render(pkg, gpu, oracle, sd, rp)
    o, d = _random_rays(40000, 21)
    o[:, 1] += 1.0; o *= np.float32(0.5)
    tmax = np.full(len(o), np.inf, np.float32)
    gp, gt, gb = g.trace_closest(o, d, tmax); op, ot, ob = orc.trace_closest(o, d, tmax)
    assert np.array_equal(gp, op) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32)) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
    assert np.array_equal(g.trace_any(o, d, np.full(len(o), 6.0, np.float32)), orc.trace_any(o, d, np.full(len(o), 6.0, np.float32)))


@pytest.mark.parametrize("textured", [False, True])
def test_translucent_material_matches_oracle(pkg, gpu, oracle, textured):
    """materials/translucent.rs: Lambertian reflection + transmission and microfacet reflection + transmission scaled by
    `reflect` / `transmit`; a sheet with neither has no BSDF and is passed through (App. A #14)."""
    sd, rp = pkg.scenes.translucent_panels(textured=textured).world_end()
    film, ref = _compare_render(pkg, gpu, oracle, sd, rp, rtol=2e-5 if textured else 2e-6, atol=1e-6 if textured else 1e-7)
    assert film[..., :3].sum() > 0


@pytest.mark.parametrize("textured", [False, True])
def test_mix_material_matches_oracle(pkg, gpu, oracle, textured):
    """materials/mix.rs: both materials' BxDFs as ScaledBxDFs (amount, 1 - amount) in the first material's frame; the second
    material's textures are evaluated on an interaction without differentials; only the first material's bump map acts."""
    sd, rp = pkg.scenes.mix_materials(textured=textured).world_end()
    film, ref = _compare_render(pkg, gpu, oracle, sd, rp, rtol=2e-5 if textured else 2e-6, atol=1e-6 if textured else 1e-7)
    assert film[..., :3].sum() > 0


def test_mix_material_limits_are_reported(pkg, gpu):
    b = pkg.scenes.mix_materials()
    b.material("uber"); u = b.material_id
    b.material("plastic"); p2 = b.material_id
    b.material("mix", namedmaterial1=u, namedmaterial2=p2)   # 5 + 2 BxDFs
    P, I = pkg.scenes.quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)); b.trianglemesh(P, I)
    sd, rp = b.world_end()
    with pytest.raises(Exception, match="more than 5 BxDFs"): pkg.Scene(gpu, sd)


@pytest.mark.parametrize("textured", [False, True])
def test_disney_material_matches_oracle(pkg, gpu, oracle, textured):
    """materials/disney.rs (no BSSRDF): DisneyDiffuse / FakeSS / Retro / Sheen / Clearcoat (GTR1, its own sampling and the
    `wi + wi` pdf), the separable-G microfacet distribution with DisneyFresnel, specular transmission, the thin-surface set."""
    sd, rp = pkg.scenes.disney_spheres(textured=textured).world_end()
    film, ref = _compare_render(pkg, gpu, oracle, sd, rp, rtol=2e-5 if textured else 2e-6, atol=1e-6 if textured else 1e-7)
    assert film[..., :3].sum() > 0


def test_disk_shapes_and_lights_match_oracle(pkg, gpu, oracle):
    """shapes/disk.rs: intersect (incl. its world-space r.d.z parallel test), sample / pdf as diffuse area lights, annulus and
    partial sweeps, scaled / mirrored / reversed / instanced disks."""
    sd, rp = pkg.scenes.disk_scene().world_end()
    film, ref = _compare_render(pkg, gpu, oracle, sd, rp)
    assert film[..., :3].sum() > 0
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    o, d = _random_rays(100000, 11); tmax = np.full(len(o), np.inf, np.float32)
    gp, gt, gb = g.trace_closest(o, d, tmax); op, ot, ob = orc.trace_closest(o, d, tmax)
    assert np.array_equal(gp, op) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32))
    assert np.array_equal(g.trace_any(o, d, tmax), orc.trace_any(o, d, tmax))
    for k in ckeys(("bvh_nodes_visited", "intersect_tests", "shadow_tests")): assert g.counters()[k] == orc.counters()[k], k


def test_disney_limits_are_reported(pkg, gpu):
    b = pkg.scenes.disney_spheres()
    b.material("disney", sheen=0.5, clearcoat=0.5, spectrans=0.5, thin=True)   # 8 BxDFs
    P, I = pkg.scenes.quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)); b.trianglemesh(P, I)
    sd, rp = b.world_end()
    with pytest.raises(Exception, match="more than 5 BxDFs"): pkg.Scene(gpu, sd)
    b.texture("c", "spectrum", "checkerboard")
    with pytest.raises(NotImplementedError): b.material("disney", color="c", scatterdistance=(0.1, 0.1, 0.1))


@pytest.mark.parametrize("g", [0.1, 0.03])
def test_disney_bssrdf_matches_oracle(pkg, gpu, oracle, g):
    """DisneyMaterial with a scatter distance (disney.rs:442-704,768-776): a specular-transmission lobe and the DisneyBSSRDF
    (two-exponential profile, analytic sample_sr / pdf_sr) walked by the same probe-chain kernels as the tabulated BSSRDF."""
    b = pkg.scenes.disney_spheres(xres=64, yres=48, spp=8)
    b.material("disney", color=(0.8, 0.5, 0.4), scatterdistance=(g, g * 0.6, g * 0.3), roughness=0.4, eta=1.4)
    b.attribute_begin(); b.translate(0.0, 0.6, 1.6); b.sphere(radius=0.6); b.attribute_end()
    P, I, N = pkg.scenes.displaced_sphere(8, with_normals=True)
    b.attribute_begin(); b.translate(-1.6, 0.5, 1.8); b.scale(0.5, 0.5, 0.5); b.trianglemesh(P, I, N=N); b.attribute_end()
    sd, rp = b.world_end()
    gsc = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film, ref = gsc.render(rp), orc.render(rp, nthreads=4)
    gc, oc = gsc.counters(), orc.counters()
    for k in ckeys(("camera_rays", "shadow_tests", "path_length_hist", "film_splats", "intersect_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests")): assert gc[k] == oc[k], k
    assert np.array_equal(film[..., 3], ref[..., 3])
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)


def test_textured_triangle_only_scene(pkg, gpu, oracle):
    """Regression (found by tests/test_fuzz_parity.py, seed 25): textures select the general shade kernels, which read the instance
    of a hit, while a scene of triangles only is traversed by k_trace<*, 0>, which never writes it."""
    b = pkg.scenes.ganesha_scale(n=12, xres=64, yres=48, spp=4)
    b.texture("chk", "spectrum", "checkerboard", uscale=4.0, vscale=4.0, tex1=(0.8, 0.2, 0.2), tex2=(0.2, 0.2, 0.8))
    b.material("matte", Kd="chk")
    P, I = pkg.scenes.quad((-1.5, -1.2, 1.0), (1.5, -1.2, 1.0), (1.5, 0.5, 0.2), (-1.5, 0.5, 0.2))
    b.trianglemesh(P, I, UV=np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32))
    sd, rp = b.world_end()
    assert len(b.spheres) == 0 and not b.instances
    _compare_render(pkg, gpu, oracle, sd, rp, rtol=2e-5, atol=1e-6)


def _negative_light_scene(pkg, spp=4):
    """A matte floor under a point light with a NEGATIVE intensity (a .pbrt file may say so): every lit vertex's estimate has
    Ld.y() < 0, where the reference's `assert!(Ld.y() >= 0.0)` (path.rs:143) panics."""
    b = pkg.host.SceneBuilder()
    b.film.update(xres=32, yres=24); b.spp = spp
    b.integ.update(maxdepth=3, strategy="uniform")   # (the spatial strategy floors a light with a negative contribution at 0.001 x the average)
    b.look_at((0.0, 2.0, 5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=40.0)
    b.world_begin()
    b.light_source("point", from_=(0.0, 3.0, 0.0), I=(-5.0, -4.0, -3.0))
    b.light_source("infinite", L=(0.2, 0.2, 0.2))
    b.material("matte", Kd=(0.6, 0.5, 0.4))
    P, I = pkg.scenes.quad((-4.0, 0.0, -4.0), (-4.0, 0.0, 4.0), (4.0, 0.0, 4.0), (4.0, 0.0, -4.0)); b.trianglemesh(P, I)
    return b.world_end()


def _odd_triangle_lights_scene(pkg, strategy, reverse=False):
    """Every branch of the per-light record of DeviceScene::light_rec (k_light_area): an emissive mesh with interpolated normals (Triangle::sample face-forwards
    to them), one with uvs whose partials are degenerate (uv of all three vertices equal: coordinate_system fallback, still a valid hit), one with
    three collinear vertices (area 0, 1 / area = inf, degenerate partials: Shape::pdf_wi finds no intersection), a two-sided one and a reversed one."""
    b = pkg.host.SceneBuilder()
    b.film.update(xres=48, yres=32); b.spp = 8
    b.integ.update(maxdepth=4, strategy=strategy)
    b.look_at((0.0, 2.2, 6.0), (0.0, 0.6, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=42.0)
    b.world_begin()
    b.light_source("infinite", L=(0.05, 0.06, 0.08))
    tri = np.array([[0, 1, 2]])
    b.attribute_begin(); b.area_light_source(L=(9.0, 8.0, 6.0))   # interpolated normals, pointing down and outwards
    P = np.array([(-1.5, 3.0, -0.5), (-0.5, 3.0, 0.8), (-0.3, 3.0, -0.7)], np.float32)
    b.trianglemesh(P, tri, N=np.array([(0.2, -1.0, 0.0), (0.0, -1.0, 0.3), (-0.3, -1.0, -0.1)], np.float32)); b.attribute_end()
    b.attribute_begin(); b.area_light_source(L=(4.0, 7.0, 9.0), twosided=True)   # all uvs equal: degenerate uv determinant
    P = np.array([(0.4, 2.6, -0.6), (1.6, 2.9, -0.2), (0.9, 2.7, 0.9)], np.float32)
    b.trianglemesh(P, tri, UV=np.array([(0.3, 0.3)] * 3, np.float32)); b.attribute_end()
    b.attribute_begin(); b.area_light_source(L=(20.0, 20.0, 20.0))   # three collinear vertices
    b.trianglemesh(np.array([(-1.0, 2.0, 1.0), (0.0, 2.0, 1.0), (1.0, 2.0, 1.0)], np.float32), tri); b.attribute_end()
    b.attribute_begin(); b.area_light_source(L=(6.0, 3.0, 3.0))
    if reverse: b.toggle_reverse_orientation()
    P, I = pkg.scenes.quad((2.0, 0.2, -1.0), (2.0, 0.2, 1.0), (2.0, 1.8, 1.0), (2.0, 1.8, -1.0)); b.trianglemesh(P, I); b.attribute_end()
    b.material("matte", Kd=(0.6, 0.55, 0.5))
    P, I = pkg.scenes.quad((-5.0, 0.0, -5.0), (-5.0, 0.0, 5.0), (5.0, 0.0, 5.0), (5.0, 0.0, -5.0)); b.trianglemesh(P, I)
    b.material("plastic", Kd=(0.3, 0.4, 0.5), Ks=(0.4, 0.4, 0.4), roughness=0.15)
    b.attribute_begin(); b.translate(-0.6, 0.7, 0.3); b.sphere(radius=0.7); b.attribute_end()
    return b.world_end()


@pytest.mark.gpu
@pytest.mark.parametrize("strategy,reverse", [("spatial", False), ("uniform", True), ("power", False)])
def test_triangle_light_records_match_oracle(pkg, gpu, oracle, strategy, reverse):
    _compare_render(pkg, gpu, oracle, *_odd_triangle_lights_scene(pkg, strategy, reverse))


@pytest.mark.gpu
def test_reference_asserts_counter_matches_oracle(pkg, gpu, oracle):
    sd, rp = _negative_light_scene(pkg, spp=8)
    film, ref = _compare_render(pkg, gpu, oracle, sd, rp)      # compares reference_asserts exactly
    g = pkg.Scene(gpu, sd); g.render(rp)
    assert g.counters()["reference_asserts"] > 0


_PAD_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from _pkg import import_pkg
pkg = import_pkg()
import torch
torch.cuda.init()
lib = pkg.load_library(); lib.init(0)
sd, rp = pkg.scenes.instanced_garden(xres=96, yres=64, spp=8).world_end()
g = pkg.Scene(lib, sd)
d = np.load({rays!r})
gp, gt, gb = g.trace_closest(d["o"], d["d"], d["tmax"]); c1 = g.counters()
gh = g.trace_any(d["o"], d["d"], d["tm2"]); c2 = g.counters()
film = g.render(rp); c3 = g.counters()
np.savez({out!r}, gp=gp, gt=gt, gb=gb, gh=gh, film=film, tri1=c1["triangle_tests"], tri2=c2["triangle_tests"], tri3=c3["triangle_tests"], rays3=c3["intersect_tests"] + c3["shadow_tests"])
"""


def test_records_and_packets_beyond_four_gigabytes(pkg, gpu, oracle, tmp_path, trace_mode):
    """Round 5: a pool of four-wide records and packets beyond 4 GB (what the buffer loads' 32-bit BYTE offsets of k_trace<.., 1> reach) is walked by k_trace<.., 2>: the same
    walk through 64-bit global loads (up to 64 GB; a structured buffer resource was tried first and dropped -- its index x stride wraps at 32 bits, profiles/r5/NOTES.md section 5).
    PT_TEST_POOL_PAD_RECORDS puts 34 M unused records (4.35 GB) in front of a small instanced scene's pool, in a process of
    its own (pt_init reads it): every record and packet -- top-level tree, object trees, instance packets -- then lies beyond the old reach. Hits, films and
    triangle counters == oracle. (tools/big_scene_parity.py does the same with a scene that is that large by itself.)"""
    if trace_mode == "exact":
        pytest.skip("the pool is the production walk's (the two-wide records have their own array)")
    sd, rp = pkg.scenes.instanced_garden(xres=96, yres=64, spp=8).world_end()
    orc = oracle.scene(sd)
    o, d = _random_rays(60000, 11)
    o = (o * np.float32(3.0)).astype(np.float32)
    tmax = np.full(len(o), np.inf, np.float32); tm2 = np.full(len(o), 3.0, np.float32)
    np.savez(tmp_path / "rays.npz", o=o, d=d, tmax=tmax, tm2=tm2)
    code = _PAD_CHILD.format(root=ROOT, rays=str(tmp_path / "rays.npz"), out=str(tmp_path / "out.npz"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(trace_env(), PT_TEST_POOL_PAD_RECORDS="34000000"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    g = np.load(tmp_path / "out.npz")
    op, ot, ob = orc.trace_closest(o, d, tmax); t1 = orc.counters()["triangle_tests"]
    assert np.array_equal(g["gp"], op) and np.array_equal(g["gt"].view(np.uint32), ot.view(np.uint32)) and np.array_equal(g["gb"].view(np.uint32), ob.view(np.uint32))
    assert (op != 0xFFFFFFFF).mean() > 0.05 and int(g["tri1"]) == t1
    oh = orc.trace_any(o, d, tm2); t2 = orc.counters()["triangle_tests"]
    assert np.array_equal(g["gh"], oh) and int(g["tri2"]) == t2
    ref = orc.render(rp, nthreads=4); oc = orc.counters()
    assert int(g["tri3"]) == oc["triangle_tests"] and int(g["rays3"]) == oc["intersect_tests"] + oc["shadow_tests"]
    assert np.array_equal(g["film"][..., 3], ref[..., 3])
    np.testing.assert_allclose(g["film"][..., :3], ref[..., :3], rtol=3e-6, atol=1e-6)


@pytest.mark.parametrize("as_written,split", [(0, "sah"), (1, "sah"), (0, "hlbvh")])
def test_triangle_watertight_twin_through_the_hip_path(pkg, gpu, oracle, as_written, split):
    """tests/shapes.rs:36-146 triangle_watertight through pt_trace_closest: the reference's mesh (RNG::new(12111), 16 x 16) as a scene, the 200 000 rays of its
    100 000 seeds (oracle/ref_kats_shapes.cpp generates both from the same PCG32 streams); under the reference's SAH tree and under the GPU-built HLBVH. Every ray must hit (closed mesh; on the mesh as the Rust file
    writes it, which is open along phi = 0 -- tests/test_oracle_kats.py -- exactly the rays that hit no triangle one by one must miss), and (prim, t, b) must be
    the oracle's BVHAccel::intersect over the same scene, bit for bit."""
    from test_oracle_kats import _watertight
    failures, v, idx, ro, rd, nh = _watertight(oracle, 100000, as_written)
    b = pkg.host.SceneBuilder()
    b.split_method = split    # "hlbvh": the GPU-built tree (multi-triangle leaves, another visiting order): the vertex rays' near-ties resolve by THAT tree's order, on device as in the oracle
    b.trianglemesh(v, idx)
    sd, _ = b.world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    tmax = np.full(len(ro), np.inf, np.float32)
    gp, gt, gb = g.trace_closest(ro, rd, tmax); gc = g.counters()
    op, ot, ob = orc.trace_closest(ro, rd, tmax); oc = orc.counters()
    assert np.array_equal(gp != 0xFFFFFFFF, nh >= 1)
    if not as_written:
        assert failures == 0 and (gp != 0xFFFFFFFF).all()
    assert np.array_equal(gp, op)
    assert np.array_equal(gt.view(np.uint32), ot.view(np.uint32)) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
    for k in ckeys(("bvh_nodes_visited", "triangle_tests", "intersect_tests")):
        assert gc[k] == oc[k], k


def test_distribution1d_twins_on_the_device(pkg, gpu, oracle):
    """tests/sampling.rs:259-283 distribution1d_continuous on the device's dist_sample_continuous (csrc/dev_light.h, the environment map's sampler) through
    pt_dist1d_sample, the same checks as the oracle twin; then device == oracle bit for bit on 20 000 numbers incl. both ends and every cdf knot +- 1 ulp, for the
    continuous and the discrete routine (tests/sampling.rs:202-257's distribution [0, 1, 0, 3] among them)."""
    from test_oracle_kats import dist1d_continuous_checks
    A = pkg._abi

    def device(func, u, discrete):
        func = np.ascontiguousarray(func, np.float32); u = np.ascontiguousarray(u, np.float32)
        x = np.zeros(len(u), np.float32); pdf = np.zeros(len(u), np.float32); off = np.zeros(len(u), np.int32)
        assert gpu.lib.pt_dist1d_sample(func.ctypes.data_as(A.fp), len(func), discrete, len(u), u.ctypes.data_as(A.fp), x.ctypes.data_as(A.fp),
                                        pdf.ctypes.data_as(A.fp), off.ctypes.data_as(A.i32p)) == 0
        return x, pdf, off
    func = [1.0, 1.0, 2.0, 4.0, 8.0]
    dist1d_continuous_checks(lambda u: tuple(a[0].item() for a in device(func, [u], 0)))
    rng = np.random.default_rng(5)
    for func in ([1.0, 1.0, 2.0, 4.0, 8.0], [0.0, 1.0, 0.0, 3.0], [0.0, 0.0, 0.0], [5.0], list(rng.random(257) ** 4), [0.0] * 7 + [2.0] + [0.0] * 9):
        f32 = np.array(func, np.float32)
        cdf = np.concatenate([[0], np.cumsum(f32.astype(np.float64)) / max(f32.sum(dtype=np.float64), 1e-30)]).astype(np.float32)
        knots = np.concatenate([np.nextafter(cdf, np.float32(-1)), cdf, np.nextafter(cdf, np.float32(2))]).clip(0, 1)
        u = np.concatenate([[0.0, 1.0, np.float32(1) - np.float32(2 ** -24)], knots, rng.random(20000)]).astype(np.float32)
        for discrete in (0, 1):
            gx, gpdf, goff = device(f32, u, discrete)
            for i, ui in enumerate(u):
                pdf = C.c_float(); off = C.c_int(-1)
                if discrete:
                    o_off = oracle.lib.orc_dist1d_sample_discrete(f32.ctypes.data_as(A.fp), len(f32), float(ui), C.byref(pdf), None); o_x = 0.0
                else:
                    o_x = oracle.lib.orc_dist1d_sample_continuous(f32.ctypes.data_as(A.fp), len(f32), float(ui), C.byref(pdf), C.byref(off)); o_off = off.value
                assert goff[i] == o_off, (func[:5], discrete, ui)
                assert np.float32(gx[i]).view(np.uint32) == np.float32(o_x).view(np.uint32) and np.float32(gpdf[i]).view(np.uint32) == np.float32(pdf.value).view(np.uint32), (func[:5], discrete, ui)
