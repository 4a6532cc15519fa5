"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Integer/index work must be bit-exact; radiance is compared with the tolerance stated per test."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _small_scene(pkg, **kw):
    args = dict(n=24, xres=64, yres=48, spp=4)
    args.update(kw)
    return pkg.scenes.ganesha_scale(**args).world_end()


def test_sobol_samples_bit_exact(pkg, gpu, oracle):
    A = pkg._abi
    rng = np.random.default_rng(1)
    n, nd = 4096, 48
    sb = (C.c_int32 * 4)(0, 0, 1920, 1080)
    xy = np.stack([rng.integers(0, 1920, n), rng.integers(0, 1080, n)], axis=1).astype(np.int32)
    sn = rng.integers(0, 4096, n).astype(np.uint32)
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
    for k in ("bvh_nodes_visited", "triangle_tests", "intersect_tests"):
        assert gc[k] == oc[k], k
    tm2 = np.full(len(o), 3.0, np.float32)
    gh = g.trace_any(o, d, tm2); gc = g.counters()
    oh = orc.trace_any(o, d, tm2); oc = orc.counters()
    assert np.array_equal(gh, oh)
    for k in ("bvh_nodes_visited", "triangle_tests", "shadow_tests"):
        assert gc[k] == oc[k], k


@pytest.mark.parametrize("kw", [dict(), dict(env=False), dict(with_normals=True), dict(strategy="uniform"), dict(maxdepth=1)])
def test_film_matches_oracle(pkg, gpu, oracle, kw):
    sd, rp = _small_scene(pkg, **kw)
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film = g.render(rp)
    ref = orc.render(rp, nthreads=1)
    gc, oc = g.counters(), orc.counters()
    for k in ("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "zero_radiance_paths_num",
              "zero_radiance_paths_den", "path_length_hist", "film_splats"):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    # identical sample radiances; the only difference allowed is float summation order of filter splats
    assert np.array_equal(film[..., 3], ref[..., 3])
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)
    rgb, rrgb = g.resolve(film), orc.resolve(ref)
    assert np.abs(rgb - rrgb).max() < 1e-5  # north_star gate is 1e-3
