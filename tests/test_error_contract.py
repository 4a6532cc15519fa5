"""The boundary's error contract (SURVEY 8b "Errors"; include/mi355pt.h PtStatus): where the reference panics -- or, worse, has no check -- the C ABI returns a
status, never aborts, and the scene handle stays usable. Each test provokes one status through the C ABI on the GPU and then renders / traces again with the same handle."""
import ctypes as C
import os
import numpy as np
import pytest
from conftest import ckeys

pytestmark = pytest.mark.gpu


def _status(gpu, fn, *args):
    st = fn(*args)
    return st, gpu.lib.pt_last_error().decode(errors="replace")


def _furnace(pkg, depth, spp=4, res=16, rr=0.0):
    """A closed box whose faces are white Lambertian two-sided emitters: no ray leaves, beta stays 1, roulette off (rrthreshold 0): every path runs to maxdepth."""
    b = pkg.host.SceneBuilder()
    b.film.update(xres=res, yres=res); b.spp = spp
    b.integ.update(maxdepth=depth, rrthreshold=rr)
    b.look_at((0.1, 0.2, 0.0), (0.3, 0.1, 1.0), (0.0, 1.0, 0.0)); b.camera(fov=70.0)
    b.world_begin()
    b.material("matte", Kd=(1.0, 1.0, 1.0))
    b.area_light_source(L=(0.01, 0.01, 0.01), twosided=True)
    c = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
    faces = [(0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (3, 2, 6, 7), (0, 3, 7, 4), (1, 2, 6, 5)]
    idx = []
    for f in faces:
        idx += [(f[0], f[1], f[2]), (f[0], f[2], f[3])]
    b.trianglemesh(np.array(c, np.float32), np.array(idx, np.uint32))
    return b


def test_sobol_dimension_overflow_is_a_status_not_an_abort(pkg, gpu, oracle):
    """samplers/sobol.rs:69-73: `sample_dimension` panics at dimension >= 1024. A matte vertex draws 7 dimensions (1 + 2 + 2 for the light, 2 for the BSDF) after the camera's 5:
    vertex 146 asks for dimension 1027. maxdepth 200 in the furnace box gets every path there: pt_render returns PT_ERR_SOBOL_DIMENSIONS (the oracle's render reports the same
    status at the same point), and the same scene renders at maxdepth 100 (705 dimensions) right afterwards, film == oracle."""
    A = pkg._abi
    sd, rp = _furnace(pkg, 200).world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film = np.zeros((16, 16, 4), np.float32)
    st, msg = _status(gpu, gpu.lib.pt_render, g.h, C.byref(rp), film.ctypes.data_as(C.c_void_p), 0)
    assert st == A.PT_ERR_SOBOL_DIMENSIONS and "1024" in msg
    ofilm = np.zeros((16, 16, 4), np.float32)
    assert orc.O.lib.orc_render(orc.h, C.byref(rp), ofilm.ctypes.data_as(A.fp), 4) == A.PT_ERR_SOBOL_DIMENSIONS
    rp.max_depth = 100
    film = g.render(rp); ofilm = orc.render(rp, nthreads=4)
    np.testing.assert_allclose(film, ofilm, rtol=2e-6, atol=1e-7)
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(("camera_rays", "intersect_tests", "shadow_tests", "triangle_tests", "bvh_nodes_visited", "path_length_hist")):
        assert gc[k] == oc[k], k
    assert gc["path_length_hist"][15] > 0     # (the histogram's last bucket: paths of >= 15 vertices -- they all are)


def _chain_scene(pkg, n):
    """n big triangles stacked along z under an ADOPTED degenerate tree: interior node i = {interior i + 1 (or the last leaf), leaf i}, split axis z. A ray up the z axis descends the
    interior chain first (near child = first child for a positive direction, bvh.rs:745-751) and leaves one pending far leaf per level: n - 1 stack entries."""
    A = pkg._abi
    b = pkg.host.SceneBuilder()
    P = np.zeros((3 * n, 3), np.float32); I = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    for k in range(n):
        z = float(k + 1)
        P[3 * k:3 * k + 3] = [(-1.0, -1.0, z), (3.0, -1.0, z), (-1.0, 3.0, z)]
    b.trianglemesh(P, I)
    sd, _ = b.world_end()
    m = n - 1                                   # interior nodes 0 .. m-1, the last leaf at m, leaf of interior i at 2m - i (pre-order, bvh.rs:662-703)
    nodes = (A.PtBVHNode * (2 * m + 1))()
    ordered = np.zeros(n, np.uint32)

    def box(node, k0, k1):                      # bounds of triangles k0 .. k1
        node.bmin[0], node.bmin[1], node.bmin[2] = -1.0, -1.0, float(k0 + 1)
        node.bmax[0], node.bmax[1], node.bmax[2] = 3.0, 3.0, float(k1 + 1)
    for i in range(m):
        box(nodes[i], i, n - 1); nodes[i].offset = 2 * m - i; nodes[i].n_prims = 0; nodes[i].axis = 2
    box(nodes[m], n - 1, n - 1); nodes[m].offset = 0; nodes[m].n_prims = 1; ordered[0] = n - 1
    for i in range(m):
        leaf = nodes[2 * m - i]
        box(leaf, i, i); leaf.offset = m - i; leaf.n_prims = 1; ordered[m - i] = i
    sd.set_bvh(nodes, ordered)
    return sd


def test_traversal_stack_overflow_is_a_status_in_both_walks(pkg, gpu, oracle):
    """accelerators/bvh.rs:722: the reference's `nodes_to_visit` has 64 entries and no check of its own. An adopted chain tree of depth 40 traces like the oracle (the ray
    meets its 40 triangles far to near and returns the nearest); one of depth 300 needs 299 pending entries: PT_ERR_STACK_OVERFLOW from pt_trace_closest and pt_trace_any
    in the production walk (96 entries) and in the exact walk (64) -- this test runs in both -- and rays that miss the chain still trace on the same handle."""
    A = pkg._abi
    up = (np.array([[0.0, 0.0, 0.0]], np.float32), np.array([[0.0, 0.0, 1.0]], np.float32), np.array([np.inf], np.float32))
    sd = _chain_scene(pkg, 40)
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    gp, gt, gb = g.trace_closest(*up); op, ot, ob = orc.trace_closest(*up)
    assert gp[0] == op[0] == 0 and gt[0] == ot[0] == 1.0 and np.array_equal(gb, ob)
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(("bvh_nodes_visited", "triangle_tests", "intersect_tests")):
        assert gc[k] == oc[k], k
    assert gc["triangle_tests"] == 40
    sd = _chain_scene(pkg, 300)
    g = pkg.Scene(gpu, sd)
    o, d, tmax = (np.ascontiguousarray(x) for x in up)
    prim = np.zeros(1, np.uint32); t = np.zeros(1, np.float32); bb = np.zeros((1, 3), np.float32); hit = np.zeros(1, np.uint8)
    fp = lambda a: a.ctypes.data_as(A.fp)
    st, msg = _status(gpu, gpu.lib.pt_trace_closest, g.h, 1, fp(o), fp(d), fp(tmax), prim.ctypes.data_as(A.u32p), fp(t), fp(bb))
    assert st == A.PT_ERR_STACK_OVERFLOW and "96" in msg and "64" in msg
    down = np.array([[0.0, 0.0, -1.0]], np.float32)     # away from the chain: the root record's boxes are missed, nothing is pushed
    gp, gt, gb = g.trace_closest(o, down, tmax)
    assert gp[0] == 0xFFFFFFFF
    side = (np.array([[0.5, 0.5, 299.5]], np.float32), np.array([[0.0, 0.0, 1.0]], np.float32), np.array([np.inf], np.float32))   # starts above all but the last triangle:
    gp, gt, gb = g.trace_closest(*side)                                                                                         # the lower leaves' boxes lie behind it and are never pushed
    assert gp[0] == 299 and gt[0] == 0.5
    # an any-hit ray up the axis stops at the first triangle it tests -- the LAST leaf, reached with 299 entries pending in the reference: a status here too
    st, msg = _status(gpu, gpu.lib.pt_trace_any, g.h, 1, fp(o), fp(d), fp(tmax), hit.ctypes.data_as(A.u8p))
    assert st == A.PT_ERR_STACK_OVERFLOW


def test_out_of_memory_is_a_status_and_the_scene_stays_usable(pkg, gpu, oracle):
    """`spp_per_pass` is the caller's: 1024 samples of every pixel of a 1080p frame in ONE pass are 2.1e9 paths = 550 GB of path state, more than the device has. pt_render
    returns PT_ERR_OUT_OF_MEMORY (no abort, no partial workspace left behind: the allocation that failed came after the previous workspace had been freed), pt_pass_size
    agrees with pt_render about what it would do, and the same handle then renders the same frame in passes of the library's choosing, film == the oracle's crop."""
    A = pkg._abi
    b = pkg.scenes.ganesha_scale(n=24, xres=1920, yres=1080, spp=1024)
    sd, rp = b.world_end()
    g = pkg.Scene(gpu, sd)
    small = pkg.scenes.ganesha_scale(n=24, xres=64, yres=48, spp=4).world_end()[1]
    ref_small = g.render(small)                                 # a workspace exists before the failing call
    rp.spp_per_pass = 1024
    film = np.zeros((1080, 1920, 4), np.float32)
    st, msg = _status(gpu, gpu.lib.pt_render, g.h, C.byref(rp), film.ctypes.data_as(C.c_void_p), 0)
    assert st == A.PT_ERR_OUT_OF_MEMORY and "path-state slab" in msg
    assert not film.any()
    rp.spp_per_pass = 2048                                      # 2048 x 2 088 960 pixel slots > 2^31 paths: refused before any allocation, by pt_pass_size as by pt_render
    rp.spp = 2048
    s = C.c_uint32()
    st, msg = _status(gpu, gpu.lib.pt_pass_size, g.h, C.byref(rp), C.byref(s))
    assert st == A.PT_ERR_INVALID_ARG and "pass too large" in msg
    st2, _ = _status(gpu, gpu.lib.pt_render, g.h, C.byref(rp), film.ctypes.data_as(C.c_void_p), 0)
    assert st2 == A.PT_ERR_INVALID_ARG
    for bad in ("filter", "tile", "film"):                      # what pt_render refuses, pt_pass_size refuses
        q = pkg.scenes.ganesha_scale(n=24, xres=64, yres=48, spp=4).world_end()[1]
        if bad == "filter": q.filter_radius[0] = 0.0
        if bad == "tile": q.tile_world = 2; q.tile_rank = 2
        if bad == "film": q.cropped_pixel_bounds[2] = q.cropped_pixel_bounds[0]
        dummy = np.zeros((48, 64, 4), np.float32)
        assert gpu.lib.pt_pass_size(g.h, C.byref(q), C.byref(s)) == A.PT_ERR_INVALID_ARG, bad
        assert gpu.lib.pt_render(g.h, C.byref(q), dummy.ctypes.data_as(C.c_void_p), 0) == A.PT_ERR_INVALID_ARG, bad
    again = g.render(small)                                     # the handle is as good as new: same bits as before the failures
    assert np.array_equal(again[..., 3], ref_small[..., 3])
    np.testing.assert_allclose(again, ref_small, rtol=2e-6, atol=1e-7)
    orc = oracle.scene(sd)
    np.testing.assert_allclose(again, orc.render(small, nthreads=4), rtol=2e-6, atol=1e-7)


def test_a_pass_that_does_not_end_is_a_status(pkg, gpu, oracle):
    """PT_ERR_PROBE_CHAIN: the render is abandoned when a pass needs more wavefront iterations than the cap (2^20; a path needs about maxdepth of them -- only the
    reference's own endless loops, e.g. the phantom segments of a rotated disk shell, DESIGN.md section 4 "Shells", get anywhere near). PT_TEST_MAX_ITERATIONS lowers the cap to
    3 for one render of a maxdepth-5 scene: the status comes back, and without the hook the same handle renders film == oracle."""
    A = pkg._abi
    sd, rp = pkg.scenes.ganesha_scale(n=24, xres=64, yres=48, spp=4).world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film = np.zeros((48, 64, 4), np.float32)
    os.environ["PT_TEST_MAX_ITERATIONS"] = "3"
    try:
        st, msg = _status(gpu, gpu.lib.pt_render, g.h, C.byref(rp), film.ctypes.data_as(C.c_void_p), 0)
    finally:
        del os.environ["PT_TEST_MAX_ITERATIONS"]
    assert st == A.PT_ERR_PROBE_CHAIN and "3 wavefront iterations" in msg
    np.testing.assert_allclose(g.render(rp), orc.render(rp, nthreads=4), rtol=2e-6, atol=1e-7)


def test_every_class_of_the_config_scenes_has_a_queue(pkg, gpu):
    """k_film_final takes every path that is neither finished nor dead for an escaped ray, so a HIT that k_route dropped (its class had no queue) would be shaded
    as sky. k_route now raises PT_ERR_UNSUPPORTED for such an entry (kern_misc.h) and pt_render returns it. A scene cannot provoke the status through the ABI -- the host hands every
    class it assigns a queue -- so what is tested is the absence of false alarms on the scene kinds with the most classes: the six-slot router (material zoo: matte, one-, two- and
    many-lobe, specular, metal, plastic-like ...) and the subsurface classes; any status would fail the render."""
    for b in (pkg.scenes.material_zoo(n=12, xres=48, yres=32, spp=2), pkg.scenes.subsurface_c5(n=12, xres=48, yres=32, spp=2),
              pkg.scenes.country_kitchen_s3(xres=48, yres=32, spp=2, wall_n=8, box_n=4, obj_n=8, mixed=True)):
        sd, rp = b.world_end()
        pkg.Scene(gpu, sd).render(rp)
