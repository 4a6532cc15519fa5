"""How the wavefront is scheduled must not change what is computed: one traversal launch per iteration (k_trace<2, ..>: continuation, MIS
and shadow rays together) against one launch per ray kind, any pass size, and the material classes a scene's vertices are shaded by
(scene_create.hip: material_class). The reference has no counterpart of these choices (core/integrator.rs:263-403 is one loop per sample), so
the checks are: identical films and counters between the schedules, and GPU == oracle for the scenes that exercise each class."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
from conftest import ckeys, trace_env

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COUNTERS = ("camera_rays", "intersect_tests", "shadow_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests", "path_length_hist", "film_splats",
            "zero_radiance_paths_num", "zero_radiance_paths_den")

_CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
from _pkg import import_pkg
pkg = import_pkg()
import torch   # before the library touches HIP (tests/conftest.py)
torch.cuda.init()
lib = pkg.load_library(); lib.init(0)
out = {{}}
for name in ("zoo", "spheres", "instances"):
    b = {{"zoo": lambda: pkg.scenes.material_zoo(n=16, xres=96, yres=64, spp=8), "spheres": lambda: pkg.scenes.spheres_c1(xres=64, yres=64, spp=8),
         "instances": lambda: pkg.scenes.instanced_garden(xres=64, yres=48, spp=4)}}[name]()
    sd, rp = b.world_end()
    g = pkg.Scene(lib, sd)
    film = g.render(rp)
    c = g.counters()
    np.save({out!r} + "/" + name + ".npy", film)
    out[name] = dict(counters={{k: c[k] for k in {counters!r}}}, stats=sorted(s["name"] for s in g.kernel_stats() if s["launches"]))
json.dump(out, open({out!r} + "/out.json", "w"))
"""


def _scenes(pkg):
    return {"zoo": lambda: pkg.scenes.material_zoo(n=16, xres=96, yres=64, spp=8), "spheres": lambda: pkg.scenes.spheres_c1(xres=64, yres=64, spp=8),
            "instances": lambda: pkg.scenes.instanced_garden(xres=64, yres=48, spp=4)}


def test_mixed_traversal_launch_equals_one_launch_per_ray_kind(pkg, gpu, tmp_path):
    """PT_TRACE_SPLIT=1 (read by pt_init, so in a process of its own) traces the three ray kinds of an iteration in three launches, as
    round 1 did; the default traces them in one. Same counters, same weights bit for bit, same radiance up to the order of the film's float atomics; the launch kinds differ."""
    env = trace_env(dict(os.environ, PT_TRACE_SPLIT="1"))
    code = _CHILD.format(root=ROOT, out=str(tmp_path), counters=COUNTERS)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    split = json.load(open(tmp_path / "out.json"))
    for name, make in _scenes(pkg).items():
        sd, rp = make().world_end()
        g = pkg.Scene(gpu, sd)
        film = g.render(rp)
        c = g.counters()
        stats = sorted(s["name"] for s in g.kernel_stats() if s["launches"])
        for k in ckeys(COUNTERS):
            assert c[k] == split[name]["counters"][k], (name, k)
        other = np.load(tmp_path / (name + ".npy"))
        assert np.array_equal(film[..., 3], other[..., 3]), name
        np.testing.assert_allclose(film[..., :3], other[..., :3], rtol=2e-6, atol=1e-7, err_msg=name)   # (corner samples reach their neighbours through float atomics: order not defined)
        assert "trace" in stats and not {"extend", "extend_mis", "shadow"} & set(stats), stats
        assert {"extend", "shadow"} <= set(split[name]["stats"]) and "trace" not in split[name]["stats"], split[name]["stats"]


def test_film_kernel_that_ends_the_paths_equals_the_miss_pass(pkg, gpu, tmp_path):
    """Round 5: the plain path integrator's paths are ended by the film kernel (k_film_final: the last vertex's pending estimate, the escaped ray's Le, the histogram entry)
    instead of by a k_shade_miss pass per wavefront iteration. PT_FILM_FINAL=0 (read by pt_init: a process of its own) brings the pass back: same counters (incl. the
    path-length histogram, the zero-radiance count and the reference's asserts), same weights bit for bit, same radiance up to the order of the film's corner-sample atomics;
    only the launch kinds differ."""
    env = trace_env(dict(os.environ, PT_FILM_FINAL="0"))
    code = _CHILD.format(root=ROOT, out=str(tmp_path), counters=COUNTERS)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    old = json.load(open(tmp_path / "out.json"))
    for name, make in _scenes(pkg).items():
        sd, rp = make().world_end()
        g = pkg.Scene(gpu, sd)
        film = g.render(rp)
        c = g.counters()
        st = {s["name"]: s["kernel"] for s in g.kernel_stats() if s["launches"]}
        for k in ckeys(COUNTERS):
            assert c[k] == old[name]["counters"][k], (name, k)
        other = np.load(tmp_path / (name + ".npy"))
        assert np.array_equal(film[..., 3], other[..., 3]), name
        np.testing.assert_allclose(film[..., :3], other[..., :3], rtol=2e-6, atol=1e-7, err_msg=name)
        assert "shade_miss" not in st and st["film"].startswith("k_film_final<"), st
        assert "shade_miss" in old[name]["stats"], old[name]["stats"]


@pytest.mark.parametrize("per_pass", [1, 3, 5])
def test_pass_size_does_not_change_the_image(pkg, gpu, per_pass):
    """spp_per_pass = 0 lets the library size its passes from the free memory (here: one pass); explicit pass sizes that do not divide the
    sample count give the same film (bit-identical weights and, with the box filter's one splat per sample in sample order, radiance)."""
    sd, rp = pkg.scenes.material_zoo(n=16, xres=96, yres=64, spp=8).world_end()
    g = pkg.Scene(gpu, sd)
    rp.spp_per_pass = 0
    a = g.render(rp); ca = g.counters()
    n_pass_auto = [s["launches"] for s in g.kernel_stats() if s["name"] == "generate"][0]
    rp.spp_per_pass = per_pass
    b = g.render(rp); cb = g.counters()
    n_pass = [s["launches"] for s in g.kernel_stats() if s["name"] == "generate"][0]
    assert n_pass_auto == 1 and n_pass == -(-8 // per_pass)
    for k in ckeys(COUNTERS):
        assert ca[k] == cb[k], k
    assert np.array_equal(a[..., 3], b[..., 3])
    np.testing.assert_allclose(a[..., :3], b[..., :3], rtol=1e-6, atol=1e-7)


def test_specular_materials_have_a_shade_class_of_their_own(pkg, gpu, oracle):
    """Mirror and smooth glass (C1's two spheres) are shaded by k_shade<1, MODE, 2>, the kernel without next-event estimation
    (path.rs:131: a BSDF without non-specular components does not sample a light); rough glass and metal are not. GPU == oracle."""
    sd, rp = pkg.scenes.spheres_c1(xres=64, yres=64, spp=8).world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film, ref = g.render(rp), orc.render(rp, nthreads=4)
    st = {s["name"]: s for s in g.kernel_stats() if s["launches"]}
    assert "shade_specular" in st and st["shade_specular"]["kernel"] == "k_shade<1, 1, 2>" and "shade_1lobe" not in st
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(COUNTERS):
        assert gc[k] == oc[k], k
    np.testing.assert_allclose(film, ref, rtol=3e-6, atol=1e-6)
    # the zoo has metal + substrate (one lobe, with NEE), mirror + glass (specular), plastic + rough glass (two lobes) and an uber with a specular term (class 3)
    sd, rp = pkg.scenes.material_zoo(n=16, xres=96, yres=64, spp=4).world_end()
    g = pkg.Scene(gpu, sd); g.render(rp)
    names = {s["name"] for s in g.kernel_stats() if s["launches"]}
    assert {"shade_matte", "shade_1lobe", "shade_2lobe", "shade_uber", "shade_specular", "shade_metal", "shade_plastic"} <= names and "shade_miss" not in names   # (no miss pass: the film kernel ends the plain path integrator's paths)   # (substrate, rough glass: the general kernels; metal, plastic, uber: their own)


def test_lobe_set_specialised_kernels_are_chosen_per_material_and_change_nothing(pkg, gpu, oracle, tmp_path, trace_mode):
    """Round 5: every metal vertex is shaded by k_shade<1, 0, 3>, every plastic-like one by <2, 0, 4>, every uber by <5, 0, 5> -- a class each (kernels.h:
    kMetalClass ...), whatever else the scene holds: the C3 palette, and the same room with substrate, rough glass and translucent surfaces added, whose
    vertices go to the general kernels of their lobe count next to them (round 4 switched per SCENE: one such material sent the whole lobe-count class
    back to the general kernel). Either way GPU == oracle, and PT_SHADE_SPECIALISE=0 (general kernels everywhere, a process of its own: pt_init reads
    it) renders the same film."""
    if trace_mode == "exact":
        pytest.skip("a test of the shade kernels' choice, the same under both walks (the -m gpu suite's time budget)")
    sd, rp = pkg.scenes.country_kitchen_s3(xres=96, yres=64, spp=4, wall_n=6, box_n=3, obj_n=6).world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film, ref = g.render(rp), orc.render(rp, nthreads=4)
    st = {s["name"]: s["kernel"] for s in g.kernel_stats() if s["launches"]}
    assert st["shade_metal"] == "k_shade<1, 0, 3>" and st["shade_plastic"] == "k_shade<2, 0, 4>" and st["shade_uber"] == "k_shade<5, 0, 5>", st
    assert not ({"shade_1lobe", "shade_2lobe", "shade_manylobe"} & set(st)) and st["route"] == "k_route<6, 2048>", st
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(COUNTERS):
        assert gc[k] == oc[k], k
    assert np.array_equal(film[..., 3], ref[..., 3])
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=3e-6, atol=1e-6)
    sd2, rp2 = pkg.scenes.country_kitchen_s3(xres=96, yres=64, spp=4, wall_n=6, box_n=3, obj_n=6, mixed=True).world_end()
    g2 = pkg.Scene(gpu, sd2); orc2 = oracle.scene(sd2)
    film2, ref2 = g2.render(rp2), orc2.render(rp2, nthreads=4)
    st2 = {s["name"]: s["kernel"] for s in g2.kernel_stats() if s["launches"]}
    assert st2["shade_metal"] == "k_shade<1, 0, 3>" and st2["shade_plastic"] == "k_shade<2, 0, 4>" and st2["shade_uber"] == "k_shade<5, 0, 5>", st2
    assert st2["shade_1lobe"] == "k_shade<1, 0, 0>" and st2["shade_2lobe"] == "k_shade<2, 0, 0>" and st2["shade_manylobe"] == "k_shade<5, 0, 0>", st2   # substrate, rough glass, translucent
    assert st2["route"] == "k_route<12, 1024>", st2   # nine shade classes + the miss class
    items = {s["name"]: s["items"] for s in g2.kernel_stats() if s["launches"]}
    assert all(items[k] > 0 for k in ("shade_metal", "shade_plastic", "shade_uber", "shade_1lobe", "shade_2lobe", "shade_manylobe", "shade_matte", "shade_specular")), items
    gc, oc = g2.counters(), orc2.counters()
    for k in ckeys(COUNTERS):
        assert gc[k] == oc[k], k
    assert np.array_equal(film2[..., 3], ref2[..., 3])
    np.testing.assert_allclose(film2[..., :3], ref2[..., :3], rtol=3e-6, atol=1e-6)
    code = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from _pkg import import_pkg
pkg = import_pkg()
import torch
torch.cuda.init()
lib = pkg.load_library(); lib.init(0)
sd, rp = pkg.scenes.country_kitchen_s3(xres=96, yres=64, spp=4, wall_n=6, box_n=3, obj_n=6).world_end()
g = pkg.Scene(lib, sd); film = g.render(rp)
ks = {{s["name"]: s["kernel"] for s in g.kernel_stats() if s["launches"]}}
assert ks["shade_1lobe"] == "k_shade<1, 0, 0>" and ks["shade_2lobe"] == "k_shade<2, 0, 0>" and ks["shade_manylobe"] == "k_shade<5, 0, 0>" and "shade_metal" not in ks, ks
np.save({out!r}, film)
""".format(root=ROOT, out=str(tmp_path / "general.npy"))
    r = subprocess.run([sys.executable, "-c", code], env=trace_env(dict(os.environ, PT_SHADE_SPECIALISE="0")), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    other = np.load(tmp_path / "general.npy")
    assert np.array_equal(film[..., 3], other[..., 3])
    np.testing.assert_allclose(film[..., :3], other[..., :3], rtol=2e-6, atol=1e-7)


def _uber_ball(pkg, uber):
    """A displaced ball of the given uber material over a matte floor, one area light and a dim constant environment."""
    S = pkg.scenes
    b = S.SceneBuilder()
    b.film.update(xres=64, yres=48); b.spp = 8; b.integ.update(maxdepth=5)
    b.look_at((0.0, 1.5, 5.0), (0.0, 0.4, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=40.0)
    b.world_begin()
    b.light_source("infinite", L=(0.2, 0.25, 0.3))
    b.attribute_begin(); b.area_light_source(L=(20.0, 18.0, 15.0))
    P, I = S.quad((-1.5, 4.0, -1.5), (1.5, 4.0, -1.5), (1.5, 4.0, 1.5), (-1.5, 4.0, 1.5)); b.trianglemesh(P, I); b.attribute_end()
    b.material("matte", Kd=(0.6, 0.6, 0.55))
    P, I = S.quad((-8.0, -0.5, -8.0), (-8.0, -0.5, 8.0), (8.0, -0.5, 8.0), (8.0, -0.5, -8.0)); b.trianglemesh(P, I)
    b.attribute_begin(); b.material("uber", **uber); b.translate(0.0, 0.5, 0.0)
    P, I, N = S.displaced_sphere(12); b.trianglemesh(P, I, N=N); b.attribute_end()
    return b


@pytest.mark.parametrize("uber,expect_class", [(dict(Kd=(0.3, 0.5, 0.2), Ks=(0.3, 0.3, 0.3), roughness=0.2), "shade_plastic"),
                                               (dict(Kd=(0.3, 0.5, 0.2), Ks=(0.3, 0.3, 0.3), Kr=(0.1, 0.1, 0.1), roughness=0.2), "shade_uber"),
                                               (dict(Kd=(0.3, 0.5, 0.2), Ks=(0.3, 0.3, 0.3), opacity=(0.6, 0.6, 0.6), roughness=0.2), "shade_uber")])
def test_uber_without_specular_terms_is_a_two_lobe_material(pkg, gpu, oracle, uber, expect_class):
    """uber.rs:40-106 adds its specular reflection / transmission lobes only for non-black Kr / Kt and its pass-through lobe only for
    opacity < 1: an opaque uber with Kr = Kt = 0 is Lambertian + microfacet and is shaded by the plastic-like two-lobe kernel. GPU == oracle either way."""
    sd, rp = _uber_ball(pkg, uber).world_end()
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film, ref = g.render(rp), orc.render(rp, nthreads=4)
    names = {s["name"] for s in g.kernel_stats() if s["launches"]}
    assert expect_class in names and ({"shade_plastic", "shade_uber", "shade_2lobe", "shade_manylobe"} - {expect_class}).isdisjoint(names), names
    gc, oc = g.counters(), orc.counters()
    for k in ckeys(COUNTERS):
        assert gc[k] == oc[k], k
    np.testing.assert_allclose(film, ref, rtol=3e-6, atol=1e-6)


@pytest.mark.parametrize("knobs", [
    {"PT_TRACE_SPLIT": "1"},
    {"PT_TRACE_LEAF_QUORUM": "1", "PT_TRACE_REFILL_MIN": "4"},
    {"PT_TRACE_LEAF_QUORUM": "40", "PT_TRACE_REFILL_MIN": "64"},
    {"PT_TRACE_INST_QUORUM": "48", "PT_TRACE_LEAF_QUORUM": "40"},
    {"PT_TRACE_INST_QUORUM": "1", "PT_TRACE_REFILL_MIN": "4", "PT_TRACE_SPLIT": "1"},
], ids=lambda k: ",".join(f"{a[9:].lower()}={b}" for a, b in k.items()))
def test_golden_vectors_hold_at_the_scheduling_knobs_extremes(knobs):
    """the traversal's scheduling knobs (read once by pt_init, hence a fresh CHILD process per setting -- started with
    subprocess, never an exec of this process) must not change a single counter or weight: the committed golden vectors (tests/golden,
    every §8 row) through the HIP path at the extremes of leaf quorum / refill batch / instance quorum and with one launch per ray kind."""
    env = dict(os.environ, **knobs)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_golden.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (knobs, r.stdout[-3000:], r.stderr[-1500:])
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]
