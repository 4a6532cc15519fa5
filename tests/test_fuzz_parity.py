"""Randomised parity: scenes assembled from every supported feature with seeded random parameters, rendered by the HIP path and
by the oracle. Exact work counters and 2e-6 relative radiance, as in the hand-written parity tests; the point is the corners
nobody wrote a scene for (odd parameter combinations, partial spheres under non-uniform transforms, lights inside media, ...)."""
import numpy as np
import pytest
from conftest import ckeys


def random_scene(pkg, seed, builder=None):
    S = pkg.scenes
    rng = np.random.default_rng(seed)
    u = lambda a=0.0, b=1.0: float(rng.uniform(a, b))
    rgb = lambda a=0.05, b=0.95: tuple(float(x) for x in rng.uniform(a, b, 3))
    pick = lambda *xs: xs[int(rng.integers(0, len(xs)))]
    b = builder if builder is not None else pkg.host.SceneBuilder()
    b.film.update(xres=int(pick(48, 56, 64)), yres=int(pick(32, 40)))
    b.spp = int(pick(2, 4, 6))
    b.sampler = pick("sobol", "halton")
    b.split_method = pick("sah", "sah", "hlbvh")
    b.max_node_prims = int(pick(1, 2, 4, 6))
    b.filter.update(kind=pick("box", "gaussian", "triangle", "mitchell", "sinc"), radius=(u(0.5, 2.0), u(0.5, 2.0)))
    if rng.random() < 0.25: x0, y0 = u(0.0, 0.4), u(0.0, 0.4); b.film.update(crop=(x0, x0 + u(0.3, 0.6), y0, y0 + u(0.3, 0.6)))
    if rng.random() < 0.15: b.sample_at_pixel_center = True
    volpath = rng.random() < 0.4
    side = np.random.default_rng(seed + 770077)   # (round 3 additions draw from a stream of their own: the scenes of the old seeds keep everything else)
    grid_fog = False
    b.integ.update(maxdepth=int(pick(1, 3, 5, 8)), rrthreshold=pick(1.0, 0.3, 0.0), strategy=pick("spatial", "power", "uniform"), kind="volpath" if volpath else "path")
    if volpath:
        if rng.random() < 0.35:   # the fog as a GridDensityMedium (media/grid.rs; spectrally uniform sigma_t as it requires), shallow paths: ratio / delta tracking draw many dimensions
            nd = int(pick(2, 3, 5)); sa, ss = u(0.0, 0.06), u(0.03, 0.25); grid_fog = True
            b.integ.update(maxdepth=int(pick(1, 2, 3)))
            b.make_named_medium("fog", sigma_a=(sa, sa, sa), sigma_s=(ss, ss, ss), g=u(-0.7, 0.7), density=rng.uniform(0.0, 1.0, (nd, int(pick(2, 4)), nd)).astype(np.float32),
                                p0=(u(-8, -4), u(-2, -0.5), u(-8, -4)), p1=(u(4, 8), u(3, 6), u(4, 8)))
        else:
            b.make_named_medium("fog", sigma_a=rgb(0.0, 0.1), sigma_s=rgb(0.02, 0.3), g=u(-0.7, 0.7))
        b.make_named_medium("ink", sigma_a=rgb(0.1, 2.0), sigma_s=rgb(0.1, 2.0), g=u(-0.3, 0.3), scale=u(0.5, 2.0))
        if rng.random() < 0.7: b.medium_interface("", "fog")
    b.look_at((u(-1, 1), u(1.0, 2.5), u(4.5, 6.0)), (u(-0.3, 0.3), u(0.2, 0.8), 0.0), (0.0, 1.0, 0.0))
    b.camera(fov=u(30, 50), lensradius=pick(0.0, 0.0, u(0.01, 0.08)), focaldistance=u(4.0, 6.0))
    b.world_begin()
    # lights
    if rng.random() < 0.7: b.light_source("infinite", L=rgb(0.02, 0.5))
    if rng.random() < 0.3: b.light_source("infinite", L=rgb(0.2, 1.0), texels=S.sky_env(*pick((16, 8), (16, 8), (32, 4), (4, 16))), scale=u(0.2, 1.0))
    for _ in range(int(rng.integers(0, 3))):
        k = pick("point", "spot", "distant")
        # point lights sit at (x, y, x) (create_pointlight translates by (P.x, P.y, P.x), App. A #15)
        if k == "point": x = u(-3, 3); b.light_source("point", from_=(x, u(1.5, 4), x), I=rgb(2, 15))
        elif k == "spot": b.light_source("spot", from_=(u(-3, 3), u(2, 4), u(-1, 3)), to=(u(-1, 1), 0.0, u(-1, 1)), I=rgb(5, 40), coneangle=u(15, 50), conedeltaangle=u(2, 10))
        else: b.light_source("distant", from_=(u(-3, 3), u(2, 5), u(-3, 3)), to=(0.0, 0.0, 0.0), L=rgb(0.3, 2.0))
    # Disk::intersect divides by the WORLD ray's d.z (disk.rs:65): under a transform that does not keep z the reported hit is off
    # the ray, and a BSSRDF probe chain (bssrdf.rs:376-394) through such a disk need not make progress -- chains of > 10^5
    # segments were seen (minutes of oracle time). Scenes with subsurface materials keep their disks z-aligned.
    sss_ok = (not volpath) and rng.random() < 0.6
    # seed >= 95000: subsurface materials under the volumetric integrator too (volpath.rs:186-214; not next to a grid medium or material-less shells)
    vol_sss = bool(seed >= 95000 and volpath and (not grid_fog or seed >= 140000) and side.random() < 0.6)   # >= 140000: next to grid media and shells too (k_bssrdf's stage B)
    sss_ok = sss_ok or vol_sss
    b.attribute_begin(); b.area_light_source(L=rgb(5, 25), twosided=bool(rng.random() < 0.4))
    if rng.random() < 0.5:
        P, I = S.quad((-0.8, 3.5, -0.8), (0.8, 3.5, -0.8), (0.8, 3.5, 0.8), (-0.8, 3.5, 0.8)); b.trianglemesh(P, I)
    elif rng.random() < 0.5:
        b.translate(u(-1, 1), 3.2, u(-1, 1)); b.sphere(radius=u(0.15, 0.4))
    elif sss_ok:   # a disk light whose transform keeps z (see sss_ok above): in front of the scene, reversed so that it emits to -z
        b.translate(u(-1, 1), u(1.5, 2.5), 4.0); b.rotate(u(0, 360), 0.0, 0.0, 1.0); b.toggle_reverse_orientation(); b.disk(radius=u(0.4, 0.9), innerradius=pick(0.0, u(0.0, 0.3)), phimax=pick(360.0, u(200, 360)))
    else:   # a disk light facing down (disk.rs): rotate its +z normal towards -y
        b.translate(u(-1, 1), 3.4, u(-1, 1)); b.rotate(90.0 + u(-20, 20), 1.0, 0.0, 0.0); b.disk(radius=u(0.4, 0.9), innerradius=pick(0.0, u(0.0, 0.3)), phimax=pick(360.0, u(200, 360)))
    b.attribute_end()
    # textures
    b.texture("chk", "spectrum", "checkerboard", uscale=u(2, 8), vscale=u(2, 8), tex1=rgb(), tex2=rgb(), aamode=pick("none", "closedform"))
    b.texture("img", "spectrum", "imagemap", pixels=S.test_image(12, 10, seed=int(rng.integers(1, 99))), uscale=u(1, 4), vscale=u(1, 4), trilinear=bool(rng.random() < 0.5))
    b.texture("fchk", "float", "checkerboard", uscale=u(2, 6), vscale=u(2, 6), tex1=u(0.02, 0.2), tex2=u(0.2, 0.8))
    b.texture("bump", "float", "checkerboard", uscale=u(4, 9), vscale=u(4, 9), tex1=u(0.0, 0.03), tex2=0.0)
    # the rest of the texture zoo (texture space shifted into the positive octant: the noise lattice index of a negative
    # coordinate saturates in the reference, see tests/test_gpu_parity.py::test_textures_match_oracle)
    b.transform_begin(); b.translate(-20.0, -20.0, -20.0); b.scale(u(0.5, 2.0), u(0.5, 2.0), u(0.5, 2.0))
    b.texture("fbm", "float", "fbm", octaves=int(pick(2, 4, 6)), roughness=u(0.3, 0.7))
    b.texture("wrk", "spectrum", "wrinkled", octaves=int(pick(2, 5)), roughness=u(0.3, 0.7))
    b.texture("mrb", "spectrum", "marble", octaves=int(pick(3, 6)), roughness=u(0.3, 0.7), scale=u(0.5, 2.0), variation=u(0.1, 0.4))
    b.texture("wnd", "float", "windy")
    b.texture("chk3", "spectrum", "checkerboard", dimension=3, tex1=rgb(), tex2=rgb())
    b.transform_end()
    mapping = lambda: pick(dict(), dict(mapping="planar", v1=(u(0.2, 1), 0, u(0, 0.5)), v2=(0, u(0.2, 1), u(0, 0.5)), udelta=u(), vdelta=u()), dict(mapping="spherical"), dict(mapping="cylindrical"))
    b.texture("dots", "spectrum", "dots", uscale=u(2, 6), vscale=u(2, 6), inside=rgb(), outside=pick(rgb(), "chk"), **mapping())
    b.texture("uvt", "spectrum", "uv", uscale=u(1, 3), vscale=u(1, 3))
    b.texture("bil", "spectrum", "bilerp", v00=rgb(), v01=rgb(), v10=rgb(), v11=rgb(), **mapping())
    b.texture("mixt", "spectrum", "mix", tex1=pick(rgb(), "img"), tex2=pick(rgb(), "wrk"), amount=pick(u(), "fbm"))
    b.texture("sclt", "spectrum", "scale", tex1=pick("chk", "mrb", "bil"), tex2=rgb(0.3, 1.0))
    b.texture("img2", "spectrum", "imagemap", pixels=S.test_image(9, 7, seed=int(rng.integers(1, 99))), wrap=pick("repeat", "black"), gamma=bool(rng.random() < 0.3), scale=u(0.5, 1.5), maxanisotropy=pick(2.0, 8.0, 16.0), **mapping())
    b.texture("holes", "float", "checkerboard", uscale=u(2, 5), vscale=u(2, 5), tex1=1.0, tex2=0.0)
    col = lambda: pick(rgb(), rgb(), "chk", "img", "dots", "uvt", "bil", "mixt", "sclt", "img2", "wrk", "mrb", "chk3")
    # Later additions to the generator are gated on the seed and draw from a generator of their own, so that the scene of every earlier
    # seed (the regression seeds above all) stays what it was. seed >= 50000: subsurface materials with textured sigma_a / sigma_s.
    gen2 = seed >= 50000
    if gen2:
        rng2 = np.random.default_rng(seed + 0x5eed)
        rgb2 = lambda a, b2: tuple(float(x) for x in rng2.uniform(a, b2, 3))
        b.texture("siga", "spectrum", "checkerboard", dimension=3, tex1=rgb2(0.001, 0.02), tex2=rgb2(0.001, 0.02))
        b.texture("sigs", "spectrum", "scale", tex1="chk", tex2=rgb2(2.0, 5.0))
    def random_material(allow_mix=True, allow_sss=sss_ok):
        kinds = ["matte", "mirror", "glass", "glass_rough", "plastic", "metal", "uber", "substrate", "translucent", "disney", "disney_thin"]
        if allow_mix: kinds.append("mix")
        if allow_sss: kinds += ["subsurface", "kdsubsurface", "disney_sss"]
        if allow_sss and gen2: kinds += ["subsurface_tex", "subsurface_tex"]
        if allow_sss and seed >= 70000: kinds += ["kdsubsurface_tex", "kdsubsurface_tex"]   # kdsubsurface with a textured Kd / mfp (converted at every hit)
        k = pick(*kinds)
        bump = {"bumpmap": "bump"} if rng.random() < 0.25 else {}
        if rng.random() < 0.1: bump = {"bumpmap": pick("fbm", "wnd")}
        if k == "matte": b.material("matte", Kd=col(), sigma=pick(0.0, u(5, 40)), **bump)
        elif k == "mirror": b.material("mirror", Kr=rgb(0.5, 1.0))
        elif k == "glass": b.material("glass", Kr=rgb(0.5, 1.0), Kt=rgb(0.5, 1.0), eta=u(1.2, 1.8))
        elif k == "glass_rough": b.material("glass", Kr=rgb(0.5, 1.0), Kt=rgb(0.5, 1.0), eta=u(1.2, 1.8), uroughness=u(0.05, 0.5), vroughness=u(0.05, 0.5), remaproughness=bool(rng.random() < 0.5))
        elif k == "plastic": b.material("plastic", Kd=col(), Ks=rgb(0.1, 0.6), roughness=pick(u(0.02, 0.5), "fchk"), **bump)
        elif k == "metal": b.material("metal", roughness=u(0.005, 0.3), **({"uroughness": u(0.01, 0.4), "vroughness": u(0.01, 0.4)} if rng.random() < 0.5 else {}))
        elif k == "uber": b.material("uber", Kd=col(), Ks=rgb(0.0, 0.5), Kr=pick((0.0,) * 3, rgb(0.0, 0.4)), Kt=pick((0.0,) * 3, rgb(0.0, 0.4)), opacity=pick(1.0, rgb(0.3, 1.0)), roughness=u(0.02, 0.4), eta=u(1.1, 1.7))
        elif k == "substrate": b.material("substrate", Kd=col(), Ks=rgb(0.1, 0.6), uroughness=u(0.02, 0.4), vroughness=u(0.02, 0.4))
        elif k == "translucent": b.material("translucent", Kd=col(), Ks=rgb(0.0, 0.5), reflect=rgb(0.0, 0.8), transmit=rgb(0.0, 0.8), roughness=u(0.05, 0.5))
        elif k == "disney":
            extras = list(rng.permutation(["sheen", "clearcoat", "spectrans"]))[:int(rng.integers(0, 3))]   # at most 3 + 2 of these = five BxDFs
            b.material("disney", color=col(), metallic=pick(0.0, u()), roughness=u(0.1, 0.9), speculartint=u(), anisotropic=pick(0.0, u()), sheen=u() if "sheen" in extras else 0.0, sheentint=u(),
                       clearcoat=u() if "clearcoat" in extras else 0.0, clearcoatgloss=u(), spectrans=u() if "spectrans" in extras else 0.0, eta=u(1.2, 1.7))
        elif k == "disney_thin": b.material("disney", color=col(), thin=True, flatness=u(), difftrans=u(0, 2), roughness=u(0.1, 0.9))
        elif k == "disney_sss": b.material("disney", color=rgb(0.3, 0.9), scatterdistance=rgb(0.02, 0.2), roughness=u(0.2, 0.8), eta=u(1.2, 1.6))
        elif k == "subsurface": b.material("subsurface", name=pick("", "Skin1", "Marble"), scale=u(5, 40), eta=u(1.2, 1.5), **({} if rng.random() < 0.5 else {"sigma_a": rgb(0.001, 0.02), "sigma_s": rgb(1, 4)}))
        elif k == "subsurface_tex": b.material("subsurface", sigma_a=pick("siga", rgb(0.001, 0.02)), sigma_s=pick("sigs", "sigs", rgb(1, 4)), scale=u(5, 40), eta=u(1.2, 1.5), **({} if rng.random() < 0.5 else dict(uroughness=u(0.05, 0.3), vroughness=u(0.05, 0.3))))
        elif k == "kdsubsurface_tex": b.material("kdsubsurface", Kd=pick("chk", "img", "chk3", rgb(0.3, 0.9)), mfp=pick("sigs", rgb(0.05, 0.5)) if rng.random() < 0.5 else "sigs", eta=u(1.2, 1.5), scale=u(0.1, 0.4))
        elif k == "kdsubsurface": b.material("kdsubsurface", Kd=rgb(0.3, 0.9), mfp=u(0.05, 0.5), eta=u(1.2, 1.5))
        else:
            ids = []
            for _ in range(2):
                while True:
                    random_material(allow_mix=False, allow_sss=False)
                    m = b.materials[b.material_id]
                    lobes = {pkg._abi.PT_MAT_GLASS: 2, pkg._abi.PT_MAT_PLASTIC: 2, pkg._abi.PT_MAT_UBER: 5, pkg._abi.PT_MAT_TRANSLUCENT: 4, pkg._abi.PT_MAT_DISNEY: 5}.get(m.type, 1)
                    if lobes <= 2: break
                ids.append(b.material_id)
            b.material("mix", amount=pick(u(), rgb(), "chk"), namedmaterial1=ids[0], namedmaterial2=ids[1])
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32)
    random_material(); P, I = S.quad((-9.0, -0.5, -9.0), (-9.0, -0.5, 9.0), (9.0, -0.5, 9.0), (9.0, -0.5, -9.0)); b.trianglemesh(P, I, UV=uv * 6)
    if rng.random() < 0.6:
        b.object_begin("thing"); random_material(allow_sss=False)
        Pm, Im, Nm = S.displaced_sphere(int(pick(4, 6)), with_normals=bool(rng.random() < 0.5)); b.trianglemesh(Pm * np.float32(0.4), Im, N=Nm)
        b.object_end()
        for _ in range(int(rng.integers(1, 4))):
            b.attribute_begin(); b.translate(u(-2.5, 2.5), u(-0.1, 0.6), u(-2, 1)); b.rotate(u(0, 360), 0, 1, 0); b.scale(u(0.6, 1.5), u(0.6, 1.5), u(0.6, 1.5)); b.object_instance("thing"); b.attribute_end()
    for _ in range(int(rng.integers(2, 6))):
        b.attribute_begin()
        interface = bool(volpath and rng.random() < 0.4)
        if interface: b.medium_interface("ink", "fog" if b.camera_medium is not None else "")
        random_material()
        shell = bool(seed >= 90000 and interface and (not vol_sss or seed >= 140000) and side.random() < 0.6)   # a material-less shell around the ink (api.rs:597): volpath walks its shadow / MIS rays through it
        b.translate(u(-2.5, 2.5), u(-0.1, 0.8), u(-2.0, 1.5))
        if rng.random() < 0.3: b.toggle_reverse_orientation()
        shape = pick("sphere", "partial", "mesh", "quad", "disk")
        # (not a disk: Disk::intersect divides by the WORLD ray's d.z, disk.rs:66, so a shadow ray that leaves a rotated disk shell "hits" it again
        #  at t ~ 1e-8, creeping along the disk in ~10^5 phantom segments -- in the reference and the oracle too; on the device each is a wavefront iteration)
        if shell and shape != "disk": b.material("none")
        if shape == "sphere": b.sphere(radius=u(0.3, 0.7))
        elif shape == "partial": b.rotate(u(0, 360), u(-1, 1), 1.0, u(-1, 1)); b.scale(u(0.7, 1.3), u(0.7, 1.3), u(0.7, 1.3)); r = u(0.3, 0.7); b.sphere(radius=r, zmin=-r * u(0.2, 1.0), zmax=r * u(0.2, 1.0), phimax=u(120, 360))
        elif shape == "disk":
            if sss_ok: b.rotate(u(0, 360), 0.0, 0.0, 1.0)
            else: b.rotate(u(0, 360), u(-1, 1), 1.0, u(-1, 1))
            b.scale(u(0.7, 1.3), u(0.7, 1.3), 1.0); b.disk(height=u(-0.2, 0.2), radius=u(0.4, 0.9), innerradius=pick(0.0, u(0.05, 0.3)), phimax=pick(360.0, u(90, 360)))
        elif shape == "mesh":
            Pm, Im, Nm = S.displaced_sphere(int(pick(4, 8)), with_normals=bool(rng.random() < 0.5)); b.trianglemesh(Pm * np.float32(u(0.3, 0.6)), Im, N=Nm)
        else:
            b.rotate(u(-60, 60), 1.0, u(-1, 1), 0.0); h = u(0.4, 0.9); P, I = S.quad((-h, 0, 0), (h, 0, 0), (h, 2 * h, 0), (-h, 2 * h, 0))
            masks = pick(dict(), dict(), dict(alpha="holes"), dict(alpha="holes", shadowalpha="holes"), dict(shadowalpha=0.0))
            b.trianglemesh(P, I, UV=uv, **masks)
        b.attribute_end()
    return b


@pytest.mark.parametrize("seed", range(4))
def test_oracle_renders_random_scenes(pkg, oracle, seed):
    """CPU smoke of the generator: finite films, every path accounted for."""
    sd, rp = random_scene(pkg, seed).world_end()
    s = oracle.scene(sd)
    film = s.render(rp, nthreads=4)
    c = s.counters()
    assert np.isfinite(film[..., 3]).all() and sum(c["path_length_hist"]) == c["camera_rays"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(160)) + [2005, 13269] + list(range(50000, 50024)) + list(range(70000, 70016)) + list(range(90000, 90096)) + list(range(95000, 95060)) + list(range(140000, 140060)))   # >= 90000: + material-less medium shells under volpath (28 of the 96); >= 95000: + subsurface materials under volpath (10 of the 60); >= 140000: those next to grid media and shells as well   # found by a 14 000-seed sweep (tools/fuzz_sweep.py): 2005 volpath's `if beta.is_black() { break }` outside any medium; 13269 `L += beta * Ld` with a black Ld and a NaN beta (BSSRDF exit point a few ulps from the entry point)
def test_gpu_matches_oracle_on_random_scenes(pkg, gpu, oracle, seed, trace_mode):
    if trace_mode == "exact" and seed % 3 != 0 and seed not in (2005, 13269):
        pytest.skip("the exact (two-wide) walk runs every third seed: the shading code under test is the same in both walks, and the -m gpu suite has a time budget (round 5)")
    b = random_scene(pkg, seed)
    sd, rp = b.world_end()
    if seed >= 90000 and seed % 3 == 0 and rp.light_strategy == pkg._abi.PT_LS_SPATIAL: rp.light_strategy = pkg._abi.PT_LS_SPATIAL_LAZY   # the first-touch form of the light grid (the same distribution: the oracle does not care)
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    film, ref = g.render(rp), orc.render(rp, nthreads=8)
    gc, oc = g.counters(), orc.counters()
    sss = any(m.type == pkg._abi.PT_MAT_SUBSURFACE or (m.type == pkg._abi.PT_MAT_DISNEY and any(m.disney_scatter)) for m in b.materials)
    exact = ("camera_rays", "shadow_tests", "path_length_hist", "film_splats", "sanitized_nan", "sanitized_negative", "sanitized_infinite",
             "intersect_tests", "bvh_nodes_visited", "triangle_tests", "sphere_tests", "reference_asserts")   # reference_asserts: the assert!()s of path.rs:143,162-163,184,201,213 / volpath.rs:176,194,210,223 that would have fired; BSSRDF probe chains included: walked once, inside k_trace<.., PROBE>
    for k in ckeys(exact): assert gc[k] == oc[k], (k, gc[k], oc[k])
    # filter-weight sums: exact for the box filter, float summation order otherwise
    if b.filter["kind"] == "box" and max(b.filter["radius"]) <= 0.5: assert np.array_equal(film[..., 3], ref[..., 3])
    else: np.testing.assert_allclose(film[..., 3], ref[..., 3], rtol=2e-6)
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=3e-5, atol=2e-6)


@pytest.mark.parametrize("seed", list(range(40)) + [90005, 90007, 90038, 90041, 95000, 95054])
def test_front_end_twin_of_random_scenes(pkg, oracle, tmp_path, seed):
    """The same seeded scene through the Python mirror of api.rs and -- as .pbrt text written by tests/pbrt_recorder.py -- through
    the C++ front end: identical structure and parameters, and the same image up to the ulp differences of the two hosts'
    float arithmetic (matrix inverses, MIP pyramids)."""
    from pbrt_recorder import make_recorder
    rec = random_scene(pkg, seed, builder=make_recorder(pkg, str(tmp_path)))
    sd, rp = rec.world_end()
    fs = pkg.frontend.FrontScene(text=rec.text(), base_dir=str(tmp_path))
    d, d2, rp2 = sd.desc(), fs.desc(), fs.render_params()
    unused_tex = d2.n_textures == 0 and d.n_textures > 0   # the front end uploads no texture table when nothing refers to one
    if unused_tex: assert all(t < 0 for i in range(d.n_materials) for t in d.materials[i].tex) and not d.tri_alpha
    for f in ("n_vertices", "n_triangles", "n_spheres", "n_prims", "n_lights", "n_materials", "n_media", "n_objects", "n_instances", "n_top", "max_node_prims", "split_method", "env_width", "env_height"):
        assert getattr(d, f) == getattr(d2, f), f
    for f in ("cropped_pixel_bounds", "sample_bounds", "pixel_bounds", "full_resolution"): assert list(getattr(rp, f)) == list(getattr(rp2, f)), f
    for f in ("spp", "max_depth", "light_strategy", "sampler_type", "sample_at_pixel_center", "integrator", "camera_medium"): assert getattr(rp, f) == getattr(rp2, f), f
    assert np.allclose(list(rp.filter_table), list(rp2.filter_table), rtol=1e-5, atol=1e-7) and rp.rr_threshold == pytest.approx(rp2.rr_threshold)
    arr = lambda p, n, dt=None: np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0)
    for f, n in (("indices", 3 * d.n_triangles), ("tri_flags", d.n_triangles), ("prim_shape", d.n_prims), ("prim_material", d.n_prims), ("prim_light", d.n_prims)):
        assert np.array_equal(arr(getattr(d, f), n), arr(getattr(d2, f), n)), f
    assert np.allclose(arr(d.P, 3 * d.n_vertices), arr(d2.P, 3 * d2.n_vertices), atol=2e-5)
    if d.n_media: assert np.array_equal(arr(d.prim_medium_inside, d.n_prims), arr(d2.prim_medium_inside, d.n_prims)) and np.array_equal(arr(d.prim_medium_outside, d.n_prims), arr(d2.prim_medium_outside, d.n_prims))
    A = pkg._abi
    used = {A.PT_MAT_MATTE: ("kd", "sigma"), A.PT_MAT_MIRROR: ("kr",), A.PT_MAT_GLASS: ("kr", "kt", "eta", "u_roughness", "v_roughness"),
            A.PT_MAT_PLASTIC: ("kd", "ks", "roughness"), A.PT_MAT_METAL: ("eta_rgb", "k_rgb", "roughness", "u_roughness", "v_roughness"),
            A.PT_MAT_UBER: ("kd", "ks", "kr", "kt", "opacity", "roughness", "u_roughness", "v_roughness", "eta"), A.PT_MAT_SUBSTRATE: ("kd", "ks", "u_roughness", "v_roughness"),
            A.PT_MAT_SUBSURFACE: ("kr", "kt", "sigma_a", "sigma_s", "scale", "eta", "u_roughness", "v_roughness"), A.PT_MAT_TRANSLUCENT: ("kd", "ks", "kr", "kt", "roughness"),
            A.PT_MAT_MIX: ("kd",), A.PT_MAT_DISNEY: ("kd", "eta", "roughness", "disney", "disney_scatter")}
    slot = dict(kd=A.PT_MP_KD, ks=A.PT_MP_KS, kr=A.PT_MP_KR, kt=A.PT_MP_KT, opacity=A.PT_MP_OPACITY, eta_rgb=A.PT_MP_ETA_RGB, k_rgb=A.PT_MP_K_RGB, sigma=A.PT_MP_SIGMA,
                roughness=A.PT_MP_ROUGHNESS, u_roughness=A.PT_MP_U_ROUGHNESS, v_roughness=A.PT_MP_V_ROUGHNESS, eta=A.PT_MP_ETA)
    for i in range(d.n_materials):
        a, b = d.materials[i], d2.materials[i]
        assert a.type == b.type and list(a.tex) == list(b.tex) and a.remap_roughness == b.remap_roughness, i
        if a.type == A.PT_MAT_MIX: assert list(a.mix) == list(b.mix)
        if a.type == A.PT_MAT_DISNEY: assert a.disney_thin == b.disney_thin
        for f in used[a.type]:
            if f in slot and a.tex[slot[f]] >= 0: continue   # textured: the constant field is not read
            va, vb = getattr(a, f), getattr(b, f)
            va, vb = (list(va), list(vb)) if hasattr(va, "__len__") else ([va], [vb])
            assert np.allclose(va, vb, rtol=2e-5, atol=1e-7), (i, f, va, vb)
    if not unused_tex: assert (d.n_textures, d.n_images) == (d2.n_textures, d2.n_images)
    for i in range(0 if unused_tex else d.n_textures):
        a, b = d.textures[i], d2.textures[i]
        assert (a.type, list(a.child), a.mapping, a.aa_closedform, a.image, a.trilinear, a.wrap, a.octaves) == (b.type, list(b.child), b.mapping, b.aa_closedform, b.image, b.trilinear, b.wrap, b.octaves), i
        for f in ("value", "v00", "v01", "v10", "v11", "vs", "vt"): assert np.allclose(list(getattr(a, f)), list(getattr(b, f)), rtol=1e-6), (i, f)
        for f in ("su", "sv", "du", "dv", "max_anisotropy", "omega", "marble_scale", "variation"): assert getattr(a, f) == pytest.approx(getattr(b, f), rel=1e-6), (i, f)
    for i in range(d.n_lights):
        assert d.lights[i].type == d2.lights[i].type and d.lights[i].two_sided == d2.lights[i].two_sided
        assert np.allclose(list(d.lights[i].L), list(d2.lights[i].L), rtol=1e-5) and np.allclose(list(d.lights[i].pos), list(d2.lights[i].pos), atol=2e-5)
    for i in range(d.n_media): assert np.allclose(list(d.media[i].sigma_a) + list(d.media[i].sigma_s) + [d.media[i].g], list(d2.media[i].sigma_a) + list(d2.media[i].sigma_s) + [d2.media[i].g], rtol=1e-6)
    a = oracle.scene(sd); b = oracle.scene(fs)
    ia = a.resolve(a.render(rp, nthreads=4), scale=rp.scale); ib = b.resolve(b.render(rp2, nthreads=4), scale=rp2.scale)
    ok = np.isfinite(ia).all(axis=2) & np.isfinite(ib).all(axis=2)
    diff = np.abs(ia - ib)[ok]; ref = np.maximum(np.abs(ia)[ok], 1e-2)
    assert (np.max(diff / ref, axis=1) > 0.05).mean() < 0.06   # a few pixels flip at silhouettes / checker edges; the rest agree


def _edge_case_geometry(pkg, seed):
    """Axis-aligned boxes on an integer lattice, degenerate and sliver triangles, coincident faces, far-away and tiny geometry."""
    rng = np.random.default_rng(seed)
    b = pkg.host.SceneBuilder()
    b.max_node_prims = int(rng.choice([1, 2, 4]))
    P, I = [], []
    def add(p, i): base = sum(len(q) for q in P); P.append(np.asarray(p, np.float32)); I.append(np.asarray(i, np.uint32) + np.uint32(base))
    for _ in range(int(rng.integers(3, 9))):   # lattice boxes: faces, edges and corners shared exactly
        lo = rng.integers(-3, 3, 3).astype(np.float32); hi = lo + rng.integers(1, 3, 3).astype(np.float32)
        c = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], np.float32)
        add(c, [[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]])
    add([[0, 0, 0], [1, 0, 0], [1, 0, 0]], [[0, 1, 2]])                                  # two equal vertices
    add([[0, 0, 0], [1, 1, 1], [2, 2, 2]], [[0, 1, 2]])                                  # collinear
    add([[0, 5, 0], [1, 5, 0], [0.5, 5 + 1e-7, 0]], [[0, 1, 2]])                         # sliver
    add([[-1, 4, -1], [1, 4, -1], [0, 4, 1]], [[0, 1, 2], [0, 1, 2], [2, 1, 0]])         # coincident, both windings
    add(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]]) * 1e-4 + [0.5, 0.5, 0.5], [[0, 1, 2]])   # tiny
    add(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]]) * 1e4 + [0, 0, -2e4], [[0, 1, 2]])       # huge and far
    b.trianglemesh(np.concatenate(P), np.concatenate(I))
    return b.world_end()[0]


def _edge_case_rays(seed, n):
    rng = np.random.default_rng(seed + 1000)
    o = rng.integers(-4, 5, (n, 3)).astype(np.float32) * np.float32(0.5)                 # lattice and half-lattice origins: on faces / edges
    d = np.zeros((n, 3), np.float32)
    kind = rng.integers(0, 4, n)
    ax = rng.integers(0, 3, n); sg = rng.choice([-1.0, 1.0], n).astype(np.float32)
    d[np.arange(n), ax] = sg                                                               # axis aligned (two zero components -> infinite reciprocals)
    diag = kind == 1; d[diag] = rng.choice([-1.0, 0.0, 1.0], (int(diag.sum()), 3)).astype(np.float32)
    rnd = kind >= 2; d[rnd] = rng.normal(size=(int(rnd.sum()), 3)).astype(np.float32)
    zero = (d == 0).all(axis=1); d[zero, 0] = 1.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = np.where(rng.random(n) < 0.5, np.inf, rng.integers(1, 12, n) * 0.5).astype(np.float32)   # finite t_max exactly at lattice distances
    return o, d.astype(np.float32), t


@pytest.mark.parametrize("seed", range(3))
def test_oracle_traces_edge_case_rays(pkg, oracle, seed):
    s = oracle.scene(_edge_case_geometry(pkg, seed))
    o, d, t = _edge_case_rays(seed, 5000)
    prim, th, bb = s.trace_closest(o, d, t)
    assert (prim != 0xFFFFFFFF).mean() > 0.1 and np.isfinite(th[prim != 0xFFFFFFFF]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(40))
def test_gpu_traversal_of_edge_case_rays(pkg, gpu, oracle, seed):
    """Traversal corners: axis-aligned rays (zero direction components), origins on faces / edges / corners, t_max exactly at a hit,
    degenerate, sliver, coincident, tiny and huge triangles. Hits (primitive, t, barycentrics), occlusion and the node / triangle
    counters must be bit-identical."""
    sd = _edge_case_geometry(pkg, seed)
    g = pkg.Scene(gpu, sd); orc = oracle.scene(sd)
    o, d, t = _edge_case_rays(seed, 20000)
    gp, gt, gb = g.trace_closest(o, d, t); gc = g.counters()
    op, ot, ob = orc.trace_closest(o, d, t); oc = orc.counters()
    assert np.array_equal(gp, op) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32)) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
    for k in ckeys(("bvh_nodes_visited", "triangle_tests", "intersect_tests")): assert gc[k] == oc[k], k
    tf = np.where(np.isinf(t), np.float32(50.0), t)
    gh = g.trace_any(o, d, tf); gc = g.counters()
    oh = orc.trace_any(o, d, tf); oc = orc.counters()
    assert np.array_equal(gh, oh)
    for k in ckeys(("bvh_nodes_visited", "triangle_tests", "shadow_tests")): assert gc[k] == oc[k], k
