"""Synthetic config scenes: the real Ganesha / Kitchen / ... assets are not
available (no network), so the builder authors deterministic stand-ins of the same scale."""
import numpy as np
from .host import SceneBuilder, F


def _hash3(ix, iy, iz, seed):
    """Integer lattice hash -> [0,1) (uint32 arithmetic, seed 0x9E3779B9 by default)."""
    h = (ix.astype(np.uint32) * np.uint32(0x8DA6B343)) ^ (iy.astype(np.uint32) * np.uint32(0xD8163841)) ^ \
        (iz.astype(np.uint32) * np.uint32(0xCB1AB31F)) ^ np.uint32(seed)
    h ^= h >> np.uint32(16); h = h * np.uint32(0x7FEB352D); h ^= h >> np.uint32(15); h = h * np.uint32(0x846CA68B); h ^= h >> np.uint32(16)
    return (h >> np.uint32(8)).astype(np.float64) * (1.0 / (1 << 24))


def _value_noise(p, seed):
    pf = np.floor(p); f = p - pf
    i = pf.astype(np.int64)
    w = f * f * (3.0 - 2.0 * f)
    acc = 0.0
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                c = _hash3(i[:, 0] + dx, i[:, 1] + dy, i[:, 2] + dz, seed)
                wx = w[:, 0] if dx else 1.0 - w[:, 0]
                wy = w[:, 1] if dy else 1.0 - w[:, 1]
                wz = w[:, 2] if dz else 1.0 - w[:, 2]
                acc = acc + c * wx * wy * wz
    return acc


def fbm(p, octaves=6, seed=0x9E3779B9):
    amp, freq, total = 0.5, 1.0, 0.0
    for o in range(octaves):
        total = total + amp * (2.0 * _value_noise(p * freq, (seed + o * 0x632BE5AB) & 0xFFFFFFFF) - 1.0)
        amp *= 0.5; freq *= 2.0
    return total


def displaced_sphere(n, with_normals=False):
    """UV sphere, n x n quads -> 2*n*n triangles; radius 1 + 0.15*fbm(4p, 6 octaves)."""
    u = np.linspace(0.0, 1.0, n + 1); v = np.linspace(0.0, 1.0, n + 1)
    uu, vv = np.meshgrid(u, v, indexing="xy")
    theta = vv.ravel() * np.pi; phi = uu.ravel() * 2.0 * np.pi
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], axis=1)
    r = 1.0 + 0.15 * fbm(4.0 * d + 100.0, 6)
    P = (d * r[:, None]).astype(F)
    i0 = (np.arange(n)[None, :] + (n + 1) * np.arange(n)[:, None]).ravel().astype(np.uint32)
    tris = np.stack([np.stack([i0, i0 + n + 1, i0 + 1], axis=1), np.stack([i0 + 1, i0 + n + 1, i0 + n + 2], axis=1)], axis=1).reshape(-1, 3)
    N = d.astype(F) if with_normals else None
    return P, tris.astype(np.uint32), N


def quad(p0, p1, p2, p3):
    return np.array([p0, p1, p2, p3], dtype=F), np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)


def ganesha_scale(n=1466, xres=1920, yres=1080, spp=256, maxdepth=5, env=True, with_normals=False, strategy="spatial"):
    """S2 / config C2: 2*n*n-triangle displaced sphere (n=1466 -> 4,298,312 tris), matte Kd .5, ground quad
    20x20, one-sided quad area light 2x2 at y=4 L=(17,12,4), constant env L=.1, camera (0,1.5,5) fov 35,
    sobol, path maxdepth 5, rrthreshold 1, spatial light sampling, bvh sah maxnodeprims 4, box filter."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres)
    b.spp = spp
    b.integ.update(maxdepth=maxdepth, strategy=strategy)
    b.look_at((0.0, 1.5, 5.0), (0.0, 0.3, 0.0), (0.0, 1.0, 0.0))
    b.camera(fov=35.0)
    b.world_begin()
    if env:
        b.light_source("infinite", L=(0.1, 0.1, 0.1))
    b.attribute_begin()
    b.area_light_source(L=(17.0, 12.0, 4.0))
    P, I = quad((-1.0, 4.0, -1.0), (1.0, 4.0, -1.0), (1.0, 4.0, 1.0), (-1.0, 4.0, 1.0))  # normal faces -y
    b.trianglemesh(P, I)
    b.attribute_end()
    b.material("matte", Kd=(0.5, 0.5, 0.5))
    P, I = quad((-10.0, -1.2, -10.0), (-10.0, -1.2, 10.0), (10.0, -1.2, 10.0), (10.0, -1.2, -10.0))
    b.trianglemesh(P, I)
    P, I, N = displaced_sphere(n, with_normals)
    b.trianglemesh(P, I, N=N)
    return b


def material_zoo(n=24, xres=96, yres=64, spp=16, maxdepth=5):
    """S3-style mixed-BSDF test scene: a row of small displaced spheres, one per material class, in a lit room."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres)
    b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((0.0, 2.0, 9.0), (0.0, 0.6, 0.0), (0.0, 1.0, 0.0))
    b.camera(fov=40.0)
    b.world_begin()
    b.light_source("infinite", L=(0.2, 0.25, 0.3))
    b.attribute_begin()
    b.area_light_source(L=(20.0, 18.0, 15.0))
    P, I = quad((-2.0, 5.0, -2.0), (2.0, 5.0, -2.0), (2.0, 5.0, 2.0), (-2.0, 5.0, 2.0))
    b.trianglemesh(P, I)
    b.attribute_end()
    b.attribute_begin()
    b.area_light_source(L=(4.0, 6.0, 9.0), twosided=True)
    P, I = quad((-6.0, 0.5, -3.0), (-6.0, 3.0, -3.0), (-6.0, 3.0, 1.0), (-6.0, 0.5, 1.0))
    b.trianglemesh(P, I)
    b.attribute_end()
    b.material("matte", Kd=(0.6, 0.6, 0.55), sigma=20.0)
    P, I = quad((-12.0, -0.5, -12.0), (-12.0, -0.5, 12.0), (12.0, -0.5, 12.0), (12.0, -0.5, -12.0))
    b.trianglemesh(P, I)
    mats = [("matte", dict(Kd=(0.7, 0.2, 0.2))), ("plastic", dict(Kd=(0.1, 0.3, 0.6), Ks=(0.4, 0.4, 0.4), roughness=0.1)),
            ("metal", dict(eta_rgb=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), roughness=0.05)), ("glass", dict(eta=1.5)),
            ("mirror", dict(Kr=(0.9, 0.9, 0.9))), ("uber", dict(Kd=(0.3, 0.5, 0.2), Ks=(0.3, 0.3, 0.3), Kr=(0.1, 0.1, 0.1), roughness=0.2)),
            ("substrate", dict(Kd=(0.5, 0.3, 0.1), Ks=(0.2, 0.2, 0.2), uroughness=0.1, vroughness=0.2)),
            ("glass", dict(eta=1.4, uroughness=0.2, vroughness=0.2))]
    for k, (kind, kw) in enumerate(mats):
        b.attribute_begin()
        b.material(kind, **kw)
        b.translate(-5.25 + 1.5 * k, 0.35, 0.5 * ((k % 3) - 1))
        b.scale(0.6, 0.6, 0.6)
        P, I, N = displaced_sphere(n, with_normals=(k % 2 == 1))
        b.trianglemesh(P, I, N=N)
        b.attribute_end()
    return b


def emissive_field(n_lights=50000, xres=24, yres=16, spp=2, maxdepth=2, strategy="spatial", seed=5):
    """An emissive mesh: every triangle of an AreaLightSource shape is a light of its own (api.rs:1531-1546) -- n_lights small emissive
    triangles, in warm and cold patches, hang over a matte floor with a blocker. Under "lightsamplestrategy" "spatial" each touched
    voxel of SpatialLightDistribution holds a Distribution1D over ALL of them (lightdistrib.rs:151-228): the case the library's
    first-touch voxel grid exists for (include/mi355pt.h: PtLightStrategy)."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth, strategy=strategy)
    b.look_at((0.0, 2.2, 7.0), (0.0, 0.4, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=42.0)
    b.world_begin()
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(n_lights)))
    k = np.arange(n_lights)
    cx = ((k % side) + 0.5) / side * 8.0 - 4.0 + rng.uniform(-0.3, 0.3, n_lights) * (8.0 / side)
    cz = ((k // side) + 0.5) / side * 8.0 - 4.0 + rng.uniform(-0.3, 0.3, n_lights) * (8.0 / side)
    cy = 3.0 + 0.4 * np.sin(cx * 1.7) * np.cos(cz * 1.3)
    h = 0.35 * 8.0 / side
    P = np.empty((n_lights, 3, 3), dtype=np.float32)
    P[:, 0] = np.stack([cx - h, cy, cz - h], axis=1); P[:, 1] = np.stack([cx + h, cy, cz - h], axis=1); P[:, 2] = np.stack([cx, cy, cz + h], axis=1)
    I = np.arange(3 * n_lights, dtype=np.uint32).reshape(-1, 3)
    half = n_lights // 2
    for lo, hi, L in ((0, half, (900.0, 500.0, 200.0)), (half, n_lights, (150.0, 400.0, 900.0))):   # two AreaLightSource blocks
        if hi <= lo: continue
        b.attribute_begin(); b.area_light_source(L=L, twosided=True)
        b.trianglemesh(P[lo:hi].reshape(-1, 3), (I[lo:hi] - 3 * lo).astype(np.uint32))
        b.attribute_end()
    b.material("matte", Kd=(0.6, 0.6, 0.6))
    Pq, Iq = quad((-4.0, 0.0, -4.0), (-4.0, 0.0, 4.0), (4.0, 0.0, 4.0), (4.0, 0.0, -4.0)); b.trianglemesh(Pq, Iq)
    b.material("plastic", Kd=(0.7, 0.3, 0.2), Ks=(0.3, 0.3, 0.3), roughness=0.2)
    Pq, Iq = quad((-1.5, 0.0, -0.5), (-1.5, 1.6, -0.5), (1.5, 1.6, -0.5), (1.5, 0.0, -0.5)); b.trianglemesh(Pq, Iq)
    return b


def spheres_c1(xres=400, yres=400, spp=64, maxdepth=5):
    """S1 / config C1: LookAt 2 2 5 -> 0 -.4 0, fov 30, sobol, path maxdepth 5, box filter; matte ground quad
    (2 tris, Kd .5), mirror sphere r=1 at x=-1.3, glass sphere (index 1.5) r=1 at x=+1.3, one distant light L=pi from
    (0,10,0) and one two-triangle area light L=10."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres)
    b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((2.0, 2.0, 5.0), (0.0, -0.4, 0.0), (0.0, 1.0, 0.0))
    b.camera(fov=30.0)
    b.world_begin()
    b.light_source("distant", L=(3.14159265, 3.14159265, 3.14159265), from_=(0.0, 10.0, 0.0), to=(0.0, 0.0, 0.0))
    b.attribute_begin()
    b.area_light_source(L=(10.0, 10.0, 10.0))
    P, I = quad((-0.75, 3.5, -0.75), (0.75, 3.5, -0.75), (0.75, 3.5, 0.75), (-0.75, 3.5, 0.75))
    b.trianglemesh(P, I)
    b.attribute_end()
    b.material("matte", Kd=(0.5, 0.5, 0.5))
    P, I = quad((-8.0, -1.0, -8.0), (-8.0, -1.0, 8.0), (8.0, -1.0, 8.0), (8.0, -1.0, -8.0))
    b.trianglemesh(P, I)
    b.attribute_begin(); b.material("mirror", Kr=(0.9, 0.9, 0.9)); b.translate(-1.3, 0.0, 0.0); b.sphere(radius=1.0); b.attribute_end()
    b.attribute_begin(); b.material("glass", eta=1.5); b.translate(1.3, 0.0, 0.0); b.sphere(radius=1.0); b.attribute_end()
    b.attribute_begin(); b.material("plastic", Kd=(0.2, 0.5, 0.2)); b.translate(0.0, -0.6, 1.6); b.rotate(35.0, 1.0, 0.0, 0.0)
    b.sphere(radius=0.4, zmin=-0.25, zmax=0.3, phimax=300.0); b.attribute_end()
    return b


def instanced_garden(n_inst=24, plant_n=10, xres=96, yres=64, spp=8, flatten=False, seed=7):
    """S4-style instancing test (config C4 in miniature): a ground mesh + `n_inst` ObjectInstances of two small
    "plant" objects (one multi-primitive, one single-triangle object) with random translate/rotate/scale (numpy PCG64
    seed), constant environment + one area light. `flatten=True` emits the same geometry without instancing."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.look_at((0.0, 3.0, 9.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=45.0)
    b.world_begin()
    b.light_source("infinite", L=(0.4, 0.45, 0.5))
    b.attribute_begin(); b.area_light_source(L=(25.0, 22.0, 18.0))
    P, I = quad((-1.5, 7.0, -1.5), (1.5, 7.0, -1.5), (1.5, 7.0, 1.5), (-1.5, 7.0, 1.5)); b.trianglemesh(P, I); b.attribute_end()
    b.material("matte", Kd=(0.35, 0.3, 0.2))
    P, I = quad((-12.0, 0.0, -12.0), (-12.0, 0.0, 12.0), (12.0, 0.0, 12.0), (12.0, 0.0, -12.0)); b.trianglemesh(P, I)
    plantP, plantI, plantN = displaced_sphere(plant_n, with_normals=True)
    leafP = np.array([(0.0, 0.0, 0.0), (0.6, 1.2, 0.0), (-0.6, 1.2, 0.1)], dtype=F); leafI = np.array([[0, 1, 2]], dtype=np.uint32)
    if not flatten:
        b.object_begin("plant"); b.material("plastic", Kd=(0.1, 0.5, 0.15), Ks=(0.2, 0.2, 0.2), roughness=0.2)
        b.translate(0.0, 0.5, 0.0); b.scale(0.5, 0.9, 0.5); b.trianglemesh(plantP, plantI, N=plantN); b.object_end()
        b.object_begin("leaf"); b.material("matte", Kd=(0.6, 0.2, 0.1)); b.trianglemesh(leafP, leafI); b.object_end()
    for k in range(n_inst):
        tx, tz = rng.uniform(-6, 6, 2); ang = rng.uniform(0, 360); sc = rng.uniform(0.6, 1.4)
        b.attribute_begin()
        b.translate(float(tx), 0.0, float(tz)); b.rotate(float(ang), 0.0, 1.0, 0.0); b.scale(float(sc), float(sc), float(sc))
        name = "plant" if k % 3 else "leaf"
        if not flatten:
            b.object_instance(name)
        elif name == "plant":
            b.material("plastic", Kd=(0.1, 0.5, 0.15), Ks=(0.2, 0.2, 0.2), roughness=0.2)
            b.translate(0.0, 0.5, 0.0); b.scale(0.5, 0.9, 0.5); b.trianglemesh(plantP, plantI, N=plantN)
        else:
            b.material("matte", Kd=(0.6, 0.2, 0.1)); b.trianglemesh(leafP, leafI)
        b.attribute_end()
    # one instance with the identity transform (transform_surface_interaction is skipped: primitive.rs:73-75)
    if not flatten:
        b.object_instance("leaf")
    else:
        b.material("matte", Kd=(0.6, 0.2, 0.1)); b.trianglemesh(leafP, leafI)
    return b


def subsurface_c5(n=24, xres=96, yres=64, spp=16, maxdepth=5, rough=False, textured_sigma=False, textured_kd=False):
    """Config C5: a displaced sphere with a `subsurface` material (skin-like medium, mm units
    scaled so the mean free path is a visible fraction of the object), a `kdsubsurface` sphere shape and a matte floor,
    lit by an area light and a dim environment.  Exercises path.rs:177-204 / bssrdf.rs sample_s."""
    from .host import SceneBuilder
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((0.0, 1.6, 6.0), (0.0, 0.2, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=38.0)
    b.world_begin()
    b.light_source("infinite", L=(0.25, 0.3, 0.35))
    b.attribute_begin(); b.area_light_source(L=(30.0, 27.0, 22.0))
    P, I = quad((-1.5, 4.0, -0.5), (1.5, 4.0, -0.5), (1.5, 4.0, 1.5), (-1.5, 4.0, 1.5))
    b.trianglemesh(P, I); b.attribute_end()
    b.material("matte", Kd=(0.45, 0.45, 0.5))
    P, I = quad((-8.0, -1.0, -8.0), (-8.0, -1.0, 8.0), (8.0, -1.0, 8.0), (8.0, -1.0, -8.0))
    b.trianglemesh(P, I)
    b.attribute_begin()
    kw = dict(uroughness=0.2, vroughness=0.1) if rough else {}
    if textured_sigma:   # subsurface.rs:100-101,127-128: sigma_a / sigma_s as spectrum textures, evaluated at every entry point
        b.texture("siga", "color", "checkerboard", dimension=3, tex1=(0.0011, 0.0024, 0.014), tex2=(0.02, 0.004, 0.002))
        b.texture("sigs", "color", "checkerboard", mapping="planar", v1=(1.5, 0.0, 0.0), v2=(0.0, 1.5, 0.0), aamode="closedform", tex1=(2.55, 3.21, 3.77), tex2=(1.0, 1.4, 2.2))
        b.material("subsurface", sigma_a="siga", sigma_s="sigs", scale=8.0, eta=1.33, **kw)
    else:
        b.material("subsurface", name="Skin1", scale=8.0, eta=1.33, **kw)
    b.translate(-1.1, 0.0, 0.0)
    P, I, N = displaced_sphere(n, with_normals=True)
    b.trianglemesh(P, I, N=N); b.attribute_end()
    b.attribute_begin()
    if textured_kd:   # kdsubsurface.rs:96-99 with textures: subsurface_from_diffuse at every hit
        b.texture("kdtex", "color", "checkerboard", uscale=6.0, vscale=3.0, tex1=(0.7, 0.35, 0.2), tex2=(0.2, 0.5, 0.8))
        b.texture("mfptex", "color", "checkerboard", dimension=3, tex1=(0.25, 0.15, 0.08), tex2=(0.1, 0.2, 0.3))
        b.material("kdsubsurface", Kd="kdtex", mfp="mfptex", eta=1.4, scale=1.5)
    else:
        b.material("kdsubsurface", Kd=(0.7, 0.35, 0.2), mfp=(0.25, 0.15, 0.08), eta=1.4)
    b.translate(1.2, 0.0, 0.3); b.sphere(radius=0.9); b.attribute_end()
    return b


def subsurface_in_fog(n=16, xres=64, yres=48, spp=8, maxdepth=5, fog=True, sampler="sobol", fog_density=None):
    """Subsurface materials under the VOLUMETRIC integrator (volpath.rs:186-214). With fog=True the camera and the world sit in a homogeneous
    fog, one subsurface object is an ordinary primitive (its hits hand the probe ray's medium on: none for the first probe ray, so the path
    leaves the object in VACUUM -- bssrdf.rs:362-366 starts the chain from an interaction without a MediumInterface) and the other carries a
    MediumInterface of its own (its hits name "juice" inside / "fog" outside, whichever the chain reaches first)."""
    from .host import SceneBuilder
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp; b.sampler = sampler
    b.integ.update(maxdepth=maxdepth, kind="volpath")
    if fog:
        if fog_density is not None: b.make_named_medium("fog", sigma_a=(0.03, 0.03, 0.03), sigma_s=(0.1, 0.1, 0.1), g=0.2, density=fog_density, p0=(-8.0, -1.5, -8.0), p1=(8.0, 5.0, 8.0))   # a GridDensityMedium
        else: b.make_named_medium("fog", sigma_a=(0.03, 0.03, 0.03), sigma_s=(0.1, 0.1, 0.12), g=0.2)
        b.make_named_medium("juice", sigma_a=(0.2, 0.6, 0.9), sigma_s=(0.8, 0.6, 0.4), g=-0.1)
        b.medium_interface("", "fog")
    b.look_at((0.0, 1.6, 6.0), (0.0, 0.2, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=38.0)
    b.world_begin()
    b.light_source("infinite", L=(0.25, 0.3, 0.35))
    b.attribute_begin(); b.area_light_source(L=(30.0, 27.0, 22.0))
    P, I = quad((-1.5, 4.0, -0.5), (1.5, 4.0, -0.5), (1.5, 4.0, 1.5), (-1.5, 4.0, 1.5)); b.trianglemesh(P, I); b.attribute_end()
    b.light_source("point", from_=(2.5, 2.0, 2.5), I=(8.0, 8.0, 7.0))
    b.material("matte", Kd=(0.45, 0.45, 0.5))
    P, I = quad((-8.0, -1.0, -8.0), (-8.0, -1.0, 8.0), (8.0, -1.0, 8.0), (8.0, -1.0, -8.0)); b.trianglemesh(P, I)
    b.attribute_begin()
    b.material("subsurface", name="Skin1", scale=8.0, eta=1.33)
    b.translate(-1.1, 0.0, 0.0)
    P, I, N = displaced_sphere(n, with_normals=True); b.trianglemesh(P, I, N=N); b.attribute_end()
    b.attribute_begin()
    if fog: b.medium_interface("juice", "fog")
    b.material("kdsubsurface", Kd=(0.7, 0.35, 0.2), mfp=(0.25, 0.15, 0.08), eta=1.4)
    b.translate(1.2, 0.0, 0.3); b.sphere(radius=0.9); b.attribute_end()
    return b


def subsurface_sheets(xres=64, yres=48, spp=8, maxdepth=5, n_sheets=40):
    """A stack of thin parallel sheets (plus one sphere) sharing ONE subsurface material, spaced far below the mean free path:
    probe chains (bssrdf.rs:373-395) along the sheets' normal collect tens of matching intersections."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((0.0, 1.2, 4.0), (0.0, 0.3, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=40.0)
    b.world_begin()
    b.light_source("infinite", L=(0.4, 0.45, 0.5))
    b.attribute_begin(); b.area_light_source(L=(20.0, 18.0, 15.0))
    P, I = quad((-1.0, 3.0, -1.0), (1.0, 3.0, -1.0), (1.0, 3.0, 1.0), (-1.0, 3.0, 1.0)); b.trianglemesh(P, I); b.attribute_end()
    b.material("matte", Kd=(0.5, 0.5, 0.5))
    P, I = quad((-6.0, -0.5, -6.0), (-6.0, -0.5, 6.0), (6.0, -0.5, 6.0), (6.0, -0.5, -6.0)); b.trianglemesh(P, I)
    b.material("subsurface", sigma_a=(0.01, 0.02, 0.04), sigma_s=(1.0, 1.2, 1.5), scale=1.0, eta=1.3)
    for k in range(n_sheets):
        z = 0.6 - 0.03 * k
        P, I = quad((-1.2, -0.3, z), (1.2, -0.3, z), (1.2, 1.3, z), (-1.2, 1.3, z)); b.trianglemesh(P, I)
    b.attribute_begin(); b.translate(1.9, 0.2, 0.4); b.sphere(radius=0.5); b.attribute_end()
    return b


def sky_env(w=16, h=8):
    """A small procedural lat-long environment (rows = theta from +z pole, columns = phi): blue-to-white gradient with a
    bright 'sun' patch, so that the importance image is strongly non-uniform."""
    v = (np.arange(h, dtype=np.float64) + 0.5) / h
    u = (np.arange(w, dtype=np.float64) + 0.5) / w
    sky = np.stack([0.25 + 0.5 * v, 0.35 + 0.45 * v, 0.8 - 0.2 * v], axis=-1)[:, None, :] * np.ones((1, w, 1))
    sun = np.exp(-(((u[None, :] - 0.3) * w / 1.5) ** 2 + ((v[:, None] - 0.35) * h / 1.2) ** 2))
    return (sky + 40.0 * sun[..., None] * np.array([1.0, 0.9, 0.7])).astype(F)


def sphere_lights(xres=96, yres=64, spp=16, maxdepth=4):
    """Sphere area lights (row a14 `Sphere::sample_interaction/pdf_wi`, a21 DiffuseAreaLight on a sphere): a two-sided small
    sphere light (cone sampling), a one-sided sphere light (App. A #7: NEE light samples return L = 0; only BSDF-sampled rays
    see it), and a large two-sided partial sphere around a diffuse object (reference point INSIDE the light: uniform-area
    branch + shape_pdfwi)."""
    from .host import SceneBuilder
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((0.0, 2.0, 7.0), (0.0, 0.4, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=40.0)
    b.world_begin()
    b.attribute_begin(); b.area_light_source(L=(30.0, 26.0, 20.0), twosided=True); b.translate(-1.5, 2.5, 0.5); b.sphere(radius=0.3); b.attribute_end()
    b.attribute_begin(); b.area_light_source(L=(8.0, 12.0, 20.0)); b.translate(2.0, 1.2, -0.5); b.sphere(radius=0.5); b.attribute_end()
    b.attribute_begin(); b.area_light_source(L=(0.6, 0.5, 0.4), twosided=True); b.translate(0.0, 0.0, 0.0); b.rotate(-90.0, 1.0, 0.0, 0.0)
    b.sphere(radius=30.0, zmin=-5.0, zmax=30.0); b.attribute_end()
    b.material("matte", Kd=(0.55, 0.55, 0.5))
    P, I = quad((-6.0, -1.0, -6.0), (-6.0, -1.0, 6.0), (6.0, -1.0, 6.0), (6.0, -1.0, -6.0))
    b.trianglemesh(P, I)
    b.attribute_begin(); b.material("plastic", Kd=(0.5, 0.2, 0.2), Ks=(0.4, 0.4, 0.4), roughness=0.05)
    P, I, N = displaced_sphere(12, with_normals=True); b.trianglemesh(P, I, N=N); b.attribute_end()
    b.attribute_begin(); b.material("glass", eta=1.5); b.translate(1.6, -0.4, 1.5); b.sphere(radius=0.6); b.attribute_end()
    return b


def test_image(w=20, h=12, seed=5):
    """A small procedural RGB image (top row first), non-power-of-two so that MIPMap resampling is exercised."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([0.5 + 0.5 * np.sin(x * 0.9) * np.cos(y * 0.7), 0.2 + 0.8 * ((x.astype(int) // 3 + y.astype(int) // 2) % 2), 0.15 + 0.8 * (y / h)], axis=-1)
    rng = np.random.default_rng(seed)
    return np.clip(img + 0.05 * rng.random((h, w, 3)), 0.0, 1.0).astype(F)


def textured(xres=96, yres=64, spp=8, maxdepth=4, trilinear=False, bump=False, noise=False):
    """image map (EWA or trilinear MIPMap, uv and planar mappings), checkerboard (closed-form and point
    sampled, 2-D and 3-D), scale, mix, bilerp, uv, a float image texture driving a roughness, spherical + cylindrical
    mappings; camera-ray differentials drive the filtering at the first hit, later bounces use zero-width lookups."""
    from .host import SceneBuilder
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((0.0, 2.2, 7.0), (0.0, 0.2, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=40.0)
    b.world_begin()
    b.light_source("infinite", L=(0.5, 0.55, 0.6))
    b.attribute_begin(); b.area_light_source(L=(25.0, 22.0, 18.0))
    P, I = quad((-1.5, 5.0, -1.0), (1.5, 5.0, -1.0), (1.5, 5.0, 1.5), (-1.5, 5.0, 1.5)); b.trianglemesh(P, I); b.attribute_end()
    img = test_image()
    b.texture("img", "color", "imagemap", pixels=img, gamma=True, trilinear=trilinear, uscale=4.0, vscale=4.0)
    b.texture("imgplanar", "color", "imagemap", pixels=img, mapping="planar", v1=(0.25, 0.0, 0.0), v2=(0.0, 0.0, 0.25), trilinear=trilinear, maxanisotropy=4.0, wrap="black", scale=1.5)
    b.texture("check", "color", "checkerboard", mapping="planar", v1=(1.0, 0.0, 0.0), v2=(0.0, 0.0, 1.0), aamode="closedform", tex1="imgplanar", tex2=(0.1, 0.1, 0.12))
    b.texture("check3", "color", "checkerboard", dimension=3, tex1=(0.8, 0.3, 0.2), tex2="img")
    b.texture("uvtex", "color", "uv", uscale=3.0, vscale=2.0)
    b.texture("bil", "color", "bilerp", v00=(1, 0, 0), v01=(0, 1, 0), v10=(0, 0, 1), v11=(1, 1, 0))
    b.texture("amount", "float", "checkerboard", uscale=6.0, vscale=6.0, tex1=0.2, tex2=0.9)
    b.texture("mixed", "color", "mix", tex1="uvtex", tex2="bil", amount="amount")
    b.texture("scaled", "color", "scale", tex1="img", tex2=(0.9, 0.6, 0.5))
    b.texture("rough", "float", "imagemap", pixels=img, scale=0.3, uscale=2.0, vscale=2.0, trilinear=True)
    b.texture("sph", "color", "imagemap", pixels=img, mapping="spherical")
    b.texture("cyl", "color", "checkerboard", mapping="cylindrical", tex1=(0.9, 0.9, 0.2), tex2=(0.1, 0.2, 0.7), aamode="closedform")
    bm = lambda name: dict(bumpmap=name) if bump else {}
    if bump:   # displacement textures for bump() (core/material.rs:46-87)
        b.texture("bumpimg", "float", "imagemap", pixels=img, scale=0.08, uscale=6.0, vscale=6.0, trilinear=True)
        b.texture("bumpchk", "float", "checkerboard", mapping="planar", v1=(2.0, 0.0, 0.0), v2=(0.0, 0.0, 2.0), tex1=0.03, tex2=0.0, aamode="closedform")
        b.texture("bumpmix", "float", "mix", tex1="bumpimg", tex2=0.02, amount=0.3)
    b.material("matte", Kd="check", **bm("bumpchk"))
    P, I = quad((-8.0, -1.0, -8.0), (-8.0, -1.0, 8.0), (8.0, -1.0, 8.0), (8.0, -1.0, -8.0))
    b.trianglemesh(P, I, UV=np.array([[0, 0], [0, 1], [1, 1], [1, 0]], dtype=F))
    b.attribute_begin(); b.material("plastic", Kd="mixed", Ks=(0.3, 0.3, 0.3), roughness="rough", **bm("bumpimg")); b.translate(-2.0, 0.0, 0.5); b.sphere(radius=1.0); b.attribute_end()
    b.attribute_begin(); b.material("matte", Kd="check3", **bm("bumpmix")); b.translate(0.3, -0.2, -0.5)
    P, I, N = displaced_sphere(12, with_normals=True); b.trianglemesh(P, I, N=N); b.attribute_end()
    b.attribute_begin(); b.material("uber", Kd="scaled", Ks=(0.2, 0.2, 0.2), opacity=(1, 1, 1), **bm("bumpimg")); b.translate(2.3, 0.0, 0.8); b.sphere(radius=1.0); b.attribute_end()
    b.attribute_begin(); b.translate(0.0, 1.8, -2.5); b.material("matte", Kd="sph"); b.sphere(radius=0.8); b.attribute_end()
    b.attribute_begin(); b.translate(-3.2, 0.6, -2.0); b.material("matte", Kd="cyl"); b.sphere(radius=0.9); b.attribute_end()
    if noise:   # Perlin-noise textures (fbm, wrinkled, windy, marble, dots), as colours, a float roughness and a bump map
        # texture space is shifted so that coordinates stay positive: the reference's noise() saturates negative lattice
        # cells to 0 (`x.floor() as usize`), which makes the interpolation weights explode for negative inputs
        b.transform_begin(); b.translate(-20.0, -20.0, -20.0); b.scale(0.5, 0.5, 0.5)   # TransformBegin: AttributeBegin would pop the textures
        b.texture("marb", "color", "marble", scale=2.0, variation=0.4, octaves=6)
        b.texture("fbmf", "float", "fbm", octaves=5, roughness=0.6)
        b.texture("wrk", "color", "wrinkled", octaves=4)
        b.texture("wind", "float", "windy")
        b.transform_end()
        b.texture("dots", "color", "dots", uscale=8.0, vscale=6.0, inside=(0.9, 0.1, 0.1), outside="marb")
        b.texture("windbump", "float", "scale", tex1="wind", tex2=0.2)
        b.texture("wrkscaled", "color", "scale", tex1="wrk", tex2=(0.35, 0.3, 0.25))
        b.attribute_begin(); b.material("matte", Kd="marb"); b.translate(-1.0, 0.0, 2.6); b.sphere(radius=0.7); b.attribute_end()
        b.attribute_begin(); b.material("plastic", Kd="dots", Ks=(0.2, 0.2, 0.2), roughness=0.2, bumpmap="windbump"); b.translate(1.0, -0.3, 2.8); b.sphere(radius=0.7); b.attribute_end()
        b.attribute_begin(); b.material("matte", Kd="wrkscaled", sigma="fbmf"); b.translate(3.6, 0.2, -1.5)
        P, I, N = displaced_sphere(10, with_normals=True); b.trianglemesh(P, I, N=N); b.attribute_end()
    return b


def alpha_foliage(xres=96, yres=64, spp=8, maxdepth=4, n_cards=40, seed=3, instanced=True):
    """Alpha-masked geometry (SURVEY 8f-1, config C4's foliage): leaf cards whose `alpha` is a checkerboard / image float
    texture (cut-outs seen by camera, bounce and shadow rays), one card with `shadowalpha` only (visible but casts
    partially no shadow), a constant-zero alpha mesh (fully invisible), inside object instances and at top level."""
    from .host import SceneBuilder
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((0.0, 2.5, 8.0), (0.0, 0.8, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=42.0)
    b.world_begin()
    b.light_source("infinite", L=(0.35, 0.4, 0.5))
    b.attribute_begin(); b.area_light_source(L=(30.0, 28.0, 24.0))
    P, I = quad((-2.0, 6.0, -1.0), (2.0, 6.0, -1.0), (2.0, 6.0, 2.0), (-2.0, 6.0, 2.0)); b.trianglemesh(P, I); b.attribute_end()
    b.texture("holes", "float", "checkerboard", uscale=5.0, vscale=5.0, tex1=1.0, tex2=0.0)
    img = test_image(16, 16, seed=9); img[(img[..., 1] < 0.5)] = 0.0
    b.texture("leafmask", "float", "imagemap", pixels=img, trilinear=True)
    b.texture("leafcol", "color", "imagemap", pixels=test_image(16, 16, seed=2), gamma=True)
    b.material("matte", Kd=(0.5, 0.5, 0.45))
    P, I = quad((-9.0, 0.0, -9.0), (-9.0, 0.0, 9.0), (9.0, 0.0, 9.0), (9.0, 0.0, -9.0)); b.trianglemesh(P, I)
    uvq = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=F)
    card = quad((-0.5, 0.0, 0.0), (0.5, 0.0, 0.0), (0.5, 1.0, 0.0), (-0.5, 1.0, 0.0))
    b.material("matte", Kd="leafcol")
    if instanced:
        b.object_begin("plant")
        for k in range(3):
            b.attribute_begin(); b.rotate(60.0 * k, 0, 1, 0); b.trianglemesh(card[0], card[1], UV=uvq, alpha="leafmask"); b.attribute_end()
        b.object_end()
    for _ in range(n_cards):
        b.attribute_begin()
        b.translate(float(rng.uniform(-4, 4)), 0.0, float(rng.uniform(-4, 3))); b.rotate(float(rng.uniform(0, 360)), 0, 1, 0)
        sc = float(rng.uniform(0.6, 1.6)); b.scale(sc, sc, sc)
        if instanced and rng.random() < 0.5: b.object_instance("plant")
        else: b.trianglemesh(card[0], card[1], UV=uvq, alpha="holes")
        b.attribute_end()
    # visible card that only lets light through (shadowalpha), and a fully invisible blocker (alpha = 0)
    b.attribute_begin(); b.material("matte", Kd=(0.8, 0.2, 0.2)); b.translate(0.0, 2.2, 1.0); b.rotate(-90.0, 1, 0, 0); b.scale(3.0, 3.0, 1.0)
    b.trianglemesh(card[0], card[1], UV=uvq, shadowalpha="holes"); b.attribute_end()
    b.attribute_begin(); b.translate(0.0, 0.5, 5.0); b.scale(6.0, 4.0, 1.0); b.trianglemesh(card[0], card[1], UV=uvq, alpha=0.0); b.attribute_end()
    return b


def translucent_panels(xres=96, yres=64, spp=16, maxdepth=5, textured=False):
    """materials/translucent.rs: thin translucent sheets (diffuse + glossy reflection and transmission) between an area
    light and a matte floor, one of them lit from behind only; a textured variant drives reflect / transmit / roughness."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((0.0, 1.6, 5.5), (0.0, 0.6, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=38.0)
    b.world_begin()
    b.light_source("infinite", L=(0.08, 0.08, 0.1))
    b.attribute_begin(); b.area_light_source(L=(20.0, 18.0, 15.0))
    P, I = quad((-0.8, 3.5, -2.5), (0.8, 3.5, -2.5), (0.8, 3.5, -1.0), (-0.8, 3.5, -1.0)); b.trianglemesh(P, I); b.attribute_end()
    b.light_source("point", from_=(0.0, 1.0, -3.0), I=(30.0, 30.0, 40.0))
    b.material("matte", Kd=(0.5, 0.5, 0.5))
    P, I = quad((-10.0, -0.5, -10.0), (-10.0, -0.5, 10.0), (10.0, -0.5, 10.0), (10.0, -0.5, -10.0)); b.trianglemesh(P, I)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=F)
    if textured:
        b.texture("chk", "spectrum", "checkerboard", uscale=4.0, vscale=4.0, tex1=(0.9, 0.2, 0.2), tex2=(0.1, 0.6, 0.9))
        b.texture("rough", "float", "checkerboard", uscale=2.0, vscale=2.0, tex1=0.05, tex2=0.4)
        b.material("translucent", Kd=(0.5, 0.5, 0.5), Ks=(0.4, 0.4, 0.4), reflect="chk", transmit="chk", roughness="rough")
    else:
        b.material("translucent", Kd=(0.6, 0.5, 0.3), Ks=(0.3, 0.3, 0.3), reflect=(0.4, 0.4, 0.4), transmit=(0.7, 0.7, 0.7), roughness=0.2)
    P, I = quad((-2.2, -0.5, -1.5), (-0.4, -0.5, -1.5), (-0.4, 2.0, -1.5), (-2.2, 2.0, -1.5)); b.trianglemesh(P, I, UV=uv)
    b.material("translucent", Kd=(0.0, 0.0, 0.0), Ks=(0.8, 0.8, 0.8), reflect=(0.0, 0.0, 0.0), transmit=(0.9, 0.9, 0.9), roughness=0.05, remaproughness=False)
    P, I = quad((0.4, -0.5, -1.5), (2.2, -0.5, -1.5), (2.2, 2.0, -1.5), (0.4, 2.0, -1.5)); b.trianglemesh(P, I, UV=uv)
    b.material("translucent", reflect=(0.0, 0.0, 0.0), transmit=(0.0, 0.0, 0.0))   # no BSDF at all: skipped like a null surface
    P, I = quad((-0.3, -0.5, 0.5), (0.3, -0.5, 0.5), (0.3, 0.6, 0.5), (-0.3, 0.6, 0.5)); b.trianglemesh(P, I)
    b.material("translucent", transmit=(0.0, 0.0, 0.0))                              # reflection only
    b.attribute_begin(); b.translate(0.0, 0.1, 1.2); b.sphere(radius=0.5); b.attribute_end()
    return b


def mix_materials(xres=96, yres=64, spp=16, maxdepth=5, textured=False):
    """materials/mix.rs: ScaledBxDF blends -- matte + mirror, plastic + rough glass, metal + translucent (1 + 4 lobes) -- with a
    constant or (textured variant) checkerboard "amount"; in the textured variant both sub-materials and a bump map are textured,
    so the second material's textures are evaluated without ray differentials as in the reference."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((0.0, 2.0, 6.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=36.0)
    b.world_begin()
    b.light_source("infinite", L=(0.3, 0.35, 0.45))
    b.attribute_begin(); b.area_light_source(L=(18.0, 16.0, 14.0))
    P, I = quad((-1.0, 4.0, -1.0), (1.0, 4.0, -1.0), (1.0, 4.0, 1.0), (-1.0, 4.0, 1.0)); b.trianglemesh(P, I); b.attribute_end()
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=F)
    amount = (0.3, 0.5, 0.7)
    if textured:
        b.texture("amt", "spectrum", "checkerboard", uscale=3.0, vscale=3.0, tex1=(0.9, 0.9, 0.9), tex2=(0.1, 0.2, 0.3))
        b.texture("img", "spectrum", "imagemap", pixels=test_image(16, 16), uscale=2.0, vscale=2.0, trilinear=True)
        b.texture("bumps", "float", "checkerboard", uscale=6.0, vscale=6.0, tex1=0.02, tex2=0.0)
        amount = "amt"
    b.material("matte", Kd=("img" if textured else (0.7, 0.3, 0.2)), **({"bumpmap": "bumps"} if textured else {})); m_matte = b.material_id
    b.material("mirror", Kr=(0.9, 0.9, 0.9)); m_mirror = b.material_id
    b.material("plastic", Kd=(0.1, 0.4, 0.1), Ks=(0.5, 0.5, 0.5), roughness=0.15); m_plastic = b.material_id
    b.material("glass", Kr=(0.8, 0.8, 0.8), Kt=(0.9, 0.9, 0.9), uroughness=0.2, vroughness=0.3); m_glass = b.material_id
    b.material("metal", roughness=0.05); m_metal = b.material_id
    b.material("translucent", Kd=("img" if textured else (0.5, 0.5, 0.6)), reflect=(0.5, 0.5, 0.5), transmit=(0.6, 0.6, 0.6)); m_trans = b.material_id
    b.material("matte", Kd=(0.5, 0.5, 0.5))
    P, I = quad((-10.0, -0.5, -10.0), (-10.0, -0.5, 10.0), (10.0, -0.5, 10.0), (10.0, -0.5, -10.0)); b.trianglemesh(P, I, UV=uv * 8)
    b.material("mix", amount=amount, namedmaterial1=m_matte, namedmaterial2=m_mirror)
    b.attribute_begin(); b.translate(-1.7, 0.2, 0.0); b.sphere(radius=0.7); b.attribute_end()
    b.material("mix", amount=amount, namedmaterial1=m_plastic, namedmaterial2=m_glass)
    P, I = quad((-0.7, -0.5, 0.3), (0.7, -0.5, 0.3), (0.7, 1.2, -0.3), (-0.7, 1.2, -0.3)); b.trianglemesh(P, I, UV=uv)
    b.material("mix", amount=0.4, namedmaterial1=m_metal, namedmaterial2=m_trans)
    b.attribute_begin(); b.translate(1.7, 0.2, 0.0); b.sphere(radius=0.7); b.attribute_end()
    b.material("mix", amount=amount, namedmaterial1=m_trans, namedmaterial2=m_matte)   # textured second material: no differentials
    P, I = quad((-2.6, -0.5, -2.0), (2.6, -0.5, -2.0), (2.6, 1.6, -2.0), (-2.6, 1.6, -2.0)); b.trianglemesh(P, I, UV=uv * 2)
    return b


def disney_spheres(xres=96, yres=64, spp=16, maxdepth=5, textured=False):
    """materials/disney.rs without the BSSRDF branch: plastic-like (diffuse + retro + sheen + specular + clearcoat), anisotropic
    metal, specular transmission ("glass"), a thin sheet (diffuse + fake subsurface + retro + specular + diffuse transmission),
    and a tinted dielectric; the textured variant drives color and roughness."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((0.0, 2.2, 6.5), (0.0, 0.4, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=36.0)
    b.world_begin()
    b.light_source("infinite", L=(0.35, 0.4, 0.5))
    b.attribute_begin(); b.area_light_source(L=(20.0, 18.0, 16.0))
    P, I = quad((-1.2, 4.5, -1.0), (1.2, 4.5, -1.0), (1.2, 4.5, 1.0), (-1.2, 4.5, 1.0)); b.trianglemesh(P, I); b.attribute_end()
    b.light_source("distant", from_=(3.0, 4.0, 5.0), to=(0.0, 0.0, 0.0), L=(1.5, 1.5, 1.5))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=F)
    color, rough = (0.8, 0.25, 0.15), 0.4
    if textured:
        b.texture("col", "spectrum", "checkerboard", uscale=6.0, vscale=6.0, tex1=(0.8, 0.25, 0.15), tex2=(0.15, 0.3, 0.8))
        b.texture("rgh", "float", "checkerboard", uscale=3.0, vscale=3.0, tex1=0.15, tex2=0.6)
        color, rough = "col", "rgh"
    b.material("disney", color=(0.4, 0.4, 0.4), roughness=0.7, sheen=0.3)
    P, I = quad((-10.0, -0.5, -10.0), (-10.0, -0.5, 10.0), (10.0, -0.5, 10.0), (10.0, -0.5, -10.0)); b.trianglemesh(P, I, UV=uv * 4)
    b.material("disney", color=color, roughness=rough, sheen=0.8, sheentint=0.3, clearcoat=0.7, clearcoatgloss=0.6, speculartint=0.4)
    b.attribute_begin(); b.translate(-2.4, 0.2, 0.0); b.sphere(radius=0.7); b.attribute_end()
    b.material("disney", color=(0.9, 0.7, 0.3), metallic=1.0, roughness=0.3, anisotropic=0.8)
    b.attribute_begin(); b.translate(-0.8, 0.2, 0.0); b.sphere(radius=0.7); b.attribute_end()
    b.material("disney", color=(0.7, 0.9, 0.8), spectrans=1.0, roughness=0.15, eta=1.45, clearcoat=0.2)
    b.attribute_begin(); b.translate(0.8, 0.2, 0.0); b.sphere(radius=0.7); b.attribute_end()
    b.material("disney", color=color, metallic=0.3, spectrans=0.4, roughness=rough, eta=1.3)
    b.attribute_begin(); b.translate(2.4, 0.2, 0.0); b.sphere(radius=0.7); b.attribute_end()
    b.material("disney", color=(0.3, 0.7, 0.3), thin=True, flatness=0.6, difftrans=1.2, roughness=0.5)
    P, I = quad((-1.5, -0.5, -2.0), (1.5, -0.5, -2.0), (1.5, 1.8, -2.0), (-1.5, 1.8, -2.0)); b.trianglemesh(P, I, UV=uv)
    return b


def disk_scene(xres=96, yres=64, spp=16, maxdepth=5):
    """shapes/disk.rs as surfaces and as area lights: full, annular and partial-sweep disks, a non-uniformly scaled and a mirrored
    (handedness-swapping) one, ReverseOrientation, a two-sided-looking pair of emitters (disk lights emit towards their normal
    only, diffuse.rs:73-82) and a disk inside an object instance."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((0.0, 2.6, 6.5), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=38.0)
    b.world_begin()
    b.light_source("infinite", L=(0.15, 0.17, 0.2))
    b.attribute_begin(); b.area_light_source(L=(14.0, 13.0, 12.0))   # facing down: rotate the +z normal to -y
    b.translate(-1.0, 4.0, 0.0); b.rotate(90.0, 1.0, 0.0, 0.0); b.disk(radius=1.1, innerradius=0.3); b.attribute_end()
    b.attribute_begin(); b.area_light_source(L=(4.0, 9.0, 14.0))     # partial sweep, reversed so that it emits upwards-sideways
    b.translate(2.4, 0.4, 0.5); b.rotate(-60.0, 0.0, 1.0, 0.0); b.toggle_reverse_orientation()
    b.disk(radius=0.6, phimax=250.0); b.attribute_end()
    b.material("matte", Kd=(0.5, 0.5, 0.5))
    b.attribute_begin(); b.translate(0.0, -0.5, 0.0); b.rotate(-90.0, 1.0, 0.0, 0.0); b.disk(radius=9.0); b.attribute_end()   # floor
    b.material("plastic", Kd=(0.7, 0.3, 0.2), Ks=(0.4, 0.4, 0.4), roughness=0.08)
    b.attribute_begin(); b.translate(-2.2, 0.6, 0.0); b.rotate(-35.0, 1.0, 0.2, 0.0); b.disk(radius=0.9, innerradius=0.35, phimax=300.0); b.attribute_end()
    b.material("metal", roughness=0.05)
    b.attribute_begin(); b.translate(0.0, 0.8, -1.5); b.scale(1.6, 0.8, 1.0); b.disk(height=0.2, radius=1.0); b.attribute_end()
    b.material("glass", index=1.5)
    b.attribute_begin(); b.translate(1.2, 0.5, 1.2); b.scale(-1.0, 1.0, 1.0); b.rotate(25.0, 0.0, 1.0, 0.0); b.disk(radius=0.7, phimax=200.0); b.attribute_end()
    # z-preserving transforms: the only ones for which Disk::intersect's world-space r.d.z (disk.rs:65) is the right divisor
    b.material("uber", Kd=(0.3, 0.4, 0.7), Ks=(0.3, 0.3, 0.3), roughness=0.2)
    b.attribute_begin(); b.translate(0.0, 1.2, -3.0); b.rotate(30.0, 0.0, 0.0, 1.0); b.scale(1.5, 1.0, 1.0); b.disk(radius=1.6, innerradius=0.5, phimax=300.0); b.attribute_end()
    b.attribute_begin(); b.area_light_source(L=(6.0, 6.0, 6.0)); b.translate(0.0, 2.0, 5.0); b.toggle_reverse_orientation(); b.disk(radius=0.8); b.attribute_end()
    b.material("matte", Kd=(0.2, 0.6, 0.3))
    b.object_begin("d"); b.disk(radius=0.4, innerradius=0.1); b.sphere(radius=0.15); b.object_end()
    for k in range(3):
        b.attribute_begin(); b.translate(-1.5 + 1.5 * k, 0.1, 2.6); b.rotate(-60.0 + 20.0 * k, 1.0, 0.0, 0.0); b.object_instance("d"); b.attribute_end()
    return b


def foggy_room(xres=96, yres=64, spp=16, maxdepth=5, g=0.3, camera_in_fog=True, strategy="spatial"):
    """integrators/volpath.rs + media/homogeneous.rs: the camera sits in a thin homogeneous fog that fills the world (surfaces
    without a MediumInterface keep the ray's medium, primitive.rs:139-145); a glass sphere holds a dense coloured medium (its
    MediumInterface switches media at the dielectric boundary); lights: area light (MIS through the fog), point light, env."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth, kind="volpath", strategy=strategy)
    b.make_named_medium("fog", sigma_a=(0.02, 0.02, 0.02), sigma_s=(0.12, 0.12, 0.14), g=g)
    b.make_named_medium("juice", sigma_a=(0.3, 1.2, 2.0), sigma_s=(1.5, 1.0, 0.6), g=-0.2, scale=1.5)
    if camera_in_fog: b.medium_interface("", "fog")
    b.look_at((0.0, 1.6, 5.5), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=38.0)
    b.world_begin()
    b.light_source("infinite", L=(0.05, 0.06, 0.08))
    b.attribute_begin(); b.area_light_source(L=(25.0, 22.0, 18.0))
    P, I = quad((-0.7, 3.2, -0.7), (0.7, 3.2, -0.7), (0.7, 3.2, 0.7), (-0.7, 3.2, 0.7)); b.trianglemesh(P, I); b.attribute_end()
    b.light_source("point", from_=(-2.0, 2.0, 1.0), I=(12.0, 10.0, 8.0))
    b.material("matte", Kd=(0.5, 0.5, 0.5))
    P, I = quad((-10.0, -0.5, -10.0), (-10.0, -0.5, 10.0), (10.0, -0.5, 10.0), (10.0, -0.5, -10.0)); b.trianglemesh(P, I)
    b.material("plastic", Kd=(0.6, 0.2, 0.2), Ks=(0.3, 0.3, 0.3), roughness=0.2)
    b.attribute_begin(); b.translate(-1.5, 0.1, 0.0); b.sphere(radius=0.6); b.attribute_end()
    b.attribute_begin()
    b.medium_interface("juice", "fog" if camera_in_fog else "")
    b.material("glass", Kr=(1.0, 1.0, 1.0), Kt=(1.0, 1.0, 1.0), eta=1.33)
    b.translate(0.9, 0.3, 0.3); b.sphere(radius=0.8)
    b.attribute_end()
    return b


def shell_media(xres=64, yres=48, spp=8, maxdepth=5, grid=True, sampler="sobol", light_inside=True):
    """Media bounded the way a .pbrt file bounds them: by shapes WITHOUT a material (`Material "none"`, api.rs:597) that only carry a
    MediumInterface. The camera stands in vacuum; a sphere shell holds a homogeneous coloured fog, a box shell a GridDensityMedium
    (grid=True), the two overlap the light paths of an opaque floor, a wall, an area light, a point light (inside the fog when
    light_inside) and an environment. Every shadow ray and MIS ray that crosses a shell walks VisibilityTester::tr / Scene::intersect_tr
    segment by segment (light.rs:125-150, scene.rs:68-87); camera paths cross with volpath.rs:152-156's `bounces -= 1`."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp; b.sampler = sampler
    b.integ.update(maxdepth=maxdepth, kind="volpath")
    b.make_named_medium("fog", sigma_a=(0.15, 0.3, 0.6), sigma_s=(0.9, 0.7, 0.5), g=0.3)
    if grid:
        rng = np.random.default_rng(9)
        dens = rng.uniform(0.1, 1.0, (5, 4, 6)).astype(np.float32)
        b.make_named_medium("smoke", sigma_a=(0.4, 0.4, 0.4), sigma_s=(1.6, 1.6, 1.6), g=-0.2, density=dens, p0=(0.6, 0.0, -1.0), p1=(2.6, 1.6, 1.0))
    b.look_at((0.0, 1.8, 6.5), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=40.0)
    b.world_begin()
    b.light_source("infinite", L=(0.08, 0.1, 0.14))
    b.attribute_begin(); b.area_light_source(L=(30.0, 27.0, 22.0))
    P, I = quad((-1.0, 3.6, -1.0), (1.0, 3.6, -1.0), (1.0, 3.6, 1.0), (-1.0, 3.6, 1.0)); b.trianglemesh(P, I); b.attribute_end()
    b.light_source("point", from_=((-1.4, 0.9, 0.0) if light_inside else (-3.5, 2.5, 1.0)), I=(6.0, 6.0, 5.0))
    b.material("matte", Kd=(0.55, 0.55, 0.5))
    P, I = quad((-8.0, 0.0, -8.0), (-8.0, 0.0, 8.0), (8.0, 0.0, 8.0), (8.0, 0.0, -8.0)); b.trianglemesh(P, I)
    b.material("plastic", Kd=(0.2, 0.5, 0.3), Ks=(0.3, 0.3, 0.3), roughness=0.15)
    P, I = quad((-4.0, 0.0, -2.5), (-4.0, 3.0, -2.5), (4.0, 3.0, -2.5), (4.0, 0.0, -2.5)); b.trianglemesh(P, I)
    # the homogeneous fog in a sphere shell (a real sphere: Sphere::intersect through the same loops)
    b.attribute_begin(); b.material("none"); b.medium_interface("fog", ""); b.translate(-1.4, 0.95, 0.0); b.sphere(radius=0.9); b.attribute_end()
    if grid:   # the grid medium in a box shell of twelve triangles
        b.attribute_begin(); b.material("none"); b.medium_interface("smoke", "")
        x0, y0, z0, x1, y1, z1 = 0.6, 0.0, -1.0, 2.6, 1.6, 1.0
        c = [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]
        faces = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (3, 7, 6, 2), (0, 4, 7, 3), (1, 2, 6, 5)]   # outward normals
        for f in faces:
            P, I = quad(*[c[k] for k in f]); b.trianglemesh(P, I)
        b.attribute_end()
    b.material("metal", eta_rgb=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), roughness=0.1)
    b.attribute_begin(); b.translate(0.2, 0.35, 1.6); b.sphere(radius=0.35); b.attribute_end()
    return b


def ganesha_halton_hlbvh(**kw):
    """The S2 analogue with the reference's default sampler (halton) and the GPU-built accelerator (splitmethod "hlbvh")."""
    b = ganesha_scale(**kw)
    b.sampler = "halton"; b.split_method = "hlbvh"
    return b


# ---- the S3 / S4 / S5 workloads at their specified scale (configs C3 / C4 / C5) --------------------

class PCG32:
    """core/rng.rs:10-75 (PCG32, `RNG::new(sequence_index)`): the generator SURVEY 8(d) names for S4's random transforms."""
    MULT, DEFAULT_STATE, M64 = 0x5851F42D4C957F2D, 0x853C49E6748FEA9B, (1 << 64) - 1

    def __init__(self, seq):
        self.state, self.inc = 0, ((seq << 1) | 1) & self.M64
        self.u32(); self.state = (self.state + self.DEFAULT_STATE) & self.M64; self.u32()

    def u32(self):
        old = self.state
        self.state = (old * self.MULT + self.inc) & self.M64
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF; rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def f(self):
        return min(float(np.float32(0.99999994)), float(np.float32(self.u32()) * np.float32(2.0 ** -32)))

    def uniform(self, lo, hi):
        return lo + (hi - lo) * self.f()


def grid_patch(p00, du, dv, nu, nv):
    """Planar patch p00 + s*du + t*dv, s,t in [0,1], subdivided into nu x nv quads (2*nu*nv triangles, normal du x dv)."""
    p00, du, dv = (np.asarray(x, dtype=np.float64) for x in (p00, du, dv))
    s = np.linspace(0.0, 1.0, nu + 1); t = np.linspace(0.0, 1.0, nv + 1)
    ss, tt = np.meshgrid(s, t, indexing="xy")
    P = p00[None, :] + ss.ravel()[:, None] * du[None, :] + tt.ravel()[:, None] * dv[None, :]
    i0 = (np.arange(nu)[None, :] + (nu + 1) * np.arange(nv)[:, None]).ravel().astype(np.uint32)
    tris = np.stack([np.stack([i0, i0 + 1, i0 + nu + 2], axis=1), np.stack([i0, i0 + nu + 2, i0 + nu + 1], axis=1)], axis=1).reshape(-1, 3)
    return P.astype(F), tris.astype(np.uint32)


def merge_meshes(parts):
    Ps, Is, base = [], [], 0
    for P, I in parts:
        Ps.append(P); Is.append(I + np.uint32(base)); base += len(P)
    return np.concatenate(Ps).astype(F), np.concatenate(Is).astype(np.uint32)


def subdivided_box(lo, hi, n):
    """Axis-aligned box, every face an n x n grid (12*n*n triangles), outward normals."""
    x0, y0, z0 = lo; x1, y1, z1 = hi
    dx, dy, dz = (x1 - x0, 0, 0), (0, y1 - y0, 0), (0, 0, z1 - z0)
    faces = [((x0, y0, z0), dz, dy), ((x1, y0, z0), dy, dz), ((x0, y0, z0), dx, dz), ((x0, y1, z0), dz, dx), ((x0, y0, z0), dy, dx), ((x0, y0, z1), dx, dy)]
    return merge_meshes([grid_patch(p, a, b, n, n) for p, a, b in faces])


def torus_mesh(nu, nv, R, r):
    """Torus around the y axis: nu x nv quads, per-vertex normals."""
    u = np.linspace(0.0, 2.0 * np.pi, nu + 1); v = np.linspace(0.0, 2.0 * np.pi, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="xy"); uu, vv = uu.ravel(), vv.ravel()
    P = np.stack([(R + r * np.cos(vv)) * np.cos(uu), r * np.sin(vv), (R + r * np.cos(vv)) * np.sin(uu)], axis=1)
    N = np.stack([np.cos(vv) * np.cos(uu), np.sin(vv), np.cos(vv) * np.sin(uu)], axis=1)
    i0 = (np.arange(nu)[None, :] + (nu + 1) * np.arange(nv)[:, None]).ravel().astype(np.uint32)
    tris = np.stack([np.stack([i0, i0 + nu + 1, i0 + 1], axis=1), np.stack([i0 + 1, i0 + nu + 1, i0 + nu + 2], axis=1)], axis=1).reshape(-1, 3)
    return P.astype(F), tris.astype(np.uint32), N.astype(F)


def pot_mesh(nu, nv):
    """A lathe "pot" (teapot-body stand-in): surface of revolution of a bulged profile around the y axis, nu x nv quads."""
    u = np.linspace(0.0, 2.0 * np.pi, nu + 1); t = np.linspace(0.0, 1.0, nv + 1)
    uu, tt = np.meshgrid(u, t, indexing="xy"); uu, tt = uu.ravel(), tt.ravel()
    rad = 0.08 + 0.55 * np.sin(np.pi * np.clip(tt * 0.92 + 0.04, 0.0, 1.0)) ** 0.8 + 0.12 * np.exp(-((tt - 0.93) / 0.04) ** 2) + 0.03 * np.cos(12.0 * uu) * np.sin(np.pi * tt)
    P = np.stack([rad * np.cos(uu), 1.1 * tt, rad * np.sin(uu)], axis=1)
    i0 = (np.arange(nu)[None, :] + (nu + 1) * np.arange(nv)[:, None]).ravel().astype(np.uint32)
    tris = np.stack([np.stack([i0, i0 + nu + 1, i0 + 1], axis=1), np.stack([i0 + 1, i0 + nu + 1, i0 + nu + 2], axis=1)], axis=1).reshape(-1, 3)
    return P.astype(F), tris.astype(np.uint32)


def country_kitchen_s3(xres=1920, yres=1080, spp=1024, maxdepth=5, wall_n=280, box_n=48, obj_n=112, n_objects=8, mixed=False):
    """S3 / config C3: Cornell-style closed room, ~1 M triangles from subdivided walls and boxes (6 walls of wall_n^2
    quads + two boxes of box_n^2 quads per face) + ~200 k triangles of tessellated tori / lathe pots (n_objects x 2*obj_n^2),
    materials assigned round-robin from {matte, plastic roughness .1, uber, metal (copper defaults, metal.rs:13-53,116-117),
    mirror, glass index 1.5} and 64 emissive triangles (a 8x4 grid of quads under the ceiling, one DiffuseAreaLight each).
    `mixed=True` adds substrate, rough glass and translucent to the round-robin (the materials SURVEY 8's C3 row lists besides S3's six): one of each
    lobe count that the lobe-set shade kernels do not cover, next to the metals, plastics and ubers that they do."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((0.0, 3.0, 4.7), (0.0, 2.0, -2.0), (0.0, 1.0, 0.0)); b.camera(fov=62.0)
    b.world_begin()
    palette = [("matte", dict(Kd=(0.6, 0.6, 0.6))), ("plastic", dict(Kd=(0.25, 0.35, 0.6), Ks=(0.25, 0.25, 0.25), roughness=0.1)),
               ("uber", dict(Kd=(0.5, 0.25, 0.2), Ks=(0.25, 0.25, 0.25), Kr=(0.1, 0.1, 0.1), roughness=0.1)), ("metal", dict()),
               ("mirror", dict(Kr=(0.9, 0.9, 0.9))), ("glass", dict(eta=1.5))]
    if mixed:
        palette += [("substrate", dict(Kd=(0.5, 0.3, 0.1), Ks=(0.2, 0.2, 0.2), uroughness=0.1, vroughness=0.2)), ("glass", dict(eta=1.4, uroughness=0.2, vroughness=0.2)),
                    ("translucent", dict(Kd=(0.6, 0.5, 0.3), Ks=(0.3, 0.3, 0.3), reflect=(0.4, 0.4, 0.4), transmit=(0.7, 0.7, 0.7), roughness=0.2))]
    counter = [0]
    def next_material():
        kind, kw = palette[counter[0] % len(palette)]; counter[0] += 1
        b.material(kind, **kw)
    # 64 emissive triangles: 8 x 4 quads, 0.5 x 0.5 each, facing down
    b.attribute_begin(); b.area_light_source(L=(22.0, 20.0, 16.0))
    for j in range(4):
        for i in range(8):
            x0, z0 = -4.0 + i * 1.0 + 0.25, -3.0 + j * 1.5
            P, I = quad((x0, 5.98, z0), (x0 + 0.5, 5.98, z0), (x0 + 0.5, 5.98, z0 + 0.5), (x0, 5.98, z0 + 0.5))
            b.trianglemesh(P, I)
    b.attribute_end()
    # room [-5,5] x [0,6] x [-5,5], inward normals; order: floor, left, right, back, ceiling, front
    walls = [((-5, 0, -5), (0, 0, 10), (10, 0, 0)), ((-5, 0, -5), (0, 6, 0), (0, 0, 10)), ((5, 0, -5), (0, 0, 10), (0, 6, 0)),
             ((-5, 0, -5), (10, 0, 0), (0, 6, 0)), ((-5, 6, -5), (10, 0, 0), (0, 0, 10)), ((-5, 0, 5), (0, 6, 0), (10, 0, 0))]
    for p, du, dv in walls:
        next_material()
        P, I = grid_patch(p, du, dv, wall_n, wall_n); b.trianglemesh(P, I)
    for lo, hi, ang in (((-0.9, 0.0, -0.9), (0.9, 3.4, 0.9), 17.0), ((-0.8, 0.0, -0.8), (0.8, 1.6, 0.8), -20.0)):
        b.attribute_begin(); next_material()
        b.translate(-1.9 if ang > 0 else 1.8, 0.0, -2.2 if ang > 0 else -0.4); b.rotate(ang, 0.0, 1.0, 0.0)
        P, I = subdivided_box(lo, hi, box_n); b.trianglemesh(P, I); b.attribute_end()
    for k in range(n_objects):
        b.attribute_begin(); next_material()
        ang = 2.0 * np.pi * k / max(1, n_objects)
        cx, cz = 3.3 * np.cos(ang), -0.6 + 2.9 * np.sin(ang)
        if k % 2 == 0:
            b.translate(float(cx), 0.32 + 0.9 * (k % 4 == 2), float(cz)); b.rotate(25.0 * k, 1.0, 0.3, 0.0)
            P, I, N = torus_mesh(obj_n, obj_n, 0.62, 0.26); b.trianglemesh(P, I, N=N)
        else:
            b.translate(float(cx), 0.0, float(cz)); b.rotate(40.0 * k, 0.0, 1.0, 0.0); b.scale(0.9, 0.9, 0.9)
            P, I = pot_mesh(obj_n, obj_n); b.trianglemesh(P, I)
        b.attribute_end()
    return b


def terrain_height(x, z):
    """S4 terrain: 3 * fbm over a 80 x 80 field (x, z arrays -> y)."""
    p = np.stack([np.asarray(x, dtype=np.float64) * 0.05 + 31.0, np.full(np.shape(x), 7.0), np.asarray(z, dtype=np.float64) * 0.05 + 17.0], axis=-1).reshape(-1, 3)
    return (3.0 * fbm(p, 5)).reshape(np.shape(x))


def plant_meshes(scale=1.0):
    """Three 50 k-triangle "plants": a bush (displaced sphere, 2*158^2 = 49,928 triangles), a tree (cone trunk 2,048 + 23,976 leaf
    quads = 50,000) and a grass tuft (5,000 blades x 10 triangles = 50,000). `scale` < 1 thins them out for small tests."""
    rng = np.random.default_rng(20260107)
    n = max(4, int(round(158 * scale)))
    bushP, bushI, bushN = displaced_sphere(n, with_normals=True)
    bushP = (bushP * np.array([0.9, 0.8, 0.9], dtype=F) + np.array([0.0, 0.8, 0.0], dtype=F)).astype(F)
    # tree: trunk + leaf quads in an ellipsoidal crown
    nt_u, nt_v = max(4, int(64 * scale)), max(2, int(16 * scale))
    u = np.linspace(0.0, 2.0 * np.pi, nt_u + 1); t = np.linspace(0.0, 1.0, nt_v + 1)
    uu, tt = np.meshgrid(u, t, indexing="xy"); uu, tt = uu.ravel(), tt.ravel()
    rad = 0.16 * (1.0 - 0.8 * tt)
    trunkP = np.stack([rad * np.cos(uu), 2.0 * tt, rad * np.sin(uu)], axis=1)
    i0 = (np.arange(nt_u)[None, :] + (nt_u + 1) * np.arange(nt_v)[:, None]).ravel().astype(np.uint32)
    trunkI = np.stack([np.stack([i0, i0 + nt_u + 1, i0 + 1], axis=1), np.stack([i0 + 1, i0 + nt_u + 1, i0 + nt_u + 2], axis=1)], axis=1).reshape(-1, 3)
    n_leaf = max(8, int(round(23976 * scale * scale)))
    c = rng.normal(size=(n_leaf, 3)); c /= np.linalg.norm(c, axis=1, keepdims=True); c *= rng.random((n_leaf, 1)) ** (1.0 / 3.0)
    c = c * np.array([1.1, 0.9, 1.1]) + np.array([0.0, 2.3, 0.0])
    a = rng.normal(size=(n_leaf, 3)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    bb = np.cross(a, rng.normal(size=(n_leaf, 3))); bb /= np.linalg.norm(bb, axis=1, keepdims=True)
    s = 0.07
    leafP = np.stack([c - s * a - s * bb, c + s * a - s * bb, c + s * a + s * bb, c - s * a + s * bb], axis=1).reshape(-1, 3)
    q = (4 * np.arange(n_leaf, dtype=np.uint32))[:, None]
    leafI = np.concatenate([q + np.array([0, 1, 2], dtype=np.uint32), q + np.array([0, 2, 3], dtype=np.uint32)], axis=1).reshape(-1, 3)
    treeP, treeI = merge_meshes([(trunkP.astype(F), trunkI.astype(np.uint32)), (leafP.astype(F), leafI.astype(np.uint32))])
    # grass tuft: blades of 5 segments (10 triangles), bending outwards
    n_blade = max(4, int(round(5000 * scale * scale)))
    base = rng.normal(size=(n_blade, 2)) * 0.35; dirn = rng.normal(size=(n_blade, 2)); dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    height = rng.uniform(0.5, 1.3, n_blade); bend = rng.uniform(0.1, 0.6, n_blade)
    seg = np.linspace(0.0, 1.0, 6)
    side = np.stack([-dirn[:, 1], dirn[:, 0]], axis=1) * 0.012
    cx = base[:, None, 0] + dirn[:, None, 0] * bend[:, None] * seg[None, :] ** 2; cz = base[:, None, 1] + dirn[:, None, 1] * bend[:, None] * seg[None, :] ** 2
    cy = height[:, None] * seg[None, :]
    wid = (1.0 - 0.85 * seg)[None, :]
    left = np.stack([cx - side[:, None, 0] * wid, cy, cz - side[:, None, 1] * wid], axis=2); right = np.stack([cx + side[:, None, 0] * wid, cy, cz + side[:, None, 1] * wid], axis=2)
    grassP = np.stack([left, right], axis=2).reshape(n_blade, 12, 3)    # vertex 2*j = left_j, 2*j+1 = right_j
    j = np.arange(5, dtype=np.uint32)
    one = np.stack([np.stack([2 * j, 2 * j + 1, 2 * j + 3], axis=1), np.stack([2 * j, 2 * j + 3, 2 * j + 2], axis=1)], axis=1).reshape(-1, 3)
    grassI = ((12 * np.arange(n_blade, dtype=np.uint32))[:, None, None] + one[None, :, :]).reshape(-1, 3)
    return (bushP, bushI, bushN), (treeP, treeI.astype(np.uint32), None), (grassP.reshape(-1, 3).astype(F), grassI.astype(np.uint32), None)


def ecosystem_s4(xres=1920, yres=1080, spp=2048, maxdepth=5, n_inst=2000, terrain_n=500, plant_scale=1.0, env_size=(512, 256), seed=7, flatten=False):
    """S4 / config C4: `n_inst` ObjectInstances of three 50 k-triangle plants (random translate / rotate / scale from
    PCG32 sequence `seed`, rng.rs) over a 2*terrain_n^2-triangle fbm terrain, lit only by a 512x256 synthetic sky environment map
    (importance-sampled InfiniteAreaLight). `flatten=True` emits the same geometry without instancing (small tests)."""
    rng = PCG32(seed)
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((0.0, 7.0, 44.0), (0.0, 1.0, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=40.0)
    b.world_begin()
    b.attribute_begin(); b.rotate(-90.0, 1.0, 0.0, 0.0)   # lat-long map: +z pole -> +y
    b.light_source("infinite", L=(1.0, 1.0, 1.0), texels=sky_env(*env_size)); b.attribute_end()
    g = np.linspace(-40.0, 40.0, terrain_n + 1)
    gx, gz = np.meshgrid(g, g, indexing="xy")
    tP = np.stack([gx.ravel(), terrain_height(gx.ravel(), gz.ravel()), gz.ravel()], axis=1).astype(F)
    i0 = (np.arange(terrain_n)[None, :] + (terrain_n + 1) * np.arange(terrain_n)[:, None]).ravel().astype(np.uint32)
    tI = np.stack([np.stack([i0, i0 + terrain_n + 1, i0 + 1], axis=1), np.stack([i0 + 1, i0 + terrain_n + 1, i0 + terrain_n + 2], axis=1)], axis=1).reshape(-1, 3)
    b.material("matte", Kd=(0.32, 0.28, 0.18)); b.trianglemesh(tP, tI.astype(np.uint32))
    plants = plant_meshes(plant_scale)
    mats = [("plastic", dict(Kd=(0.1, 0.45, 0.12), Ks=(0.15, 0.15, 0.15), roughness=0.25)), ("matte", dict(Kd=(0.18, 0.5, 0.15))), ("matte", dict(Kd=(0.35, 0.6, 0.2)))]
    names = ("bush", "tree", "grass")
    if not flatten:
        for name, (P, I, N), (kind, kw) in zip(names, plants, mats):
            b.object_begin(name); b.material(kind, **kw); b.trianglemesh(P, I, N=N); b.object_end()
    for k in range(n_inst):
        tx, tz = rng.uniform(-38.0, 38.0), rng.uniform(-38.0, 38.0)
        ang, tilt, sc = rng.uniform(0.0, 360.0), rng.uniform(-8.0, 8.0), rng.uniform(0.6, 1.6)
        ty = float(terrain_height(np.array([tx]), np.array([tz]))[0]) - 0.05
        b.attribute_begin()
        b.translate(tx, ty, tz); b.rotate(ang, 0.0, 1.0, 0.0); b.rotate(tilt, 1.0, 0.0, 0.0); b.scale(sc, sc, sc)
        if not flatten: b.object_instance(names[k % 3])
        else:
            (P, I, N), (kind, kw) = plants[k % 3], mats[k % 3]
            b.material(kind, **kw); b.trianglemesh(P, I, N=N)
        b.attribute_end()
    return b


def dragon_s5(xres=1920, yres=1080, spp=4096, maxdepth=5, n=1466, env_size=(512, 256)):
    """S5 / config C5: the S2 mesh (2*n^2-triangle displaced sphere) scaled x0.02 with
    `Material "subsurface" "string name" "Skin1" "float scale" 20 "float eta" 1.5` (src/scenes/sss-dragon.pbrt) over a plain
    matte plane, lit by an environment map."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth)
    b.look_at((0.0, 0.03, 0.1), (0.0, 0.006, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=35.0)
    b.world_begin()
    b.attribute_begin(); b.rotate(-90.0, 1.0, 0.0, 0.0)
    b.light_source("infinite", L=(1.0, 1.0, 1.0), texels=sky_env(*env_size)); b.attribute_end()
    b.material("matte", Kd=(0.4, 0.4, 0.4))
    P, I = quad((-0.2, -0.024, -0.2), (-0.2, -0.024, 0.2), (0.2, -0.024, 0.2), (0.2, -0.024, -0.2)); b.trianglemesh(P, I)
    b.attribute_begin()
    b.scale(0.02, 0.02, 0.02)
    b.material("subsurface", name="Skin1", scale=20.0, eta=1.5)
    P, I, N = displaced_sphere(n, with_normals=False)
    b.trianglemesh(P, I, N=N); b.attribute_end()
    return b


CONFIG_SCENES = {   # bench.py --config / tests: name -> (builder, spp named by BASELINE.json, description)
    "C2": (ganesha_scale, 256, "S2 Ganesha-scale: 4,298,312-triangle displaced sphere (matte) + ground + quad area light + constant env"),
    "C3M": (lambda **kw: country_kitchen_s3(mixed=True, **kw), 1024, "S3 with substrate, rough glass and translucent added to the material round-robin (round 5: the lobe-set shade classes in a mixed scene)"),
    "C3": (country_kitchen_s3, 1024, "S3 Country-Kitchen-scale: closed room of subdivided walls/boxes (~1.0 M triangles) + 8 tessellated tori/pots (~0.2 M), "
           "materials round-robin {matte, plastic, uber, metal, mirror, glass}, 64 emissive triangles"),
    "C4": (ecosystem_s4, 2048, "S4 Ecosystem-scale: 2,000 object instances of three 50 k-triangle plants over a 500 k-triangle terrain, 512x256 environment map only"),
    "C5": (dragon_s5, 4096, "S5 Dragon-subsurface-scale: the 4,298,312-triangle S2 mesh x0.02 with subsurface Skin1 (scale 20, eta 1.5) over a matte plane, environment map"),
}


def smoke_room(xres=64, yres=48, spp=16, maxdepth=4, n=8, g=0.2, sampler="sobol"):
    """media/grid.rs: the camera (and so the whole world) sits in a GridDensityMedium whose density -- a soft blob on an n^3 grid -- lives in the
    box [-1.2, 1.2] x [-0.4, 2.0] x [-1.2, 1.2]; outside the box rays miss the medium's bounds (Tr = 1). Ratio tracking on shadow / MIS
    rays and delta tracking on path segments draw a data-dependent number of sampler dimensions in the middle of a vertex."""
    b = SceneBuilder()
    b.film.update(xres=xres, yres=yres); b.spp = spp
    b.integ.update(maxdepth=maxdepth, kind="volpath")
    b.sampler = sampler
    z, y, x = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n), np.linspace(-1, 1, n), indexing="ij")
    dens = (np.exp(-2.5 * (x * x + (y * 1.2) ** 2 + z * z)) * (1.0 + 0.3 * np.sin(5.0 * x) * np.cos(4.0 * z))).astype(F)
    b.make_named_medium("smoke", sigma_a=(0.4, 0.4, 0.4), sigma_s=(1.6, 1.6, 1.6), g=g, density=dens, p0=(-1.2, -0.4, -1.2), p1=(1.2, 2.0, 1.2))
    b.medium_interface("", "smoke")
    b.look_at((0.0, 1.4, 5.0), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=38.0)
    b.world_begin()
    b.light_source("infinite", L=(0.05, 0.06, 0.08))
    b.attribute_begin(); b.area_light_source(L=(25.0, 22.0, 18.0))
    P, I = quad((-0.7, 3.4, -0.7), (0.7, 3.4, -0.7), (0.7, 3.4, 0.7), (-0.7, 3.4, 0.7)); b.trianglemesh(P, I); b.attribute_end()
    b.light_source("point", from_=(-2.5, 1.5, -2.5), I=(14.0, 12.0, 10.0))
    b.material("matte", Kd=(0.5, 0.5, 0.5))
    P, I = quad((-10.0, -0.5, -10.0), (-10.0, -0.5, 10.0), (10.0, -0.5, 10.0), (10.0, -0.5, -10.0)); b.trianglemesh(P, I)
    b.material("plastic", Kd=(0.6, 0.2, 0.2), Ks=(0.3, 0.3, 0.3), roughness=0.2)
    b.attribute_begin(); b.translate(-1.9, 0.1, 0.3); b.sphere(radius=0.6); b.attribute_end()
    return b
