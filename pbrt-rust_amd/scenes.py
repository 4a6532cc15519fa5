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


def subsurface_c5(n=24, xres=96, yres=64, spp=16, maxdepth=5, rough=False):
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
    b.material("subsurface", name="Skin1", scale=8.0, eta=1.33, **kw)
    b.translate(-1.1, 0.0, 0.0)
    P, I, N = displaced_sphere(n, with_normals=True)
    b.trianglemesh(P, I, N=N); b.attribute_end()
    b.attribute_begin()
    b.material("kdsubsurface", Kd=(0.7, 0.35, 0.2), mfp=(0.25, 0.15, 0.08), eta=1.4)
    b.translate(1.2, 0.0, 0.3); b.sphere(radius=0.9); b.attribute_end()
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


def ganesha_halton_hlbvh(**kw):
    """The S2 analogue with the reference's default sampler (halton) and the GPU-built accelerator (splitmethod "hlbvh")."""
    b = ganesha_scale(**kw)
    b.sampler = "halton"; b.split_method = "hlbvh"
    return b
