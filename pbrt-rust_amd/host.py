"""Host-side scene assembly: a Python mirror of the reference's scene API for the hot path.

Plays the role the Rust host plays in front of the C ABI: it keeps the graphics state
(CTM, current material, area light, reverse-orientation) exactly like core/api.rs:941-1327,1493-1619
and flattens everything into the POD structs of include/mi355pt.h.  There is no compute here
that the render path depends on besides the parameter set-up the reference also does on the host
(camera matrices cameras/perspective.rs:40-86, film bounds film.rs:55-112, filter table film.rs:76-89).
"""
import ctypes as C
import math
import sys
import numpy as np
from . import _abi as A

F = np.float32


class Transform:
    """core/transform.rs Transform {m, m_inv}, row major, f32."""

    def __init__(self, m=None, m_inv=None):
        self.m = np.eye(4, dtype=F) if m is None else np.asarray(m, dtype=F)
        if m_inv is None:
            m_inv = np.linalg.inv(self.m.astype(np.float64)).astype(F)
        self.m_inv = np.asarray(m_inv, dtype=F)

    def __mul__(self, o):  # transform.rs Mul: (m1*m2, m2_inv*m1_inv)
        return Transform(_matmul(self.m, o.m), _matmul(o.m_inv, self.m_inv))

    def inverse(self):
        return Transform(self.m_inv, self.m)

    def swaps_handedness(self):  # transform.rs swaps_handedness: det of upper 3x3 < 0
        m = self.m
        det = (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0]) +
               m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))
        return bool(det < 0)

    def point(self, p):
        p = np.asarray(p, dtype=F)
        if p.ndim == 1:
            x, y, z = F(p[0]), F(p[1]), F(p[2])
            m = self.m
            r = np.array([x * m[i, 0] + y * m[i, 1] + z * m[i, 2] + m[i, 3] for i in range(4)], dtype=F)
            return r[:3] if r[3] == F(1) else (r[:3] * (F(1) / r[3])).astype(F)
        # batched, same op order as transform.rs:413-432
        m = self.m
        x, y, z = p[:, 0], p[:, 1], p[:, 2]
        out = np.stack([x * m[i, 0] + y * m[i, 1] + z * m[i, 2] + m[i, 3] for i in range(3)], axis=1).astype(F)
        return out

    def vector(self, v):
        v = np.asarray(v, dtype=F)
        m = self.m
        if v.ndim == 1:
            return np.array([v[0] * m[i, 0] + v[1] * m[i, 1] + v[2] * m[i, 2] for i in range(3)], dtype=F)
        return np.stack([v[:, 0] * m[i, 0] + v[:, 1] * m[i, 1] + v[:, 2] * m[i, 2] for i in range(3)], axis=1).astype(F)

    def normal(self, n):  # transform.rs:529-541
        n = np.asarray(n, dtype=F)
        mi = self.m_inv
        if n.ndim == 1:
            return np.array([n[0] * mi[0, i] + n[1] * mi[1, i] + n[2] * mi[2, i] for i in range(3)], dtype=F)
        return np.stack([n[:, 0] * mi[0, i] + n[:, 1] * mi[1, i] + n[:, 2] * mi[2, i] for i in range(3)], axis=1).astype(F)

    @staticmethod
    def translate(d):
        m = np.eye(4, dtype=F); mi = np.eye(4, dtype=F)
        m[:3, 3] = np.asarray(d, dtype=F); mi[:3, 3] = -np.asarray(d, dtype=F)
        return Transform(m, mi)

    @staticmethod
    def scale(x, y, z):
        m = np.diag(np.array([x, y, z, 1], dtype=F)); mi = np.diag(np.array([F(1) / F(x), F(1) / F(y), F(1) / F(z), 1], dtype=F))
        return Transform(m, mi)

    @staticmethod
    def rotate(theta, axis):  # transform.rs:329-355
        a = np.asarray(axis, dtype=F); a = a / np.sqrt((a * a).sum(dtype=F))
        r = F(math.pi / 180.0) * F(theta)
        s, c = F(math.sin(r)), F(math.cos(r))
        m = np.eye(4, dtype=F)
        m[0, 0] = a[0] * a[0] + (1 - a[0] * a[0]) * c; m[0, 1] = a[0] * a[1] * (1 - c) - a[2] * s; m[0, 2] = a[0] * a[2] * (1 - c) + a[1] * s
        m[1, 0] = a[0] * a[1] * (1 - c) + a[2] * s; m[1, 1] = a[1] * a[1] + (1 - a[1] * a[1]) * c; m[1, 2] = a[1] * a[2] * (1 - c) - a[0] * s
        m[2, 0] = a[0] * a[2] * (1 - c) - a[1] * s; m[2, 1] = a[1] * a[2] * (1 - c) + a[0] * s; m[2, 2] = a[2] * a[2] + (1 - a[2] * a[2]) * c
        return Transform(m, m.T.copy())

    @staticmethod
    def look_at(pos, look, up):  # transform.rs:357-393
        pos, look, up = (np.asarray(v, dtype=F) for v in (pos, look, up))
        d = look - pos; d = d / np.sqrt((d * d).sum(dtype=F))
        un = up / np.sqrt((up * up).sum(dtype=F))
        right = np.cross(un.astype(np.float64), d.astype(np.float64)).astype(F)
        right = right / np.sqrt((right * right).sum(dtype=F))
        new_up = np.cross(d.astype(np.float64), right.astype(np.float64)).astype(F)
        c2w = np.eye(4, dtype=F)
        c2w[:3, 0] = right; c2w[:3, 1] = new_up; c2w[:3, 2] = d; c2w[:3, 3] = pos
        return Transform(np.linalg.inv(c2w.astype(np.float64)).astype(F), c2w)

    @staticmethod
    def perspective(fov, n, f):  # transform.rs:399-411
        n, f = F(n), F(f)
        persp = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, f / (f - n), -f * n / (f - n)], [0, 0, 1, 0]], dtype=F)
        inv_tan = F(1) / F(math.tan(F(math.pi / 180.0) * F(fov) / F(2)))
        return Transform.scale(inv_tan, inv_tan, 1) * Transform(persp)


def _matmul(a, b):
    return (a.astype(F) @ b.astype(F)).astype(F)


def box_filter_table(radius=(0.5, 0.5)):
    return np.ones(256, dtype=F)  # filters/boxfilter.rs:19-21


def gaussian_filter_table(radius=(2.0, 2.0), alpha=2.0):  # filters/gaussian.rs:15-36 + film.rs:76-89
    t = np.zeros(256, dtype=F)
    a = F(alpha)
    ex, ey = F(math.exp(-a * F(radius[0]) * F(radius[0]))), F(math.exp(-a * F(radius[1]) * F(radius[1])))
    for y in range(16):
        for x in range(16):
            px = (F(x) + F(0.5)) * F(radius[0]) / F(16)
            py = (F(y) + F(0.5)) * F(radius[1]) / F(16)
            gx = max(F(0), F(math.exp(-a * px * px)) - ex)
            gy = max(F(0), F(math.exp(-a * py * py)) - ey)
            t[y * 16 + x] = F(gx) * F(gy)
    return t


def filter_table(kind, radius, alpha=2.0, B=1.0 / 3.0, Cc=1.0 / 3.0, tau=3.0):
    """Film::new's 16x16 table (film.rs:76-89) of filters/{boxfilter,gaussian,triangle,mitchell,sinc}.rs, each `evaluate` as written
    there (the Mitchell polynomial's `6B*30C` term and the sinc window that is zero INSIDE the radius included)."""
    if kind == "box": return box_filter_table()
    if kind == "gaussian": return gaussian_filter_table(radius, alpha)
    rx, ry = F(radius[0]), F(radius[1])
    if kind == "triangle":
        ev = lambda x, y: max(F(0), rx - abs(x)) * max(F(0), ry - abs(y))
    elif kind == "mitchell":
        B, Cc = F(B), F(Cc)
        def m1(x):
            a = abs(F(2) * x)
            if a > 1: return ((-B - F(6) * Cc) * a * a * a + (F(6) * B * F(30) * Cc) * a * a + (F(-12) * B - F(48) * Cc) * a + (F(8) * B + F(24) * Cc)) * F(1.0 / 6.0)
            return ((F(12) - F(9) * B - F(6) * Cc) * a * a * a + (F(-18) + F(12) * B + F(6) * Cc) * x * x + (F(6) - F(2) * B)) * F(1.0 / 6.0)
        irx, iry = F(1) / rx, F(1) / ry
        ev = lambda x, y: m1(x * irx) * m1(y * iry)
    elif kind == "sinc":
        tau = F(tau)
        def sinc(x):
            y = abs(x)
            return F(1) if y < 1e-5 else F(math.sin(F(math.pi) * y)) / (F(math.pi) * y)
        def ws(x, r):
            y = abs(x)
            if y < r: return F(0)
            return sinc(y) * sinc(y / tau)
        ev = lambda x, y: ws(x, rx) * ws(y, ry)
    else:
        raise ValueError(f"filter {kind!r}")
    t = np.zeros(256, dtype=F)
    for y in range(16):
        for x in range(16):
            t[y * 16 + x] = F(ev((F(x) + F(0.5)) * rx / F(16), (F(y) + F(0.5)) * ry / F(16)))
    return t


class SceneBuilder:
    """Directive-level mirror of core/api.rs for the rows in scope (no parser: SURVEY 8f-2)."""

    def __init__(self):
        self.ctm = Transform()
        self._stack = []
        self.reverse_orientation = False
        self.material_id = None
        self.area_light = None
        self.materials = []
        self.bssrdf_tables = []
        self.textures = []            # PtTexture nodes; named maps as GraphicsState.float_textures / spectrum_textures (api.rs)
        self.images = []              # prepared MIPMap pyramids (textures.prepare_image)
        self.float_textures, self.spectrum_textures = {}, {}
        self.tri_alpha, self.tri_shadow_alpha = [], []
        self.lights = []
        self.P, self.N, self.UV, self.S, self.idx, self.tri_flags = [], [], [], [], [], []
        self.nverts = 0
        self.ntris = 0
        self.spheres = []
        self.prim_shape, self.prim_material, self.prim_light = [], [], []
        self.nprims = 0
        self.env = None
        self.top_refs = []            # RenderOptions.primitives: prim indices / PT_TOP_INSTANCE | instance
        self.objects = {}             # name -> [first_prim, n_prims]
        self.object_list = []
        self.instances = []
        self.current_object = None
        self._tstack = []
        # options block defaults (api.rs:215-241, film.rs:364-398, sobol.rs:120, path.rs:228-249)
        self.film = dict(xres=1280, yres=720, crop=(0.0, 1.0, 0.0, 1.0), scale=1.0, max_lum=float("inf"))
        self.filter = dict(kind="box", radius=(0.5, 0.5), alpha=2.0)
        self.cam = dict(fov=90.0, lensradius=0.0, focaldistance=1e6, shutteropen=0.0, shutterclose=1.0, c2w=Transform())
        self.spp = 16
        self.sampler = "sobol"          # "sobol" | "halton" (Sampler directive; the reference's default is halton, api.rs:215-241)
        self.sample_at_pixel_center = False
        self.integ = dict(maxdepth=5, rrthreshold=1.0, strategy="spatial", pixelbounds=None, kind="path")   # kind: "path" | "volpath"
        # participating media (api.rs:706-722,1219-1253): named homogeneous media, the current MediumInterface, the camera's medium
        self.media = []; self.named_media = {}; self.medium_inside = ""; self.medium_outside = ""; self.camera_medium = None; self._undefined_media = set()
        self._keep = []
        self.prim_med_in = []; self.prim_med_out = []
        self.max_node_prims = 4
        self.split_method = "sah"   # accelerator "bvh" "string splitmethod": "sah" | "hlbvh" (bvh.rs:918-940)
        self.material("matte")  # api.rs:345-361 default material matte Kd .5

    # -- transforms / attributes (api.rs:941-1010,1268-1327)
    def identity(self): self.ctm = Transform()
    def translate(self, x, y, z): self.ctm = self.ctm * Transform.translate((x, y, z))
    def scale(self, x, y, z): self.ctm = self.ctm * Transform.scale(x, y, z)
    def rotate(self, deg, x, y, z): self.ctm = self.ctm * Transform.rotate(deg, (x, y, z))
    def look_at(self, e, l, u): self.ctm = self.ctm * Transform.look_at(e, l, u)
    # AttributeBegin / End push and pop the whole graphics state, named textures included (api.rs:1268-1327: GraphicsState owns
    # float_textures / spectrum_textures); TransformBegin / End only the CTM
    def attribute_begin(self): self._stack.append((self.ctm, self.reverse_orientation, self.material_id, self.area_light, self.medium_inside, self.medium_outside, dict(self.float_textures), dict(self.spectrum_textures)))
    def attribute_end(self): self.ctm, self.reverse_orientation, self.material_id, self.area_light, self.medium_inside, self.medium_outside, self.float_textures, self.spectrum_textures = self._stack.pop()
    def transform_begin(self): self._tstack.append(self.ctm)
    def transform_end(self): self.ctm = self._tstack.pop()

    def toggle_reverse_orientation(self): self.reverse_orientation = not self.reverse_orientation   # ReverseOrientation (api.rs)

    def make_named_medium(self, name, sigma_a=(0.0011, 0.0024, 0.014), sigma_s=(2.55, 3.21, 3.77), g=0.0, scale=1.0, preset="", density=None, p0=(0.0, 0.0, 0.0), p1=(1.0, 1.0, 1.0)):
        """MakeNamedMedium "name" "string type" "homogeneous" | "heterogeneous" (api.rs:680-762): preset from the subsurface table, then
        * scale; `density` (nz, ny, nx) makes it a GridDensityMedium over the box [p0, p1] of the current transform's space."""
        if preset:
            from . import bssrdf as B
            if preset in B.NAMED_MEDIA and sigma_a == (0.0011, 0.0024, 0.014) and sigma_s == (2.55, 3.21, 3.77): sigma_s, sigma_a = B.NAMED_MEDIA[preset]
        m = A.PtMedium()
        m.sigma_a = (C.c_float * 3)(*[float(F(x) * F(scale)) for x in sigma_a]); m.sigma_s = (C.c_float * 3)(*[float(F(x) * F(scale)) for x in sigma_s]); m.g = float(g)
        m.type = A.PT_MEDIUM_HOMOGENEOUS
        if density is not None:   # "heterogeneous" (api.rs:723-752): GridDensityMedium over [p0, p1] of the CTM's space
            d = np.ascontiguousarray(density, dtype=F)
            nz, ny, nx = d.shape          # density[z][y][x]
            med2w = self.ctm * Transform.translate(tuple(float(x) for x in p0)) * Transform.scale(float(p1[0]) - float(p0[0]), float(p1[1]) - float(p0[1]), float(p1[2]) - float(p0[2]))
            m.type = A.PT_MEDIUM_GRID; m.nx, m.ny, m.nz = nx, ny, nz
            m.world_to_medium = (C.c_float * 16)(*med2w.m_inv.flatten())
            m.density = d.ctypes.data_as(A.fp)
            self._keep.append(d)         # the struct points into the array
        self.media.append(m); self.named_media[name] = len(self.media) - 1

    def medium_interface(self, inside="", outside=""):
        """MediumInterface "inside" "outside" (api.rs:1243-1253) keeps the NAMES; "" = no medium. They are looked up when a shape or
        the camera is made (GraphicsState::create_medium_interface, api.rs:382-403)."""
        self.medium_inside = inside; self.medium_outside = outside

    def _medium_index(self, name):
        """api.rs:388-399: an undefined name is reported (error!) and means no medium."""
        if not name: return None
        if name in self.named_media: return self.named_media[name]
        if name not in self._undefined_media:
            self._undefined_media.add(name); print(f'host: error: Named medium "{name}" undefined', file=sys.stderr)
        return None

    def _prim_media(self, n):
        none = A.PT_NONE
        mi, mo = self._medium_index(self.medium_inside), self._medium_index(self.medium_outside)
        self.prim_med_in.append(np.full(n, none if mi is None else mi, dtype=np.uint32))
        self.prim_med_out.append(np.full(n, none if mo is None else mo, dtype=np.uint32))

    # -- options
    def camera(self, fov=90.0, lensradius=0.0, focaldistance=1e6):
        self.cam.update(fov=fov, lensradius=lensradius, focaldistance=focaldistance, c2w=self.ctm.inverse())  # api.rs:1208
        self.camera_medium = self._medium_index(self.medium_outside)   # provisional (what pbrt-v3 does); world_end() takes the state at WorldEnd as api.rs:1738-1741 does

    def world_begin(self): self.ctm = Transform()

    # -- world block
    def material(self, kind, **kw):
        if kind in ("none", ""):   # Material "none" (api.rs:597): shapes made from here on have no material -- medium-interface shells
            self.material_id = None
            return
        m = A.PtMaterial()
        kinds = dict(matte=A.PT_MAT_MATTE, mirror=A.PT_MAT_MIRROR, glass=A.PT_MAT_GLASS, plastic=A.PT_MAT_PLASTIC,
                     metal=A.PT_MAT_METAL, uber=A.PT_MAT_UBER, substrate=A.PT_MAT_SUBSTRATE,
                     subsurface=A.PT_MAT_SUBSURFACE, kdsubsurface=A.PT_MAT_SUBSURFACE, translucent=A.PT_MAT_TRANSLUCENT, mix=A.PT_MAT_MIX, disney=A.PT_MAT_DISNEY)
        m.type = kinds[kind]
        d = dict(  # create_*_material defaults
            matte=dict(Kd=0.5, sigma=0.0), mirror=dict(Kr=0.9), glass=dict(Kr=1.0, Kt=1.0, eta=1.5, uroughness=0.0, vroughness=0.0),
            plastic=dict(Kd=0.25, Ks=0.25, roughness=0.1), metal=dict(roughness=0.01, uroughness=-1.0, vroughness=-1.0),
            uber=dict(Kd=0.25, Ks=0.25, Kr=0.0, Kt=0.0, roughness=0.1, uroughness=-1.0, vroughness=-1.0, opacity=1.0, eta=1.5),
            substrate=dict(Kd=0.5, Ks=0.5, uroughness=0.1, vroughness=0.1),
            translucent=dict(Kd=0.25, Ks=0.25, reflect=0.5, transmit=0.5, roughness=0.1),
            mix=dict(amount=0.5, namedmaterial1=None, namedmaterial2=None),
            # disney.rs:842-887
            disney=dict(color=0.5, metallic=0.0, eta=1.5, roughness=0.5, speculartint=0.0, anisotropic=0.0, sheen=0.0, sheentint=0.5, clearcoat=0.0,
                        clearcoatgloss=1.0, spectrans=0.0, scatterdistance=0.0, thin=False, flatness=0.0, difftrans=0.0),   # mix.rs:52-56; the two materials are given as material ids
   # translucent.rs:82-92 (reflect -> kr, transmit -> kt)
            # subsurface.rs:108-139 / kdsubsurface.rs:106-126
            subsurface=dict(Kr=1.0, Kt=1.0, eta=1.33, uroughness=0.0, vroughness=0.0, scale=1.0, g=0.0, name="",
                            sigma_a=(0.0011, 0.0024, 0.014), sigma_s=(2.55, 3.21, 3.77)),
            kdsubsurface=dict(Kr=1.0, Kt=1.0, eta=1.33, uroughness=0.0, vroughness=0.0, scale=1.0, g=0.0, Kd=0.5, mfp=1.0))[kind]
        d.update(kw)
        # a parameter given as a string names a texture ("texture Kd" "name"); the constant field then keeps the default
        m.tex = (C.c_int32 * 16)(*([-1] * 16))
        slots = dict(Kd=(A.PT_MP_KD, 0), Ks=(A.PT_MP_KS, 0), Kr=(A.PT_MP_KR, 0), Kt=(A.PT_MP_KT, 0), reflect=(A.PT_MP_KR, 0), transmit=(A.PT_MP_KT, 0), amount=(A.PT_MP_KD, 0), color=(A.PT_MP_KD, 0), opacity=(A.PT_MP_OPACITY, 0),
                     eta_rgb=(A.PT_MP_ETA_RGB, 0), k=(A.PT_MP_K_RGB, 0), sigma_a=(A.PT_MP_SIGMA_A, 0), sigma_s=(A.PT_MP_SIGMA_S, 0), mfp=(A.PT_MP_MFP, 0),
                     sigma=(A.PT_MP_SIGMA, 1), roughness=(A.PT_MP_ROUGHNESS, 1), uroughness=(A.PT_MP_U_ROUGHNESS, 1),
                     vroughness=(A.PT_MP_V_ROUGHNESS, 1), eta=(A.PT_MP_ETA, 1), bumpmap=(A.PT_MP_BUMP, 1))
        for key in list(kw):
            if isinstance(kw[key], str) and key in slots:
                slot, is_float = slots[key]
                table = self.float_textures if is_float else self.spectrum_textures
                if kw[key] not in table: raise KeyError(f"texture {kw[key]!r} not declared ({'float' if is_float else 'spectrum'})")
                if kind == "subsurface" and d.get("name") and key in ("sigma_a", "sigma_s"): pass   # (an explicit parameter overrides the named medium's value, subsurface.rs:127-128)
                m.tex[slot] = table[kw[key]]
                d.pop(key)   # keep the create_*_material default in the constant field
        if kind == "metal" and m.tex[A.PT_MP_ROUGHNESS] >= 0:   # metal.rs: uroughness/vroughness fall back to "roughness"
            for sl in (A.PT_MP_U_ROUGHNESS, A.PT_MP_V_ROUGHNESS):
                if m.tex[sl] < 0 and "uroughness" not in kw and "vroughness" not in kw: pass
        three = lambda v: (C.c_float * 3)(*([float(v)] * 3 if np.isscalar(v) else [float(x) for x in v]))
        m.kd = three(d.get("Kd", d.get("amount", d.get("color", 0)))); m.ks = three(d.get("Ks", 0)); m.kr = three(d.get("Kr", d.get("reflect", 0))); m.kt = three(d.get("Kt", d.get("transmit", 0)))
        m.opacity = three(d.get("opacity", 1)); m.eta_rgb = three(d.get("eta_rgb", (0.2, 0.92, 1.1))); m.k_rgb = three(d.get("k", (3.9, 2.45, 2.14)))
        m.sigma = d.get("sigma", 0.0); m.eta = d.get("eta", 1.5); m.roughness = d.get("roughness", 0.1)
        m.u_roughness = d.get("uroughness", -1.0); m.v_roughness = d.get("vroughness", -1.0)
        m.remap_roughness = 1 if d.get("remaproughness", True) else 0
        if m.type == A.PT_MAT_SUBSURFACE:
            from . import bssrdf as B
            g = float(d["g"])
            if kind == "subsurface":
                siga, sigs = d.get("sigma_a", (0.0011, 0.0024, 0.014)), d.get("sigma_s", (2.55, 3.21, 3.77))   # (a textured parameter keeps the default in the constant field)
                if d["name"]:  # subsurface.rs:111-122: a named medium overrides the defaults and forces g = 0
                    if d["name"] in B.NAMED_MEDIA:
                        sigs, siga = B.NAMED_MEDIA[d["name"]]
                        if "sigma_a" in kw: siga = kw["sigma_a"]
                        if "sigma_s" in kw: sigs = kw["sigma_s"]
                        g = 0.0
                table = B.compute_beam_diffusion_bssrdf(g, float(d["eta"]))
                m.scale = float(d["scale"])
            elif m.tex[A.PT_MP_KD] >= 0 or m.tex[A.PT_MP_MFP] >= 0:   # kdsubsurface.rs:96-99 with a textured Kd / mfp: the conversion runs at every hit
                table = B.compute_beam_diffusion_bssrdf(g, float(d["eta"]))
                m.kd_subsurface = 1; m.scale = float(d["scale"])
                m.kd = three(d.get("Kd", 0.5)); m.mfp = three(d.get("mfp", 1.0))
                siga, sigs = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)
            else:  # kdsubsurface.rs:96-99: mfp * scale, then subsurface_from_diffuse (constant textures -> host side)
                table = B.compute_beam_diffusion_bssrdf(g, float(d["eta"]))
                three_np = lambda v: np.asarray([v] * 3 if np.isscalar(v) else v, dtype=F)
                mfree = np.maximum(three_np(d["mfp"]), F(0)) * F(d["scale"])
                siga, sigs = B.subsurface_from_diffuse(table, np.maximum(three_np(d["Kd"]), F(0)), mfree)
                m.scale = 1.0
            m.sigma_a = three(siga); m.sigma_s = three(sigs)
            for i, t in enumerate(self.bssrdf_tables):
                if t is table: m.bssrdf_table = i; break
            else:
                self.bssrdf_tables.append(table); m.bssrdf_table = len(self.bssrdf_tables) - 1
        if kind == "disney":
            sdv = np.asarray([d["scatterdistance"]] * 3 if np.isscalar(d["scatterdistance"]) else d["scatterdistance"], dtype=F)
            m.disney_scatter = (C.c_float * 3)(*[float(x) for x in sdv])
            if np.any(sdv != 0) and isinstance(kw.get("color"), str): raise NotImplementedError("disney: textured color together with scatterdistance")
            names = ("metallic", "speculartint", "anisotropic", "sheen", "sheentint", "clearcoat", "clearcoatgloss", "spectrans", "flatness", "difftrans")
            if any(isinstance(d[n], str) for n in names): raise NotImplementedError("textured disney parameters other than color / eta / roughness")
            m.disney = (C.c_float * 10)(*[float(d[n]) for n in names]); m.disney_thin = 1 if d["thin"] else 0
        if kind == "mix":
            ids = (d["namedmaterial1"], d["namedmaterial2"])
            for i in ids:
                if not isinstance(i, int) or not (0 <= i < len(self.materials)) or self.materials[i].type in (A.PT_MAT_MIX, A.PT_MAT_SUBSURFACE): raise ValueError("mix needs the ids of two plain materials")
            m.mix = (C.c_uint32 * 2)(*ids)
            m.tex[A.PT_MP_BUMP] = self.materials[ids[0]].tex[A.PT_MP_BUMP]   # mix.rs:31-45: only material 1's bump map survives
        self.materials.append(m)
        self.material_id = len(self.materials) - 1

    # -- textures (api.rs pbrt_texture; textures/*.rs create_* functions)
    def _const_tex(self, value):
        t = A.PtTexture(); t.type = A.PT_TEX_CONSTANT; t.child = (C.c_int32 * 3)(-1, -1, -1)
        v = [float(value)] * 3 if np.isscalar(value) else [float(x) for x in value]
        t.value = (C.c_float * 3)(*v)
        self.textures.append(t); return len(self.textures) - 1

    def _child(self, v, is_float):
        if isinstance(v, str):
            table = self.float_textures if is_float else self.spectrum_textures
            return table[v]
        return self._const_tex(v)

    def _mapping(self, t, kw):   # get_mapping2d (texture.rs:439-466); texture-to-world = CTM at the Texture directive
        kind = kw.get("mapping", "uv")
        t.mapping = dict(uv=A.PT_MAP_UV, planar=A.PT_MAP_PLANAR, spherical=A.PT_MAP_SPHERICAL, cylindrical=A.PT_MAP_CYLINDRICAL)[kind]
        t.su, t.sv = float(kw.get("uscale", 1.0)), float(kw.get("vscale", 1.0))
        t.du, t.dv = float(kw.get("udelta", 0.0)), float(kw.get("vdelta", 0.0))
        t.vs = (C.c_float * 3)(*[float(x) for x in kw.get("v1", (1, 0, 0))]); t.vt = (C.c_float * 3)(*[float(x) for x in kw.get("v2", (0, 1, 0))])
        t.world_to_texture = (C.c_float * 16)(*self.ctm.m_inv.flatten())

    def texture(self, name, kind, cls, **kw):
        """Texture "name" "color|spectrum|float" "class" params.  Image maps take `pixels` (h, w, 3; top row first, as
        read_image returns) instead of a filename."""
        is_float = kind == "float"
        t = A.PtTexture(); t.child = (C.c_int32 * 3)(-1, -1, -1)
        one = 1.0
        if cls == "constant":
            t.type = A.PT_TEX_CONSTANT
            v = kw.get("value", 1.0); t.value = (C.c_float * 3)(*([float(v)] * 3 if np.isscalar(v) else [float(x) for x in v]))
        elif cls == "scale":
            t.type = A.PT_TEX_SCALE
            t.child = (C.c_int32 * 3)(self._child(kw.get("tex1", one), is_float), self._child(kw.get("tex2", one), is_float), -1)
        elif cls == "mix":
            t.type = A.PT_TEX_MIX
            t.child = (C.c_int32 * 3)(self._child(kw.get("tex1", 0.0), is_float), self._child(kw.get("tex2", 1.0), is_float), self._child(kw.get("amount", 0.5), True))
        elif cls == "checkerboard":
            dim = int(kw.get("dimension", 2))
            t.child = (C.c_int32 * 3)(self._child(kw.get("tex1", 1.0), is_float), self._child(kw.get("tex2", 0.0), is_float), -1)
            if dim == 2:
                t.type = A.PT_TEX_CHECKERBOARD2D; self._mapping(t, kw)
                t.aa_closedform = 0 if kw.get("aamode", "none") == "none" else 1
            else:
                t.type = A.PT_TEX_CHECKERBOARD3D; t.world_to_texture = (C.c_float * 16)(*self.ctm.m_inv.flatten())
        elif cls == "imagemap":
            from . import textures as T
            t.type = A.PT_TEX_IMAGEMAP; self._mapping(t, kw)
            wrap = kw.get("wrap", "repeat")
            img = T.prepare_image(kw["pixels"], scale=float(kw.get("scale", 1.0)), gamma=bool(kw.get("gamma", False)),
                                  channels=1 if is_float else 3, wrap=wrap)
            self.images.append(img); t.image = len(self.images) - 1
            t.trilinear = 1 if kw.get("trilinear", False) else 0
            t.max_anisotropy = float(kw.get("maxanisotropy", 8.0))
            t.wrap = A.PT_WRAP_REPEAT if wrap == "repeat" else A.PT_WRAP_BLACK
        elif cls == "uv":
            if is_float: raise ValueError("uv textures are spectrum-only (textures/uv.rs:36-38)")
            t.type = A.PT_TEX_UV; self._mapping(t, kw)
        elif cls == "bilerp":
            t.type = A.PT_TEX_BILERP; self._mapping(t, kw)
            for nm, dv in (("v00", 0.0), ("v01", 1.0), ("v10", 0.0), ("v11", 1.0)):
                v = kw.get(nm, dv); setattr(t, nm, (C.c_float * 3)(*([float(v)] * 3 if np.isscalar(v) else [float(x) for x in v])))
        elif cls in ("fbm", "wrinkled", "windy", "marble"):   # IdentityMapping3D(texture-to-world CTM)
            if cls == "marble" and is_float: raise ValueError("marble textures are spectrum-only (textures/marble.rs:68-70)")
            t.type = dict(fbm=A.PT_TEX_FBM, wrinkled=A.PT_TEX_WRINKLED, windy=A.PT_TEX_WINDY, marble=A.PT_TEX_MARBLE)[cls]
            t.world_to_texture = (C.c_float * 16)(*self.ctm.m_inv.flatten())
            t.octaves = int(kw.get("octaves", 8)); t.omega = float(kw.get("roughness", 0.5))
            t.marble_scale = float(kw.get("scale", 1.0)); t.variation = float(kw.get("variation", 0.2))
        elif cls == "dots":
            t.type = A.PT_TEX_DOTS; self._mapping(t, kw)
            t.child = (C.c_int32 * 3)(self._child(kw.get("outside", 0.0), is_float), self._child(kw.get("inside", 1.0), is_float), -1)
        else:
            raise NotImplementedError(f"texture class {cls!r}")
        self.textures.append(t)
        (self.float_textures if is_float else self.spectrum_textures)[name] = len(self.textures) - 1

    def area_light_source(self, L=(1, 1, 1), twosided=False):
        self.area_light = dict(L=tuple(float(x) for x in L), twosided=twosided)

    def light_source(self, kind, **kw):
        l = A.PtLight()
        l2w = self.ctm
        l.light_to_world = (C.c_float * 16)(*l2w.m.flatten()); l.world_to_light = (C.c_float * 16)(*l2w.m_inv.flatten())
        l.prim = A.PT_NONE
        sc = kw.get("scale", 1.0)
        if kind == "distant":  # distant.rs:124-132
            L = np.asarray(kw.get("L", (1, 1, 1)), dtype=F) * F(sc)
            frm, to = np.asarray(kw.get("from_", (0, 0, 0)), dtype=F), np.asarray(kw.get("to", (0, 0, 1)), dtype=F)
            w = l2w.vector(frm - to); w = w / np.sqrt((w * w).sum(dtype=F))
            l.type = A.PT_LIGHT_DISTANT; l.L = (C.c_float * 3)(*L); l.dir = (C.c_float * 3)(*w)
        elif kind == "point":  # point.rs:99-106 (translation quirk P.x,P.y,P.x -- App. A #15)
            I = np.asarray(kw.get("I", (1, 1, 1)), dtype=F) * F(sc)
            p = np.asarray(kw.get("from_", (0, 0, 0)), dtype=F)
            t = l2w * Transform.translate((p[0], p[1], p[0]))
            l.type = A.PT_LIGHT_POINT; l.L = (C.c_float * 3)(*I); l.pos = (C.c_float * 3)(*t.point((0, 0, 0)))
        elif kind == "spot":  # spot.rs:118-147
            I = np.asarray(kw.get("I", (1, 1, 1)), dtype=F) * F(sc)
            frm, to = np.asarray(kw.get("from_", (0, 0, 0)), dtype=F), np.asarray(kw.get("to", (0, 0, 1)), dtype=F)
            coneangle, conedelta = F(kw.get("coneangle", 30.0)), F(kw.get("conedeltaangle", 5.0))
            d = to - frm; d = d / np.sqrt((d * d).sum(dtype=F))
            if abs(d[0]) > abs(d[1]): du = np.array([-d[2], 0, d[0]], dtype=F) / np.sqrt(d[0] * d[0] + d[2] * d[2])
            else: du = np.array([0, d[2], -d[1]], dtype=F) / np.sqrt(d[1] * d[1] + d[2] * d[2])
            dv = np.cross(d.astype(np.float64), du.astype(np.float64)).astype(F)
            mat = np.eye(4, dtype=F); mat[0, :3] = du; mat[1, :3] = dv; mat[2, :3] = d
            t = l2w * Transform.translate(frm) * Transform(mat).inverse()
            l.type = A.PT_LIGHT_SPOT; l.L = (C.c_float * 3)(*I); l.pos = (C.c_float * 3)(*t.point((0, 0, 0)))
            l.cos_total_width = math.cos(F(math.pi / 180.0) * coneangle); l.cos_falloff_start = math.cos(F(math.pi / 180.0) * (coneangle - conedelta))
            l.light_to_world = (C.c_float * 16)(*t.m.flatten()); l.world_to_light = (C.c_float * 16)(*t.m_inv.flatten())
        elif kind == "infinite":  # infinite.rs:243-259, constant-L map = 1x1 texel
            L = np.asarray(kw.get("L", (1, 1, 1)), dtype=F) * F(sc)
            l.type = A.PT_LIGHT_INFINITE
            tex = kw.get("texels")   # rows = v (theta), columns = u (phi), as read_image returns them
            if tex is None:
                tex = L.reshape(1, 1, 3)
            else:
                tex = (np.asarray(tex, dtype=F) * L.reshape(1, 1, 3)).astype(F)   # infinite.rs:46-50: texels *= L * scale
            tex = np.ascontiguousarray(tex, dtype=F)
            eh, ew, _ = tex.shape
            if (ew & (ew - 1)) or (eh & (eh - 1)):   # MIPMap::new resamples to the next powers of two (mipmap.rs:81-140);
                from .textures import build_mipmap   # `le`, the importance image and `power` all read that pyramid (infinite.rs:60-66)
                tex = np.ascontiguousarray(build_mipmap(tex, "repeat")[0][0], dtype=F)
            self.env = dict(texels=tex, importance=_env_importance(tex), power_lookup=_env_power_lookup(tex))
        else:
            raise ValueError(kind)
        self.lights.append(l)

    def _new_area_light(self, prim_index):  # api.rs:1531-1546: one DiffuseAreaLight per shape
        l = A.PtLight(); l.type = A.PT_LIGHT_DIFFUSE_AREA
        l.L = (C.c_float * 3)(*self.area_light["L"]); l.two_sided = 1 if self.area_light["twosided"] else 0
        l.prim = prim_index
        l.light_to_world = (C.c_float * 16)(*self.ctm.m.flatten()); l.world_to_light = (C.c_float * 16)(*self.ctm.m_inv.flatten())
        self.lights.append(l)
        return len(self.lights) - 1

    def trianglemesh(self, P, indices, N=None, UV=None, S=None, alpha=None, shadowalpha=None):
        """shapes/triangle.rs:21-73: vertices pre-transformed to world space, one primitive per triangle."""
        P = np.asarray(P, dtype=F).reshape(-1, 3); idx = np.asarray(indices, dtype=np.uint32).reshape(-1, 3)
        nv, nt = len(P), len(idx)
        self.P.append(self.ctm.point(P))
        self.N.append(None if N is None else self.ctm.normal(np.asarray(N, dtype=F).reshape(-1, 3)))
        self.S.append(None if S is None else self.ctm.vector(np.asarray(S, dtype=F).reshape(-1, 3)))
        self.UV.append(None if UV is None else np.asarray(UV, dtype=F).reshape(-1, 2))
        self.idx.append(idx + np.uint32(self.nverts))
        fl = (A.PT_TRI_REVERSE_ORIENTATION if self.reverse_orientation else 0) | (A.PT_TRI_SWAPS_HANDEDNESS if self.ctm.swaps_handedness() else 0)
        fl |= (A.PT_TRI_HAS_N if N is not None else 0) | (A.PT_TRI_HAS_S if S is not None else 0) | (A.PT_TRI_HAS_UV if UV is not None else 0)
        self.tri_flags.append(np.full(nt, fl, dtype=np.uint8))
        # "alpha" / "shadowalpha" (triangle.rs:727-756): a float texture name, or the constant 0 (=> ConstantTexture(0))
        def mask(v):
            if v is None: return -1
            if isinstance(v, str): return self.float_textures[v]
            return self._const_tex(0.0) if float(v) == 0.0 else -1
        self.tri_alpha.append(np.full(nt, mask(alpha), dtype=np.int32)); self.tri_shadow_alpha.append(np.full(nt, mask(shadowalpha), dtype=np.int32))
        first_prim = self.nprims
        self.prim_shape.append((np.uint32(A.PT_SHAPE_TRIANGLE << 30) | (np.arange(nt, dtype=np.uint32) + np.uint32(self.ntris))).astype(np.uint32))
        mid = A.PT_NONE if self.material_id is None else self.material_id
        self.prim_material.append(np.full(nt, mid, dtype=np.uint32))
        self._prim_media(nt)
        if self.area_light is None or self.current_object is not None:  # api.rs:1605-1608: area lights inside instances are dropped
            self.prim_light.append(np.full(nt, A.PT_NONE, dtype=np.uint32))
        else:
            self.prim_light.append(np.array([self._new_area_light(first_prim + t) for t in range(nt)], dtype=np.uint32))
        if self.current_object is None: self.top_refs.append(np.arange(first_prim, first_prim + nt, dtype=np.uint32))
        else: self.objects[self.current_object][1] += nt
        self.nverts += nv; self.ntris += nt; self.nprims += nt
        return first_prim

    def sphere(self, radius=1.0, zmin=None, zmax=None, phimax=360.0):
        """shapes/sphere.rs:31-50,424-431."""
        r = F(radius)
        zmin = -r if zmin is None else F(zmin); zmax = r if zmax is None else F(zmax)
        clampf = lambda v, lo, hi: lo if v < lo else (hi if v > hi else v)
        s = A.PtSphere()
        s.object_to_world = (C.c_float * 16)(*self.ctm.m.flatten()); s.world_to_object = (C.c_float * 16)(*self.ctm.m_inv.flatten())
        s.radius = r
        s.z_min = clampf(min(zmin, zmax), -r, r); s.z_max = clampf(max(zmin, zmax), -r, r)
        s.theta_min = math.acos(clampf(min(zmin, zmax) / r, F(-1), F(1))); s.theta_max = math.acos(clampf(max(zmin, zmax) / r, F(-1), F(1)))
        s.phi_max = F(math.pi / 180.0) * F(clampf(F(phimax), F(0), F(360)))
        s.reverse_orientation = 1 if self.reverse_orientation else 0
        s.transform_swaps_handedness = 1 if self.ctm.swaps_handedness() else 0
        self.spheres.append(s)
        first_prim = self.nprims
        self.prim_shape.append(np.array([(A.PT_SHAPE_SPHERE << 30) | (len(self.spheres) - 1)], dtype=np.uint32))
        self.prim_material.append(np.array([A.PT_NONE if self.material_id is None else self.material_id], dtype=np.uint32))
        self._prim_media(1)
        self.prim_light.append(np.array([A.PT_NONE if (self.area_light is None or self.current_object is not None) else self._new_area_light(first_prim)], dtype=np.uint32))
        if self.current_object is None: self.top_refs.append(np.array([first_prim], dtype=np.uint32))
        else: self.objects[self.current_object][1] += 1
        self.nprims += 1
        return first_prim

    # -- instancing (api.rs:1630-1713)
    def disk(self, height=0.0, radius=1.0, innerradius=0.0, phimax=360.0):
        """shapes/disk.rs:17-43,175-189: stored in the sphere table with kind = PT_QUADRIC_DISK (z_min = z_max = height)."""
        s = A.PtSphere()
        s.object_to_world = (C.c_float * 16)(*self.ctm.m.flatten()); s.world_to_object = (C.c_float * 16)(*self.ctm.m_inv.flatten())
        s.kind = A.PT_QUADRIC_DISK; s.radius = F(radius); s.inner_radius = F(innerradius); s.z_min = s.z_max = F(height)
        s.phi_max = F(math.pi / 180.0) * F(min(max(F(phimax), F(0)), F(360)))
        s.reverse_orientation = 1 if self.reverse_orientation else 0
        s.transform_swaps_handedness = 1 if self.ctm.swaps_handedness() else 0
        self.spheres.append(s)
        first_prim = self.nprims
        self.prim_shape.append(np.array([(A.PT_SHAPE_SPHERE << 30) | (len(self.spheres) - 1)], dtype=np.uint32))
        self.prim_material.append(np.array([A.PT_NONE if self.material_id is None else self.material_id], dtype=np.uint32))
        self._prim_media(1)
        self.prim_light.append(np.array([A.PT_NONE if (self.area_light is None or self.current_object is not None) else self._new_area_light(first_prim)], dtype=np.uint32))
        if self.current_object is None: self.top_refs.append(np.array([first_prim], dtype=np.uint32))
        else: self.objects[self.current_object][1] += 1
        self.nprims += 1
        return first_prim

    def object_begin(self, name):
        self.attribute_begin()
        self.objects[name] = [self.nprims, 0]
        self.current_object = name

    def object_end(self):
        self.current_object = None
        self.attribute_end()

    def object_instance(self, name):
        first, n = self.objects[name]
        if n == 0:
            return
        if name not in [o[0] for o in self.object_list]:
            self.object_list.append((name, first, n))
        oid = [o[0] for o in self.object_list].index(name)
        inst = A.PtInstance(); inst.object = oid
        inst.instance_to_world = (C.c_float * 16)(*self.ctm.m.flatten()); inst.world_to_instance = (C.c_float * 16)(*self.ctm.m_inv.flatten())
        self.instances.append(inst)
        self.top_refs.append(np.array([A.PT_TOP_INSTANCE | (len(self.instances) - 1)], dtype=np.uint32))

    # -- WorldEnd (api.rs:1715-1748): flatten to the C ABI structs
    def world_end(self):
        self.camera_medium = self._medium_index(self.medium_outside)   # api.rs:1738-1741: create_medium_interface() on the graphics state AT WorldEnd, camera = mi.outside (api.rs:830)
        return SceneData(self), self.render_params()

    def render_params(self):
        rp = A.PtRenderParams()
        xres, yres = self.film["xres"], self.film["yres"]
        cw = self.film["crop"]
        rp.full_resolution = (C.c_int32 * 2)(xres, yres)
        crop = [math.ceil(F(xres) * F(cw[0])), math.ceil(F(yres) * F(cw[2])), math.ceil(F(xres) * F(cw[1])), math.ceil(F(yres) * F(cw[3]))]  # film.rs:57-66
        rp.cropped_pixel_bounds = (C.c_int32 * 4)(*crop)
        rx, ry = self.filter["radius"]
        rp.filter_radius = (C.c_float * 2)(rx, ry)
        table = filter_table(self.filter["kind"], (rx, ry), alpha=self.filter.get("alpha", 2.0), B=self.filter.get("B", 1.0 / 3.0), Cc=self.filter.get("C", 1.0 / 3.0), tau=self.filter.get("tau", 3.0))
        rp.filter_table = (C.c_float * 256)(*table)
        rp.max_sample_luminance = self.film["max_lum"]; rp.scale = self.film["scale"]
        rp.spp = self.spp
        rp.sampler_type = {"sobol": A.PT_SAMPLER_SOBOL, "halton": A.PT_SAMPLER_HALTON}[self.sampler]
        rp.sample_at_pixel_center = 1 if (self.sample_at_pixel_center and self.sampler == "halton") else 0   # a HaltonSampler parameter (halton.rs:226-236)
        sb = [math.floor(F(crop[0]) + F(0.5) - F(rx)), math.floor(F(crop[1]) + F(0.5) - F(ry)),
              math.ceil(F(crop[2]) - F(0.5) + F(rx)), math.ceil(F(crop[3]) - F(0.5) + F(ry))]  # film.rs:104-112
        rp.sample_bounds = (C.c_int32 * 4)(*sb)
        # PerspectiveCamera::new, perspective.rs:40-86 + create_perspective_camera :298-356
        frame = F(xres) / F(yres)
        if frame > 1: sw = (-frame, frame, F(-1), F(1))
        else: sw = (F(-1), F(1), F(-1) / frame, F(1) / frame)
        c2s = Transform.perspective(self.cam["fov"], 1e-2, 1000.0)
        s2r = (Transform.scale(xres, yres, 1) * Transform.scale(F(1) / (sw[1] - sw[0]), F(1) / (sw[2] - sw[3]), 1) *
               Transform.translate((-sw[0], -sw[3], 0)))
        r2c = c2s.inverse() * s2r.inverse()
        rp.raster_to_camera = (C.c_float * 16)(*r2c.m.flatten())
        rp.camera_to_world = (C.c_float * 16)(*self.cam["c2w"].m.flatten())
        rp.lens_radius = self.cam["lensradius"]; rp.focal_distance = self.cam["focaldistance"]
        rp.shutter_open = self.cam["shutteropen"]; rp.shutter_close = self.cam["shutterclose"]
        rp.max_depth = self.integ["maxdepth"]; rp.rr_threshold = self.integ["rrthreshold"]
        rp.integrator = {"path": A.PT_INTEGRATOR_PATH, "volpath": A.PT_INTEGRATOR_VOLPATH}[self.integ.get("kind", "path")]
        rp.camera_medium = A.PT_NONE if self.camera_medium is None else self.camera_medium
        pb = self.integ["pixelbounds"]
        if pb is None: pb = sb
        else: pb = [max(pb[0], sb[0]), max(pb[2], sb[1]), min(pb[1], sb[2]), min(pb[3], sb[3])]  # path.rs:233-246
        rp.pixel_bounds = (C.c_int32 * 4)(*pb)
        rp.light_strategy = dict(uniform=A.PT_LS_UNIFORM, power=A.PT_LS_POWER, spatial=A.PT_LS_SPATIAL, spatial_eager=A.PT_LS_SPATIAL_EAGER, spatial_lazy=A.PT_LS_SPATIAL_LAZY)[self.integ["strategy"]]
        rp.tile_rank, rp.tile_world, rp.spp_per_pass, rp.profile = 0, 1, 0, 0
        return rp


def _env_power_lookup(tex):
    """`map.lookup((.5, .5), .5)` of InfiniteAreaLight::power (infinite.rs:103-109): MIPMap::lookup with width 0.5 is
    level = levels - 2 (exactly, delta = 0) -> `triangle(levels - 2, st)`; the texel for a 1x1 map (mipmap.rs:202-223)."""
    from . import textures as T
    levels, w, h = T.build_mipmap(tex, "repeat")
    n = len(levels)
    if n == 1: return levels[0][0, 0].astype(F)
    lv = levels[n - 2]; lh, lw, _ = lv.shape
    s = F(F(0.5) * F(lw) - F(0.5)); t = F(F(0.5) * F(lh) - F(0.5))
    s0, t0 = int(math.floor(s)), int(math.floor(t)); ds, dt = F(s - F(s0)), F(t - F(t0))
    tx = lambda a, b: lv[b % lh, a % lw].astype(F)
    r = (tx(s0, t0) * F(F(1 - ds) * F(1 - dt)) + tx(s0, t0 + 1) * F(F(1 - ds) * dt)).astype(F)
    r = (r + tx(s0 + 1, t0) * F(ds * F(1 - dt))).astype(F)
    return (r + tx(s0 + 1, t0 + 1) * F(ds * dt)).astype(F)


def _env_importance(tex):
    """lights/infinite.rs:62-81 importance image (2w x 2h): `map.lookup(st, fwidth).y() * sin(theta)` with
    fwidth = 0.5 / min(2w, 2h).  MIPMap::lookup (mipmap.rs:202-223) picks level = levels - 1 + log2(fwidth)
    = log2(max(w,h)/min(w,h)) - 2: negative for power-of-two maps with aspect <= 2:1 -> `triangle(0, st)`, the level-0 bilinear
    lookup with Repeat wrap (mipmap.rs:295-327); for wider maps `lerp(delta, triangle(ilevel), triangle(ilevel + 1))` on the pyramid
    (delta is exactly 0 for power-of-two sizes; the arithmetic is kept as written).  Non-power-of-two maps arrive here already
    resampled (`light_source`), so `tex` is level 0 of the reference's pyramid."""
    from . import textures as T
    h, w, _ = tex.shape
    pow2 = lambda n: n > 0 and (n & (n - 1)) == 0
    if not (pow2(w) and pow2(h)):
        raise ValueError("environment map must be resampled to powers of two first")
    levels, _, _ = T.build_mipmap(tex, "repeat")
    W, H = 2 * w, 2 * h
    y_w = np.array([0.212671, 0.715160, 0.072169], dtype=F)
    up = ((np.arange(W, dtype=F) + F(0.5)) / F(W)).astype(F)
    vp = ((np.arange(H, dtype=F) + F(0.5)) / F(H)).astype(F)
    one = F(1.0)

    def triangle(lv):
        lv = min(max(lv, 0), len(levels) - 1)
        L = levels[lv]; lh, lw, _ = L.shape
        sx = (up * F(lw) - F(0.5)).astype(F); ty = (vp * F(lh) - F(0.5)).astype(F)
        s0 = np.floor(sx).astype(np.int64); t0 = np.floor(ty).astype(np.int64)
        ds = (sx - s0.astype(F)).astype(F)[None, :]; dt = (ty - t0.astype(F)).astype(F)[:, None]
        tx = lambda si, ti: L[np.mod(ti, lh)[:, None], np.mod(si, lw)[None, :]].astype(F)   # texel(level, s, t), Repeat
        # RGBSpectrum arithmetic first (tmp4 + tmp3 + tmp2 + tmp1), then y(), exactly as `triangle(..).y()`
        rgb = ((tx(s0, t0) * ((one - ds) * (one - dt))[..., None] + tx(s0, t0 + 1) * ((one - ds) * dt)[..., None]).astype(F)
               + tx(s0 + 1, t0) * (ds * (one - dt))[..., None]).astype(F)
        return (rgb + tx(s0 + 1, t0 + 1) * (ds * dt)[..., None]).astype(F)

    fwidth = F(F(0.5) / F(min(W, H)))
    level = F(F(len(levels) - 1) + F(np.log2(max(fwidth, F(1.0e-8)))))
    if level < 0: rgb = triangle(0)
    elif level >= len(levels) - 1: rgb = np.broadcast_to(levels[-1][0, 0].astype(F), (H, W, 3)).copy()
    else:
        il = int(np.floor(level)); delta = F(level - F(il))
        rgb = (triangle(il) * F(one - delta) + triangle(il + 1) * delta).astype(F)
    img = ((y_w[0] * rgb[..., 0] + y_w[1] * rgb[..., 1]).astype(F) + y_w[2] * rgb[..., 2]).astype(F)
    sin_theta = np.sin(F(math.pi) * (np.arange(H, dtype=F) + F(0.5)) / F(H)).astype(F)
    return np.ascontiguousarray(img * sin_theta[:, None], dtype=F)


class SceneData:
    """Owns the numpy arrays behind a PtSceneDesc (keeps them alive for the C call)."""

    def __init__(self, b):
        cat = lambda parts, width, dt: (np.ascontiguousarray(np.concatenate(parts), dtype=dt) if parts else np.zeros((0, width), dtype=dt))
        self.P = cat(b.P, 3, F)
        self.idx = cat(b.idx, 3, np.uint32)
        self.tri_flags = np.ascontiguousarray(np.concatenate(b.tri_flags), dtype=np.uint8) if b.tri_flags else np.zeros(0, np.uint8)

        def opt(parts, width):
            if all(p is None for p in parts): return None
            out = np.zeros((b.nverts, width), dtype=F); o = 0
            for p, pp in zip(parts, b.P):
                if p is not None: out[o:o + len(pp)] = p
                o += len(pp)
            return np.ascontiguousarray(out)
        self.N, self.S, self.UV = opt(b.N, 3), opt(b.S, 3), opt(b.UV, 2)
        c1 = lambda parts: np.ascontiguousarray(np.concatenate(parts), dtype=np.uint32) if parts else np.zeros(0, np.uint32)
        self.prim_shape, self.prim_material, self.prim_light = c1(b.prim_shape), c1(b.prim_material), c1(b.prim_light)
        self.materials = (A.PtMaterial * max(1, len(b.materials)))(*b.materials)
        self.lights = (A.PtLight * max(1, len(b.lights)))(*b.lights)
        self.spheres = (A.PtSphere * max(1, len(b.spheres)))(*b.spheres)
        self.n_materials, self.n_lights, self.n_spheres = len(b.materials), len(b.lights), len(b.spheres)
        self.env = b.env
        self.media = (A.PtMedium * max(1, len(b.media)))(*b.media); self.n_media = len(b.media); self._keep = b._keep   # (grid media point into numpy arrays)
        self.prim_med_in = c1(b.prim_med_in) if b.media else None; self.prim_med_out = c1(b.prim_med_out) if b.media else None
        self.max_node_prims = b.max_node_prims
        self.split_method = {"sah": A.PT_SPLIT_SAH, "hlbvh": A.PT_SPLIT_HLBVH}[b.split_method]
        self.nodes = None; self.ordered = None
        self.n_objects, self.n_instances = len(b.object_list), len(b.instances)
        self.objects = (A.PtObject * max(1, self.n_objects))(*[A.PtObject(f, n) for _, f, n in b.object_list])
        self.instances = (A.PtInstance * max(1, self.n_instances))(*b.instances)
        self.top_refs = np.ascontiguousarray(np.concatenate(b.top_refs), dtype=np.uint32) if (b.instances and b.top_refs) else None
        ca = lambda parts: np.ascontiguousarray(np.concatenate(parts), dtype=np.int32) if parts else np.zeros(0, np.int32)
        self.tri_alpha, self.tri_shadow_alpha = ca(b.tri_alpha), ca(b.tri_shadow_alpha)
        if not (self.tri_alpha >= 0).any(): self.tri_alpha = None
        if not (self.tri_shadow_alpha >= 0).any(): self.tri_shadow_alpha = None
        self.n_textures = len(b.textures)
        self.textures = (A.PtTexture * max(1, self.n_textures))(*b.textures)
        self.image_src = list(b.images)
        self.images = (A.PtImage * max(1, len(self.image_src)))()
        for i, im in enumerate(self.image_src):
            e = self.images[i]
            e.width, e.height, e.n_levels, e.channels = im["width"], im["height"], im["n_levels"], im["channels"]
            e.texels = im["texels"].ctypes.data_as(A.fp)
        if self.image_src:
            from . import textures as T
            self.ewa_lut = T.ewa_weight_lut()
        else:
            self.ewa_lut = None
        self.bssrdf_src = list(b.bssrdf_tables)
        self.bssrdf_tables = (A.PtBSSRDFTable * max(1, len(self.bssrdf_src)))()
        for i, t in enumerate(self.bssrdf_src):
            e = self.bssrdf_tables[i]
            e.n_rho, e.n_radius = t.n_rho, t.n_radius
            e.rho_samples = t.rho_samples.ctypes.data_as(A.fp); e.radius_samples = t.radius_samples.ctypes.data_as(A.fp)
            e.profile = t.profile.ctypes.data_as(A.fp); e.rhoeff = t.rhoeff.ctypes.data_as(A.fp)
            e.profile_cdf = t.profile_cdf.ctypes.data_as(A.fp)

    def set_bvh(self, nodes, ordered):
        """Adopt a prebuilt accelerator (what a Rust host would pass: BVHAccel.nodes / ordered prims)."""
        self.nodes, self.ordered = nodes, np.ascontiguousarray(ordered, dtype=np.uint32)

    def desc(self):
        d = A.PtSceneDesc()
        ptr = lambda a, t: None if a is None or a.size == 0 else a.ctypes.data_as(t)
        d.n_vertices = len(self.P); d.P = ptr(self.P, A.fp); d.N = ptr(self.N, A.fp); d.S = ptr(self.S, A.fp); d.UV = ptr(self.UV, A.fp)
        d.n_triangles = len(self.idx); d.indices = ptr(self.idx, A.u32p); d.tri_flags = ptr(self.tri_flags, A.u8p)
        d.n_spheres = self.n_spheres; d.spheres = self.spheres
        d.n_prims = len(self.prim_shape)
        d.prim_shape = ptr(self.prim_shape, A.u32p); d.prim_material = ptr(self.prim_material, A.u32p); d.prim_light = ptr(self.prim_light, A.u32p)
        d.n_materials = self.n_materials; d.materials = self.materials
        d.n_lights = self.n_lights; d.lights = self.lights
        if self.env is not None:
            t = self.env["texels"]
            d.env_height, d.env_width = t.shape[0], t.shape[1]
            d.env_texels = ptr(t, A.fp); d.env_importance = ptr(self.env["importance"], A.fp)
            d.env_power_lookup = (C.c_float * 3)(*[float(x) for x in self.env["power_lookup"]])
        d.max_node_prims = self.max_node_prims
        d.split_method = self.split_method
        if self.n_media:
            d.n_media = self.n_media; d.media = self.media
            d.prim_medium_inside = ptr(self.prim_med_in, A.u32p); d.prim_medium_outside = ptr(self.prim_med_out, A.u32p)
        if self.nodes is not None:
            d.n_nodes = len(self.nodes); d.nodes = self.nodes; d.ordered_prims = ptr(self.ordered, A.u32p)
        if self.top_refs is not None:
            d.n_objects = self.n_objects; d.objects = self.objects; d.n_instances = self.n_instances; d.instances = self.instances
            d.n_top = len(self.top_refs); d.top_refs = ptr(self.top_refs, A.u32p)
        d.n_bssrdf_tables = len(self.bssrdf_src); d.bssrdf_tables = self.bssrdf_tables
        d.n_textures = self.n_textures; d.textures = self.textures
        d.tri_alpha = ptr(self.tri_alpha, A.i32p); d.tri_shadow_alpha = ptr(self.tri_shadow_alpha, A.i32p)
        d.n_images = len(self.image_src); d.images = self.images
        d.ewa_weight_lut = ptr(self.ewa_lut, A.fp)
        return d


# ---- spectral parameter values -> RGB (Spectrum = RGBSpectrum): the mirror of frontend/fe_spectrum.h, written with numpy float32 ----
_CIE = None


def _cie_tables():
    """CIE_X / CIE_Y / CIE_Z / CIE_LAMBDA (471 samples) and CIE_Y_INTEGRAL from include/pt_cie_tables.h (DATA of core/cie.rs)."""
    global _CIE
    if _CIE is None:
        import os, re
        txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "pt_cie_tables.h")).read()
        tabs = {}
        for name in ("CIE_X", "CIE_Y", "CIE_Z", "CIE_LAMBDA"):
            body = txt[txt.index("#define PT_%s_VALUES" % name):].split("\n#define")[0]
            vals = re.findall(r"([-+0-9.eE]+)f", body.split("\\", 1)[1])
            tabs[name] = np.array([float(v) for v in vals], dtype=F)
            assert len(tabs[name]) == 471
        tabs["Y_INT"] = F(float(re.search(r"PT_CIE_Y_INTEGRAL ([0-9.]+)f", txt).group(1)))
        _CIE = tabs
    return _CIE


def xyz_to_rgb(xyz):   # spectrum.rs:485-493
    x, y, z = (F(v) for v in xyz)
    return np.array([F(3.240479) * x - F(1.537150) * y - F(0.498535) * z, F(-0.969256) * x + F(1.875991) * y + F(0.041556) * z,
                     F(0.055648) * x - F(0.204043) * y + F(1.057311) * z], dtype=F)


def rgb_from_sampled(lam, vals):
    """RGBSpectrum::from_sampled (spectrum.rs:129-154) incl. its unsorted-input quirk (sorted wavelengths, unsorted values)."""
    T = _cie_tables()
    lam = np.asarray(lam, dtype=F); vals = np.asarray(vals, dtype=F)
    if len(lam) == 0: return np.zeros(3, dtype=F)
    if np.any(lam[:-1] > lam[1:]):
        order = sorted(range(len(lam)), key=lambda i: (float(lam[i]), float(vals[i])))
        lam = lam[order]
    xyz = np.zeros(3, dtype=F)
    n = len(lam)
    for i in range(471):
        l = T["CIE_LAMBDA"][i]
        if l <= lam[0]: v = vals[0]
        elif l >= lam[n - 1]: v = vals[n - 1]
        else:
            off = min(max(int(np.searchsorted(lam, l, side="right")) - 1, 0), n - 2)
            t = F(F(l - lam[off]) / F(lam[off + 1] - lam[off]))
            v = F(F(F(1) - t) * vals[off] + t * vals[off + 1])
        xyz[0] = F(xyz[0] + F(v * T["CIE_X"][i])); xyz[1] = F(xyz[1] + F(v * T["CIE_Y"][i])); xyz[2] = F(xyz[2] + F(v * T["CIE_Z"][i]))
    scale = F(F(T["CIE_LAMBDA"][470] - T["CIE_LAMBDA"][0]) / F(T["Y_INT"] * F(471)))
    return xyz_to_rgb(xyz * scale)


def rgb_from_blackbody(temperature, scale):
    """paramset.rs:163-180 + black_body_normalized (spectrum.rs:36-70), float32 arithmetic as written."""
    T = _cie_tables()
    t = F(temperature)
    def bb(lam):
        lam = np.asarray(lam, dtype=F)
        if t <= 0: return np.zeros_like(lam)
        c, h, kb = F(299792458.0), F(6.62606957e-34), F(1.3806488e-23)
        l = (lam.astype(np.float64) * np.float64(F(1.0e-9))).astype(F)
        lambda5 = ((l * l) * (l * l) * l).astype(F)
        e = np.exp(((h * c) / (l * kb * t)).astype(F)).astype(F)
        return ((F(2.0) * h * c * c) / (lambda5 * (e - F(1.0)))).astype(F)
    le = bb(T["CIE_LAMBDA"])
    lmax = F(F(F(2.8977721e-3) / t) * F(1.0e9))
    le = (le / bb([lmax])[0]).astype(F)
    return (rgb_from_sampled(T["CIE_LAMBDA"], le) * F(scale)).astype(F)


def rgb_from_spd_text(text):
    """`"spectrum x" "file.spd"`: read_float_file pushes every number twice (floatfile.rs:21-29), see fe_spectrum.h."""
    vals = []
    for line in text.splitlines():
        if not line or line.startswith("#"): continue
        for tok in line.split():
            try: v = float(np.float32(tok))
            except ValueError: return np.zeros(3, dtype=F)
            vals += [v, v]
    half = len(vals) // 2
    return rgb_from_sampled([vals[2 * j] for j in range(half)], [vals[2 * j + 1] for j in range(half)])
