"""Analytic checks of the oracle's BSDFs (VERDICT r3 "weak" item 1: oracle and device share one reading of core/reflection.rs; the reference's tests
hold no vectors for it, so the risk of a misreading made twice is bounded here by statements that do not come from the Rust source):

  * the microfacet reflection lobes against the PUBLISHED Torrance-Sparrow model written out in numpy (Trowbridge-Reitz D, Smith G through Lambda,
    the exact conductor / dielectric Fresnel equations, roughness_to_alpha's polynomial as printed in the pbrt book) -- metal and plastic;
  * Helmholtz reciprocity f(wo, wi) = f(wi, wo) of every reflection lobe set;
  * sample_f against f and pdf: the estimator mean(f |cos| / pdf) over sample_f's own draws equals the quadrature of f |cos| over the sphere
    (ties BSDF::sample_f's lobe choice, pdf averaging and "matching components" logic, reflection.rs:1602-1689, to BSDF::f / BSDF::pdf), the pdf it
    reports is the pdf() of the direction it returns, and pdf integrates to at most one;

for the non-specular material families the device shades (matte / Oren-Nayar, plastic, metal, substrate, uber, disney; rough glass and translucent
through the transmission lobe's own test: the reference's pdf there is not the density of its samples, a quirk the oracle keeps).
The device's BSDFs are tied to these by the GPU == oracle films and counters of the parity suite (tests/test_gpu_parity.py, test_fuzz_parity.py).
The reference's known quirks are respected: disney's clearcoat lobe reports a pdf for the half vector `wi + wi` (disney.rs), so it is left out of the
estimator check."""
import ctypes as C

import numpy as np
import pytest


def _material_scene(pkg, kind, kw):
    b = pkg.host.SceneBuilder()
    b.film.update(xres=8, yres=8); b.spp = 1
    b.look_at((0.0, 2.0, 5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)); b.camera(fov=40.0)
    b.world_begin()
    b.light_source("infinite", L=(1.0, 1.0, 1.0))
    b.material(kind, **kw)
    b.trianglemesh(np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32), np.array([0, 1, 2, 0, 2, 3], np.uint32))
    sd, rp = b.world_end()
    d = sd.desc()
    return sd, int(d.prim_material[0])


class Bsdf:
    def __init__(self, pkg, oracle, kind, **kw):
        self.A = pkg._abi
        self.sd, self.mi = _material_scene(pkg, kind, kw)
        self.s = oracle.scene(self.sd)
        self.fn = oracle.lib.orc_bsdf_eval
        fp, ip = self.A.fp, C.POINTER(C.c_int32)
        self.fn.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, fp, fp, fp, fp, fp, fp, fp, fp, ip, ip]
        self.fn.restype = C.c_int

    def _p(self, a):
        return a.ctypes.data_as(self.A.fp)

    def f_pdf(self, wo, wi):
        wo = np.ascontiguousarray(wo, np.float32); wi = np.ascontiguousarray(wi, np.float32); n = len(wo)
        f = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32); nl = C.c_int32()
        assert self.fn(self.s.h, self.mi, n, self._p(wo), self._p(wi), None, self._p(f), self._p(pdf), None, None, None, None, C.byref(nl)) == 0
        return f.astype(np.float64), pdf.astype(np.float64)

    def sample(self, wo, u):
        wo = np.ascontiguousarray(wo, np.float32); u = np.ascontiguousarray(u, np.float32); n = len(wo)
        wi = np.zeros((n, 3), np.float32); f = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32); ty = np.zeros(n, np.int32)
        assert self.fn(self.s.h, self.mi, n, self._p(wo), None, self._p(u), None, None, self._p(wi), self._p(f), self._p(pdf), ty.ctypes.data_as(C.POINTER(C.c_int32)), None) == 0
        return wi.astype(np.float64), f.astype(np.float64), pdf.astype(np.float64), ty


def _dir(theta, phi):
    return np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], axis=-1)


def _sphere_grid(nt=768, nphi=1536):
    """Midpoint rule in (cos theta, phi): directions and solid-angle weights over the whole sphere."""
    ct = -1.0 + (np.arange(nt) + 0.5) * (2.0 / nt)
    ph = (np.arange(nphi) + 0.5) * (2.0 * np.pi / nphi)
    CT, PH = np.meshgrid(ct, ph, indexing="ij")
    st = np.sqrt(np.maximum(0.0, 1.0 - CT * CT))
    w = np.stack([st * np.cos(PH), st * np.sin(PH), CT], axis=-1).reshape(-1, 3)
    return w, (2.0 / nt) * (2.0 * np.pi / nphi)


# ---- the published model, written from the equations (pbrt book ch. 8, Walter et al. 2007, Trowbridge & Reitz 1975) ----
def roughness_to_alpha(r):
    x = np.log(max(r, 1e-3))
    return 1.62142 + 0.819955 * x + 0.1734 * x * x + 0.0171201 * x ** 3 + 0.000640711 * x ** 4


def tr_d(wh, ax, ay):
    c2 = wh[..., 2] ** 2
    s2 = np.maximum(0.0, 1.0 - c2)
    t2 = s2 / c2
    cp2 = np.where(s2 > 0, wh[..., 0] ** 2 / np.maximum(s2, 1e-300), 1.0); sp2 = np.where(s2 > 0, wh[..., 1] ** 2 / np.maximum(s2, 1e-300), 0.0)
    e = t2 * (cp2 / ax ** 2 + sp2 / ay ** 2)
    return 1.0 / (np.pi * ax * ay * c2 * c2 * (1.0 + e) ** 2)


def tr_lambda(w, ax, ay):
    c2 = w[..., 2] ** 2
    s2 = np.maximum(0.0, 1.0 - c2)
    t2 = s2 / c2
    cp2 = np.where(s2 > 0, w[..., 0] ** 2 / np.maximum(s2, 1e-300), 1.0); sp2 = np.where(s2 > 0, w[..., 1] ** 2 / np.maximum(s2, 1e-300), 0.0)
    a2 = cp2 * ax ** 2 + sp2 * ay ** 2
    return (-1.0 + np.sqrt(1.0 + a2 * t2)) / 2.0


def fresnel_conductor(cos_i, eta, k):   # exact equations for an absorbing medium, eta and k relative to the incident side
    c2 = cos_i ** 2; s2 = 1.0 - c2
    e2, k2 = eta ** 2, k ** 2
    t0 = e2 - k2 - s2
    a2b2 = np.sqrt(t0 * t0 + 4.0 * e2 * k2)
    t1 = a2b2 + c2
    a = np.sqrt(0.5 * (a2b2 + t0))
    t2 = 2.0 * a * cos_i
    rs = (t1 - t2) / (t1 + t2)
    t3 = c2 * a2b2 + s2 * s2
    t4 = t2 * s2
    rp = rs * (t3 - t4) / (t3 + t4)
    return 0.5 * (rp + rs)


def fresnel_dielectric(cos_i, eta_i, eta_t):
    cos_i = np.clip(cos_i, -1.0, 1.0)
    ei = np.where(cos_i > 0, eta_i, eta_t); et = np.where(cos_i > 0, eta_t, eta_i); ci = np.abs(cos_i)
    st = ei / et * np.sqrt(np.maximum(0.0, 1.0 - ci * ci))
    ct = np.sqrt(np.maximum(0.0, 1.0 - st * st))
    rl = (et * ci - ei * ct) / (et * ci + ei * ct); rp = (ei * ci - et * ct) / (ei * ci + et * ct)
    return np.where(st >= 1.0, 1.0, 0.5 * (rl * rl + rp * rp))


def torrance_sparrow(wo, wi, ax, ay, fresnel):
    wh = wo + wi
    wh = wh / np.linalg.norm(wh, axis=-1, keepdims=True)
    g = 1.0 / (1.0 + tr_lambda(wo, ax, ay) + tr_lambda(wi, ax, ay))
    cos_h = np.abs(np.sum(wi * wh, axis=-1))
    return tr_d(wh, ax, ay) * g * fresnel(cos_h) / (4.0 * np.abs(wo[..., 2]) * np.abs(wi[..., 2]))


def _upper_pairs(n, seed):
    rng = np.random.default_rng(seed)
    wo = _dir(np.arccos(rng.uniform(0.15, 0.98, n)), rng.uniform(0, 2 * np.pi, n))
    wi = _dir(np.arccos(rng.uniform(0.15, 0.98, n)), rng.uniform(0, 2 * np.pi, n))
    return wo, wi


def test_metal_is_the_published_torrance_sparrow_conductor(pkg, oracle):
    eta, k = np.array([0.2, 0.92, 1.1]), np.array([3.9, 2.45, 2.14])
    for ur, vr in ((0.3, 0.3), (0.15, 0.4)):
        b = Bsdf(pkg, oracle, "metal", eta_rgb=tuple(eta), k=tuple(k), uroughness=ur, vroughness=vr)
        wo, wi = _upper_pairs(4000, 1)
        f, _ = b.f_pdf(wo, wi)
        ax, ay = roughness_to_alpha(ur), roughness_to_alpha(vr)
        for c in range(3):
            ref = torrance_sparrow(wo, wi, ax, ay, lambda ch: fresnel_conductor(ch, eta[c], k[c]))
            np.testing.assert_allclose(f[:, c], ref, rtol=3e-4, atol=1e-7)


def test_plastic_is_lambert_plus_the_published_dielectric_microfacet_lobe(pkg, oracle):
    kd, ks, r = np.array([0.1, 0.3, 0.6]), np.array([0.4, 0.35, 0.3]), 0.25
    b = Bsdf(pkg, oracle, "plastic", Kd=tuple(kd), Ks=tuple(ks), roughness=r)
    wo, wi = _upper_pairs(4000, 2)
    f, _ = b.f_pdf(wo, wi)
    a = roughness_to_alpha(r)
    spec = torrance_sparrow(wo, wi, a, a, lambda ch: fresnel_dielectric(ch, 1.5, 1.0))   # plastic.rs: FresnelDielectric::new(1.5, 1.0), as written there
    for c in range(3):
        np.testing.assert_allclose(f[:, c], kd[c] / np.pi + ks[c] * spec, rtol=3e-4, atol=1e-7)


REFLECTIVE = [("matte", dict(Kd=(0.6, 0.5, 0.4), sigma=35.0)), ("plastic", dict(Kd=(0.2, 0.3, 0.4), Ks=(0.5, 0.5, 0.5), roughness=0.3)),
              ("metal", dict(eta_rgb=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), uroughness=0.2, vroughness=0.35)),
              ("uber", dict(Kd=(0.3, 0.5, 0.2), Ks=(0.3, 0.3, 0.3), roughness=0.3))]


@pytest.mark.parametrize("kind,kw", REFLECTIVE, ids=[k for k, _ in REFLECTIVE])
def test_reflection_lobes_are_reciprocal(pkg, oracle, kind, kw):
    b = Bsdf(pkg, oracle, kind, **kw)
    wo, wi = _upper_pairs(3000, 3)
    f1, _ = b.f_pdf(wo, wi); f2, _ = b.f_pdf(wi, wo)
    np.testing.assert_allclose(f1, f2, rtol=2e-4, atol=1e-7)


def test_rough_glass_transmission_is_walters_btdf_with_the_reference_pdf(pkg, oracle):
    """f of the transmission lobe against Walter et al.'s BTDF (eq. 21, radiance transport: the eta^2 of the measure cancels the 1 / eta^2 of radiance
    scaling) -- and its pdf against the formula AS THE REFERENCE WRITES IT: MicrofacetTransmission::pdf takes eta = etaa / etab for wo above the surface
    (reflection.rs:1118), the reciprocal of what f (:1071-1075) and pbrt-v3 use, so its half vector is not the one the sample was made with. The oracle
    follows the reference, not the book: with the book's eta the two pdfs differ by 2-3x over most of the hemisphere (asserted below, so that a
    "corrected" oracle fails here). Consequence, in the reference too: mean(f |cos| / pdf) over sample_f's draws is not the integral of f for rough glass
    and translucent materials -- those two stay out of the estimator test."""
    r, eta_g = 0.35, 1.5
    b = Bsdf(pkg, oracle, "glass", eta=eta_g, uroughness=r, vroughness=r)
    a = roughness_to_alpha(r)
    rng = np.random.default_rng(5)
    n = 4000
    wo = _dir(np.arccos(rng.uniform(0.2, 0.98, n)), rng.uniform(0, 2 * np.pi, n))
    wi = _dir(np.pi - np.arccos(rng.uniform(0.2, 0.98, n)), rng.uniform(0, 2 * np.pi, n))      # below the surface
    f, pdf = b.f_pdf(wo, wi)

    def dots(eta):
        wh = wo + wi * eta
        wh = wh / np.linalg.norm(wh, axis=-1, keepdims=True)
        return wh, np.sum(wo * wh, axis=-1), np.sum(wi * wh, axis=-1)
    # f: eta = etab / etaa = 1.5, wh flipped into the upper hemisphere
    wh, owh, iwh = dots(eta_g)
    flip = wh[:, 2] < 0
    wh[flip] *= -1; owh[flip] *= -1; iwh[flip] *= -1
    valid = owh * iwh <= 0
    g = 1.0 / (1.0 + tr_lambda(wo, a, a) + tr_lambda(wi, a, a))
    fr = fresnel_dielectric(owh, 1.0, eta_g)
    ref_f = np.where(valid, (1.0 - fr) * np.abs(tr_d(wh, a, a) * g * np.abs(iwh) * np.abs(owh) / (wi[:, 2] * wo[:, 2] * (owh + eta_g * iwh) ** 2)), 0.0)
    np.testing.assert_allclose(f[:, 0], ref_f, rtol=5e-4, atol=1e-7)

    def lobe_pdf(eta):   # D(wh) G1(wo) |wo.wh| / |cos theta_o| x |eta^2 wi.wh| / (wo.wh + eta wi.wh)^2 (Walter eq. 17 with visible-normal sampling)
        wh, owh, iwh = dots(eta)
        ok = owh * iwh <= 0
        g1 = 1.0 / (1.0 + tr_lambda(wo, a, a))
        return np.where(ok, tr_d(wh, a, a) * g1 * np.abs(owh) / np.abs(wo[:, 2]) * np.abs(eta * eta * iwh) / (owh + eta * iwh) ** 2, 0.0)
    as_reference = 0.5 * lobe_pdf(1.0 / eta_g)   # BSDF::pdf averages the two lobes; the reflection lobe has no density below the surface
    np.testing.assert_allclose(pdf, as_reference, rtol=5e-4, atol=1e-7)
    as_book = 0.5 * lobe_pdf(eta_g)
    both = (as_book > 1e-3) & (as_reference > 1e-3)
    assert both.sum() > 1000 and np.median(as_book[both] / as_reference[both]) > 1.5


FAMILIES = REFLECTIVE + [
    ("substrate", dict(Kd=(0.5, 0.3, 0.1), Ks=(0.2, 0.2, 0.2), uroughness=0.3, vroughness=0.4)),
    ("disney", dict(color=(0.6, 0.4, 0.3), metallic=0.3, roughness=0.45, sheen=0.5, speculartint=0.3, flatness=0.0)),
    ("disney", dict(color=(0.8, 0.8, 0.2), metallic=1.0, roughness=0.35, anisotropic=0.5)),
]


@pytest.mark.parametrize("case", range(len(FAMILIES)), ids=[f"{k}{i}" for i, (k, _) in enumerate(FAMILIES)])
def test_sample_f_is_an_unbiased_estimator_of_the_quadrature_of_f(pkg, oracle, case):
    """mean over sample_f's draws of f |cos| / pdf == integral of f |cos| over the sphere (midpoint rule, 1.2 M directions), to 1.5 %; the pdf that
    sample_f reports is pdf() of the direction it returns and f likewise; and pdf integrates to at most one."""
    kind, kw = FAMILIES[case]
    b = Bsdf(pkg, oracle, kind, **kw)
    grid, dw = _sphere_grid()
    rng = np.random.default_rng(10 + case)
    for cos_o in (0.9, 0.45):
        wo1 = _dir(np.arccos(cos_o), 0.7)
        f, pdf = b.f_pdf(np.broadcast_to(wo1, grid.shape), grid)
        quad = (f * np.abs(grid[:, 2:3])).sum(axis=0) * dw
        assert pdf.sum() * dw < 1.0 + 2e-3, (kind, pdf.sum() * dw)
        assert pdf.sum() * dw > 0.5, (kind, pdf.sum() * dw)           # (microfacet sampling loses the draws that leave the hemisphere: 0.80 for plastic at 63 degrees)
        n = 400000
        wi, fs, ps, ty = b.sample(np.broadcast_to(wo1, (n, 3)), rng.random((n, 2)))
        ok = ps > 0
        est = np.where(ok[:, None], fs * np.abs(wi[:, 2:3]) / np.maximum(ps[:, None], 1e-300), 0.0)
        mc = est.mean(axis=0); se = est.std(axis=0) / np.sqrt(n)
        assert np.all(np.abs(mc - quad) < 0.015 * np.maximum(quad, 1e-3) + 4.0 * se), (kind, cos_o, mc, quad, se)
        # the sample's own report agrees with f() and pdf() of the direction it chose
        sel = np.flatnonzero(ok)[:5000]
        f2, p2 = b.f_pdf(np.broadcast_to(wo1, (len(sel), 3)), wi[sel])
        np.testing.assert_allclose(p2, ps[sel], rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(f2, fs[sel], rtol=2e-4, atol=1e-7)
