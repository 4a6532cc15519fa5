"""Host-side BSSRDFTable construction (photon beam diffusion), passed to the device through PtBSSRDFTable.

Mirrors the reference's material-creation-time work, which is outside the render hot path:
  core/bssrdf.rs:22-56    fresnel_moment1 / fresnel_moment2
  core/bssrdf.rs:58-112   beam_diffusion_ms
  core/bssrdf.rs:114-136  beam_diffusion_ss
  core/bssrdf.rs:138-188  compute_beam_diffusion_bssrdf (100 albedo x 64 radius samples)
  core/bssrdf.rs:190-202  subsurface_from_diffuse
  core/interpolation.rs:233-263 integrate_catmull_rom, :265-330 invert_catmull_rom
  core/medium.rs:150-154  phase_hg ; core/medium.rs named-medium table (a few entries used by tests)

All arithmetic is float32 (numpy), as in the reference; the table is *input data* for both the oracle and the HIP path
(both receive the same arrays), so transcendental rounding differences against Rust's libm do not affect parity.
"""
import functools

import numpy as np

f32 = np.float32
PI = f32(np.pi)
INV4_PI = f32(0.07957747154594766788)


def fresnel_moment1(eta):
    eta = f32(eta)
    e2 = eta * eta; e3 = e2 * eta; e4 = e3 * eta; e5 = e4 * eta
    if eta < 1.0:
        return f32(0.45966) - f32(1.73965) * eta + f32(3.37668) * e2 - f32(3.904945) * e3 + f32(2.49277) * e4 - f32(0.68441) * e5
    return f32(-4.61686) + f32(11.1136) * eta - f32(10.4646) * e2 + f32(5.11455) * e3 - f32(1.27198) * e4 + f32(0.12746) * e5


def fresnel_moment2(eta):
    eta = f32(eta)
    e2 = eta * eta; e3 = e2 * eta; e4 = e3 * eta; e5 = e4 * eta
    if eta < 1.0:
        return f32(0.27614) - f32(0.87350) * eta + f32(1.12077) * e2 - f32(0.65095) * e3 + f32(0.07883) * e4 + f32(0.04860) * e5
    r = f32(1.0) / eta; r2 = r * r; r3 = r2 * r
    return (f32(-547.033) + f32(45.3087) * r3 - f32(218.725) * r2 + f32(458.843) * r + f32(404.557) * eta - f32(189.519) * e2
            + f32(54.9327) * e3 - f32(9.00603) * e4 + f32(0.63942) * e5)


def _fr_dielectric(cos_i, eta_i, eta_t):
    """core/reflection.rs fr_dielectric, vectorised over cos_i (float32 arrays)."""
    cos_i = np.clip(cos_i, f32(-1.0), f32(1.0)).astype(f32)
    entering = cos_i > 0
    ei = np.where(entering, f32(eta_i), f32(eta_t)).astype(f32)
    et = np.where(entering, f32(eta_t), f32(eta_i)).astype(f32)
    cos_i = np.abs(cos_i)
    sin_i = np.sqrt(np.maximum(f32(0.0), f32(1.0) - cos_i * cos_i))
    sin_t = ei / et * sin_i
    tir = sin_t >= 1.0
    cos_t = np.sqrt(np.maximum(f32(0.0), f32(1.0) - sin_t * sin_t))
    with np.errstate(invalid="ignore", divide="ignore"):
        rparl = ((et * cos_i) - (ei * cos_t)) / ((et * cos_i) + (ei * cos_t))
        rperp = ((ei * cos_i) - (et * cos_t)) / ((ei * cos_i) + (et * cos_t))
    out = (rparl * rparl + rperp * rperp) / f32(2.0)
    return np.where(tir, f32(1.0), out).astype(f32)


def _phase_hg(cos_theta, g):
    g = f32(g)
    denom = f32(1.0) + g * g + f32(2.0) * g * cos_theta
    return INV4_PI * (f32(1.0) - g * g) / (denom * np.sqrt(denom))


def _beam_diffusion_ms(sigma_s, sigma_a, g, eta, r, nsamples=100):
    # sigma_s, sigma_a, r: broadcastable float32 arrays with a trailing sample axis of length 1
    g = f32(g); eta = f32(eta)
    sigmap_s = sigma_s * (f32(1.0) - g)
    sigmap_t = sigma_a + sigmap_s
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        rhop = sigmap_s / sigmap_t
        dg = (f32(2.0) * sigma_a + sigmap_s) / (f32(3.0) * sigmap_t * sigmap_t)
        sigma_tr = np.sqrt(sigma_a / dg)
        fm1 = fresnel_moment1(eta); fm2 = fresnel_moment2(eta)
        ze = f32(-2.0) * dg * (f32(1.0) + f32(3.0) * fm2) / (f32(1.0) - f32(2.0) * fm1)
        cphi = f32(0.25) * (f32(1.0) - f32(2.0) * fm1)
        ce = f32(0.5) * (f32(1.0) - f32(3.0) * fm2)
        i = np.arange(nsamples, dtype=f32)
        zr = -np.log(f32(1.0) - (i + f32(0.5)) / f32(nsamples)).astype(f32) / sigmap_t
        zv = -zr + f32(2.0) * ze
        dr = np.sqrt(r * r + zr * zr)
        dv = np.sqrt(r * r + zv * zv)
        phid = INV4_PI / dg * (np.exp(-sigma_tr * dr) / dr - np.exp(-sigma_tr * dv) / dv)
        edn = INV4_PI * (zr * (f32(1.0) + sigma_tr * dr) * np.exp(-sigma_tr * dr) / (dr * dr * dr)
                         - zv * (f32(1.0) + sigma_tr * dv) * np.exp(-sigma_tr * dv) / (dv * dv * dv))
        E = phid * cphi + edn * ce
        kappa = f32(1.0) - np.exp(f32(-2.0) * sigmap_t * (dr + zr))
        terms = (kappa * rhop * rhop * E).astype(f32)
    ed = np.zeros(terms.shape[:-1], dtype=f32)
    for k in range(nsamples):      # sequential f32 accumulation, as the reference's loop
        ed = (ed + terms[..., k]).astype(f32)
    return ed / f32(nsamples)


def _beam_diffusion_ss(sigma_s, sigma_a, g, eta, r, nsamples=100):
    g = f32(g); eta = f32(eta)
    sigma_t = sigma_a + sigma_s
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        rho = sigma_s / sigma_t
        tcrit = r * np.sqrt(eta * eta - f32(1.0))
        i = np.arange(nsamples, dtype=f32)
        ti = tcrit - np.log(f32(1.0) - (i + f32(0.5)) / f32(nsamples)).astype(f32) / sigma_t
        d = np.sqrt(r * r + ti * ti)
        cos_o = ti / d
        terms = (rho * np.exp(-sigma_t * (d + tcrit)) / (d * d) * _phase_hg(cos_o, g)
                 * (f32(1.0) - _fr_dielectric(-cos_o, 1.0, eta)) * np.abs(cos_o)).astype(f32)
    ess = np.zeros(terms.shape[:-1], dtype=f32)
    for k in range(nsamples):
        ess = (ess + terms[..., k]).astype(f32)
    return ess / f32(nsamples)


def integrate_catmull_rom(x, values):
    """interpolation.rs:233-263; returns (integral, cdf)."""
    n = len(x)
    cdf = np.zeros(n, dtype=f32)
    s = f32(0.0)
    for i in range(n - 1):
        x0, x1 = x[i], x[i + 1]
        f0, f1 = values[i], values[i + 1]
        width = x1 - x0
        d0 = width * (f1 - values[i - 1]) / (x1 - x[i - 1]) if i > 0 else f1 - f0
        d1 = width * (values[i + 2] - f0) / (x[i + 2] - x0) if i + 2 < n else f1 - f0
        s = f32(s + ((d0 - d1) * f32(1.0 / 12.0) + (f0 + f1) * f32(0.5)) * width)
        cdf[i + 1] = s
    return s, cdf


def invert_catmull_rom(x, values, u):
    """interpolation.rs:265-345."""
    n = len(x)
    u = f32(u)
    if not (u > values[0]):
        return x[0]
    if not (u < values[n - 1]):
        return x[n - 1]
    # find_interval(n, values[i] <= u)
    i = int(np.clip(np.searchsorted(values, u, side="right") - 1, 0, n - 2))
    x0, x1 = x[i], x[i + 1]
    f0, f1 = values[i], values[i + 1]
    width = x1 - x0
    d0 = width * (f1 - values[i - 1]) / (x1 - x[i - 1]) if i > 0 else f1 - f0
    d1 = width * (values[i + 2] - f0) / (x[i + 2] - x0) if i + 2 < n else f1 - f0
    a, b, t = f32(0.0), f32(1.0), f32(0.5)
    for _ in range(200):
        if not (t > a and t < b):
            t = f32(0.5) * (a + b)
        t2 = t * t; t3 = t2 * t
        Fhat = ((f32(2.0) * t3 - f32(3.0) * t2 + f32(1.0)) * f0 + (f32(-2.0) * t3 + f32(3.0) * t2) * f1
                + (t3 - f32(2.0) * t2 + t) * d0 + (t3 - t2) * d1)
        fhat = ((f32(6.0) * t2 - f32(6.0) * t) * f0 + (f32(-6.0) * t2 + f32(6.0) * t) * f1
                + (f32(3.0) * t2 - f32(4.0) * t + f32(1.0)) * d0 + (f32(3.0) * t2 - f32(2.0) * t) * d1)
        if abs(Fhat - u) < 1.0e-6 or b - a < 1.0e-6:
            break
        if Fhat - u < 0.0:
            a = t
        else:
            b = t
        t = f32(t - (Fhat - u) / fhat)
    return f32(x0 + t * width)


class BSSRDFTable:
    """core/bssrdf.rs:241-268."""

    def __init__(self, n_rho=100, n_radius=64):
        self.n_rho, self.n_radius = n_rho, n_radius
        self.rho_samples = np.zeros(n_rho, dtype=f32)
        self.radius_samples = np.zeros(n_radius, dtype=f32)
        self.profile = np.zeros(n_rho * n_radius, dtype=f32)
        self.rhoeff = np.zeros(n_rho, dtype=f32)
        self.profile_cdf = np.zeros(n_rho * n_radius, dtype=f32)


@functools.lru_cache(maxsize=16)
def compute_beam_diffusion_bssrdf(g, eta, n_rho=100, n_radius=64):
    """core/bssrdf.rs:138-188.  Cached per (g, eta): the reference rebuilds it per material instance."""
    t = BSSRDFTable(n_rho, n_radius)
    t.radius_samples[0] = 0.0
    t.radius_samples[1] = 2.5e-3
    for i in range(2, n_radius):
        t.radius_samples[i] = t.radius_samples[i - 1] * f32(1.2)
    i = np.arange(n_rho, dtype=f32)
    t.rho_samples[:] = (f32(1.0) - np.exp(f32(-8.0) * i / f32(n_rho - 1)).astype(f32)) / (f32(1.0) - np.exp(f32(-8.0)))
    rho = t.rho_samples[:, None, None]
    r = t.radius_samples[None, :, None]
    prof = f32(2.0) * PI * r[..., 0] * (_beam_diffusion_ss(rho, f32(1.0) - rho, g, eta, r)
                                      + _beam_diffusion_ms(rho, f32(1.0) - rho, g, eta, r))
    prof = np.nan_to_num(prof.astype(f32), nan=0.0, posinf=0.0, neginf=0.0)
    t.profile[:] = prof.reshape(-1)
    for k in range(n_rho):
        s, cdf = integrate_catmull_rom(t.radius_samples, t.profile[k * n_radius:(k + 1) * n_radius])
        t.rhoeff[k] = s
        t.profile_cdf[k * n_radius:(k + 1) * n_radius] = cdf
    return t


def subsurface_from_diffuse(table, rho_eff, mfp):
    """core/bssrdf.rs:190-202."""
    sa, ss = np.zeros(3, dtype=f32), np.zeros(3, dtype=f32)
    for c in range(3):
        rho = invert_catmull_rom(table.rho_samples, table.rhoeff, rho_eff[c])
        ss[c] = rho / f32(mfp[c])
        sa[c] = (f32(1.0) - rho) / f32(mfp[c])
    return sa, ss


# A few rows of core/medium.rs's named-medium table (sigma_prime_s, sigma_a in mm^-1), used by the scene catalogue.
NAMED_MEDIA = {
    "Skin1": ((0.74, 0.88, 1.01), (0.032, 0.17, 0.48)),
    "Marble": ((2.19, 2.62, 3.00), (0.0021, 0.0041, 0.0071)),
    "Ketchup": ((0.18, 0.07, 0.03), (0.061, 0.97, 1.45)),
}
