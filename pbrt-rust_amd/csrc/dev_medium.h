// dev_medium.h -- participating media on device for the volumetric path integrator.
//   media/homogeneous.rs:31-68 (tr, sample); core/medium.rs:149-194 (Henyey-Greenstein p / sample_p);
//   core/primitive.rs:139-145 + core/interaction.rs:54-66 (which medium a spawned ray travels in).
// f32::exp goes through the shared f64 exp (dev_math.h dm_expd), like the oracle.
#pragma once
#include "dev_bsdf.h"

namespace ptd {

constexpr float kInv4Pi = 0.07957747154594766788f;
PT_DEV float dm_expf(float x) { return (float)dm_expd((double)x); }

// HomogeneousMedium::tr over a ray with parameter range t_max and direction d
PT_DEV RGB medium_tr(const PtMedium &m, float t_max, V3 d) {
    const float l = minf(t_max * length(d), 3.40282347e+38f);
    return RGB(dm_expf(-(m.sigma_a[0] + m.sigma_s[0]) * l), dm_expf(-(m.sigma_a[1] + m.sigma_s[1]) * l), dm_expf(-(m.sigma_a[2] + m.sigma_s[2]) * l));
}
// HomogeneousMedium::sample: u_channel, u_dist = the two get_1d() values in call order; returns the beta factor, sets `t`
// (ray parameter of the medium vertex) when `sampled`.
PT_DEV RGB medium_sample(const PtMedium &m, float t_max, V3 d, float u_channel, float u_dist, bool &sampled, float &t) {
    const float st[3] = {m.sigma_a[0] + m.sigma_s[0], m.sigma_a[1] + m.sigma_s[1], m.sigma_a[2] + m.sigma_s[2]};
    const float uc = u_channel * 3.0f;
    const uint32_t channel = min(uc > 0.0f ? (uint32_t)uc : 0u, 2u);
    const float dist = -dm_logf(1.0f - u_dist) / (channel == 0 ? st[0] : channel == 1 ? st[1] : st[2]);
    const float dl = length(d);
    t = minf(dist / dl, t_max);
    sampled = t < t_max;
    float tr[3], pdf = 0.0f;
    for (int i = 0; i < 3; ++i) { tr[i] = dm_expf(-st[i] * minf(t, 3.40282347e+38f) * dl); pdf += sampled ? st[i] * tr[i] : tr[i]; }
    pdf *= 1.0f / 3.0f;
    if (pdf == 0.0f) pdf = 1.0f;
    const RGB Tr(tr[0], tr[1], tr[2]);
    return sampled ? Tr * RGB(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]) / pdf : Tr / pdf;
}
// ---- GridDensityMedium (media/grid.rs) --------------------------------------------------------------------------------------
PT_DEV float grid_d(const PtMedium &m, const float *density, long long x, long long y, long long z) {   // grid.rs:102-110
    if (x < 0 || y < 0 || z < 0 || x >= (long long)m.nx || y >= (long long)m.ny || z >= (long long)m.nz) return 0.0f;
    return density[((size_t)z * m.ny + (size_t)y) * m.nx + (size_t)x];
}
PT_DEV float grid_lerp(float t, float a, float b) { return a * (1.0f - t) + b * t; }   // pbrt.rs:136-144
PT_DEV float grid_density(const PtMedium &m, const float *density, V3 p) {   // grid.rs:77-100; Point3i::from is `x as isize` (truncates towards zero)
    const float sx = p.x * (float)m.nx - 0.5f, sy = p.y * (float)m.ny - 0.5f, sz = p.z * (float)m.nz - 0.5f;
    const long long ix = f2i_sat(sx), iy = f2i_sat(sy), iz = f2i_sat(sz);
    const float dx = sx - (float)ix, dy = sy - (float)iy, dz = sz - (float)iz;
    const float d00 = grid_lerp(dx, grid_d(m, density, ix, iy, iz), grid_d(m, density, ix + 1, iy, iz));
    const float d10 = grid_lerp(dx, grid_d(m, density, ix, iy + 1, iz), grid_d(m, density, ix + 1, iy + 1, iz));
    const float d01 = grid_lerp(dx, grid_d(m, density, ix, iy, iz + 1), grid_d(m, density, ix + 1, iy, iz + 1));
    const float d11 = grid_lerp(dx, grid_d(m, density, ix, iy + 1, iz + 1), grid_d(m, density, ix + 1, iy + 1, iz + 1));
    return grid_lerp(dz, grid_lerp(dy, d00, d10), grid_lerp(dy, d01, d11));
}
// The ray of grid.rs:115-117 / :155-158 in medium space (o + normalize(d) * t, t_max * |d|, through world_to_medium with the error-bound
// shift of Transform::transform_ray) and its overlap [tmin, tmax] with the unit cube (Bounds3f::intersect_p, bounds.rs:533-557).
PT_DEV bool grid_ray(const PtMedium &m, V3 o, V3 d, float t_max, V3 &ro, V3 &rd, float &tmin, float &tmax) {
    const M4 w2m = ldm4g(m.world_to_medium);
    const V3 dn = normalize(d);
    float tm = t_max * length(d);
    V3 oerr; ro = xf_point_err(w2m, o, oerr); rd = xf_vector(w2m, dn);
    const float l2 = length_squared(rd);
    if (l2 > 0.0f) { const float dt = dot(vabs(rd), oerr) / l2; ro = ro + rd * dt; tm -= dt; }
    float t0 = 0.0f, t1 = tm;
    const float oc[3] = {ro.x, ro.y, ro.z}, dc[3] = {rd.x, rd.y, rd.z};
    for (int i = 0; i < 3; ++i) {
        const float inv = 1.0f / dc[i];
        float tnear = (0.0f - oc[i]) * inv, tfar = (1.0f - oc[i]) * inv;
        if (tnear > tfar) { const float tmp = tnear; tnear = tfar; tfar = tmp; }
        tfar *= 1.0f + 2.0f * gammaf(3);
        t0 = tnear > t0 ? tnear : t0;
        t1 = tfar < t1 ? tfar : t1;
        if (t0 > t1) return false;
    }
    tmin = t0; tmax = t1;
    return true;
}
// GridDensityMedium::tr (grid.rs:113-147): ratio tracking with roulette on low transmittance; draws its steps from the path's sampler
template <class S> PT_DEV float grid_tr(const PtMedium &m, const DevGridAux &g, V3 o, V3 d, float t_max, S &smp) {
    V3 ro, rd; float tmin, tmax;
    if (!grid_ray(m, o, d, t_max, ro, rd, tmin, tmax)) return 1.0f;
    float tr = 1.0f, t = tmin;
    for (;;) {
        t -= dm_logf(1.0f - smp.get_1d()) * g.inv_max_density / g.sigma_t;
        if (t >= tmax) break;
        if (smp.overflow) return 0.0f;   // (dimension overflow is reported as an error by the caller; leave the loop)
        const float density = grid_density(m, g.density, ro + rd * t);
        tr *= 1.0f - maxf(density * g.inv_max_density, 0.0f);
        if (tr < 0.1f) {
            const float q = maxf(1.0f - tr, 0.05f);
            if (smp.get_1d() < q) return 0.0f;
            tr /= 1.0f - q;
        }
    }
    return tr;
}
// GridDensityMedium::sample (grid.rs:149-182): delta tracking. Returns the beta factor; `sampled` + the vertex's parameter `t` (applied to
// the WORLD ray as the reference does: `ray.find_point(t)`, grid.rs:173).
template <class S> PT_DEV RGB grid_sample(const PtMedium &m, const DevGridAux &g, V3 o, V3 d, float t_max, S &smp, bool &sampled, float &t_out) {
    sampled = false;
    V3 ro, rd; float tmin, tmax;
    if (!grid_ray(m, o, d, t_max, ro, rd, tmin, tmax)) return RGB(1.0f);
    float t = tmin;
    for (;;) {
        t -= dm_logf(1.0f - smp.get_1d()) * g.inv_max_density / g.sigma_t;
        if (t >= tmax) break;
        if (smp.overflow) break;
        if (grid_density(m, g.density, ro + rd * t) * g.inv_max_density > smp.get_1d()) {
            sampled = true; t_out = t;
            return RGB(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]) / g.sigma_t;
        }
    }
    return RGB(1.0f);
}

PT_DEV float phase_hg(float cos_theta, float g) {
    const float denom = 1.0f + g * g + 2.0f * g * cos_theta;
    return kInv4Pi * (1.0f - g * g) / (denom * sqrtf(denom));
}
PT_DEV float hg_sample_p(float g, V3 wo, V3 &wi, P2 u) {
    float cos_theta;
    if (fabsf(g) < 1.0e-3f) cos_theta = 1.0f - 2.0f * u.x;
    else {
        const float sqr_term = (1.0f - g * g) / (1.0f + g - 2.0f * g * u.x);
        cos_theta = -(1.0f + g * g - sqr_term * sqr_term) / (2.0f * g);
    }
    const float sin_theta = sqrtf(maxf(1.0f - cos_theta * cos_theta, 0.0f));
    const float phi = 2.0f * kPi * u.y;
    float sp, cp; dm_sincosf(phi, sp, cp);
    V3 v1, v2; coordinate_system(wo, v1, v2);
    wi = v1 * sin_theta * cp + v2 * sin_theta * sp + wo * cos_theta;   // spherical_direction_basis (geometry.rs:36-38)
    return phase_hg(cos_theta, g);
}

// The BSDF interface of estimate_direct for a MediumInteraction (integrator.rs:142-147,186-190): f = pdf = p(wo, wi), no cosine.
struct PhaseBsdf {
    float g; V3 wo_;
    PT_DEV RGB f(V3 wo, V3 wi, int) const { return RGB(phase_hg(dot(wo, wi), g)); }
    PT_DEV float pdf(V3 wo, V3 wi, int) const { return phase_hg(dot(wo, wi), g); }
    PT_DEV RGB f_pdf(V3 wo, V3 wi, int fl, float &p) const { p = pdf(wo, wi, fl); return f(wo, wi, fl); }
    PT_DEV RGB sample_f(V3 wo, V3 &wi, P2 u, float &pdf, int, int &sampled) const { const float p = hg_sample_p(g, wo, wi, u); pdf = p; sampled = 0; return RGB(p); }
};

// MediumInterface of an interaction: the primitive's own when it is a transition, else the arriving ray's medium on both sides
struct MedIface { uint32_t inside, outside; };
PT_DEV MedIface surface_iface(const DeviceScene &s, uint32_t prim, uint32_t ray_medium) {
    MedIface m{ray_medium, ray_medium};
    if (s.prim_med_in) { const uint32_t a = s.prim_med_in[prim], b = s.prim_med_out[prim]; if (a != b) { m.inside = a; m.outside = b; } }
    return m;
}
PT_DEV uint32_t medium_toward(const MedIface &m, V3 n, V3 w) { return dot(w, n) > 0.0f ? m.outside : m.inside; }

}  // namespace ptd
