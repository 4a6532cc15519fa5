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
