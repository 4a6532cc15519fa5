// fe_bssrdf.h -- BSSRDFTable construction of the .pbrt front end (material-creation-time work, outside the hot path):
//   core/bssrdf.rs:22-56 fresnel moments, :58-136 beam_diffusion_ms / _ss, :138-188 compute_beam_diffusion_bssrdf,
//   :190-202 subsurface_from_diffuse ; core/interpolation.rs:233-263 integrate_catmull_rom, :265-345 invert_catmull_rom ;
//   core/medium.rs:150-154 phase_hg and the named-media table (a few rows) ; core/reflection.rs:29-52 fr_dielectric.
// Python twin: pbrt-rust_amd/bssrdf.py.
#pragma once
#include <cmath>
#include <map>
#include <string>
#include <vector>

namespace fe {

struct BssTable { int n_rho = 100, n_radius = 64; std::vector<float> rho_samples, radius_samples, profile, rhoeff, profile_cdf; };

inline float fresnel_moment1(float eta) {
    float e2 = eta * eta, e3 = e2 * eta, e4 = e3 * eta, e5 = e4 * eta;
    if (eta < 1.0f) return 0.45966f - 1.73965f * eta + 3.37668f * e2 - 3.904945f * e3 + 2.49277f * e4 - 0.68441f * e5;
    return -4.61686f + 11.1136f * eta - 10.4646f * e2 + 5.11455f * e3 - 1.27198f * e4 + 0.12746f * e5;
}
inline float fresnel_moment2(float eta) {
    float e2 = eta * eta, e3 = e2 * eta, e4 = e3 * eta, e5 = e4 * eta;
    if (eta < 1.0f) return 0.27614f - 0.87350f * eta + 1.12077f * e2 - 0.65095f * e3 + 0.07883f * e4 + 0.04860f * e5;
    float r = 1.0f / eta, r2 = r * r, r3 = r2 * r;
    return -547.033f + 45.3087f * r3 - 218.725f * r2 + 458.843f * r + 404.557f * eta - 189.519f * e2 + 54.9327f * e3 - 9.00603f * e4 + 0.63942f * e5;
}
inline float fe_fr_dielectric(float cos_i, float eta_i, float eta_t) {
    cos_i = cos_i < -1.0f ? -1.0f : (cos_i > 1.0f ? 1.0f : cos_i);
    if (!(cos_i > 0.0f)) { std::swap(eta_i, eta_t); cos_i = std::fabs(cos_i); }
    float sin_i = std::sqrt(std::fmax(0.0f, 1.0f - cos_i * cos_i));
    float sin_t = eta_i / eta_t * sin_i;
    if (sin_t >= 1.0f) return 1.0f;
    float cos_t = std::sqrt(std::fmax(0.0f, 1.0f - sin_t * sin_t));
    float rparl = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
    float rperp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return (rparl * rparl + rperp * rperp) / 2.0f;
}
inline float phase_hg(float c, float g) { float d = 1.0f + g * g + 2.0f * g * c; return 0.07957747154594766788f * (1.0f - g * g) / (d * std::sqrt(d)); }
inline float beam_diffusion_ms(float sigma_s, float sigma_a, float g, float eta, float r) {
    const int ns = 100; float ed = 0.0f;
    float sigmap_s = sigma_s * (1.0f - g), sigmap_t = sigma_a + sigmap_s, rhop = sigmap_s / sigmap_t;
    float dg = (2.0f * sigma_a + sigmap_s) / (3.0f * sigmap_t * sigmap_t);
    float sigma_tr = std::sqrt(sigma_a / dg);
    float fm1 = fresnel_moment1(eta), fm2 = fresnel_moment2(eta);
    float ze = -2.0f * dg * (1.0f + 3.0f * fm2) / (1.0f - 2.0f * fm1);
    float cphi = 0.25f * (1.0f - 2.0f * fm1), ce = 0.5f * (1.0f - 3.0f * fm2);
    const float inv4pi = 0.07957747154594766788f;
    for (int i = 0; i < ns; ++i) {
        float zr = -std::log(1.0f - ((float)i + 0.5f) / (float)ns) / sigmap_t;
        float zv = -zr + 2.0f * ze;
        float dr = std::sqrt(r * r + zr * zr), dv = std::sqrt(r * r + zv * zv);
        float phid = inv4pi / dg * (std::exp(-sigma_tr * dr) / dr - std::exp(-sigma_tr * dv) / dv);
        float edn = inv4pi * (zr * (1.0f + sigma_tr * dr) * std::exp(-sigma_tr * dr) / (dr * dr * dr) - zv * (1.0f + sigma_tr * dv) * std::exp(-sigma_tr * dv) / (dv * dv * dv));
        float E = phid * cphi + edn * ce;
        float kappa = 1.0f - std::exp(-2.0f * sigmap_t * (dr + zr));
        ed += kappa * rhop * rhop * E;
    }
    return ed / (float)ns;
}
inline float beam_diffusion_ss(float sigma_s, float sigma_a, float g, float eta, float r) {
    float sigma_t = sigma_a + sigma_s, rho = sigma_s / sigma_t;
    float tcrit = r * std::sqrt(eta * eta - 1.0f), ess = 0.0f;
    const int ns = 100;
    for (int i = 0; i < ns; ++i) {
        float ti = tcrit - std::log(1.0f - ((float)i + 0.5f) / (float)ns) / sigma_t;
        float d = std::sqrt(r * r + ti * ti), cos_o = ti / d;
        ess += rho * std::exp(-sigma_t * (d + tcrit)) / (d * d) * phase_hg(cos_o, g) * (1.0f - fe_fr_dielectric(-cos_o, 1.0f, eta)) * std::fabs(cos_o);
    }
    return ess / (float)ns;
}
inline float integrate_catmull_rom(int n, const float *x, const float *v, float *cdf) {
    float sum = 0.0f; cdf[0] = 0.0f;
    for (int i = 0; i < n - 1; ++i) {
        float x0 = x[i], x1 = x[i + 1], f0 = v[i], f1 = v[i + 1], width = x1 - x0;
        float d0 = i > 0 ? width * (f1 - v[i - 1]) / (x1 - x[i - 1]) : f1 - f0;
        float d1 = i + 2 < n ? width * (v[i + 2] - f0) / (x[i + 2] - x0) : f1 - f0;
        sum += ((d0 - d1) * (1.0f / 12.0f) + (f0 + f1) * 0.5f) * width;
        cdf[i + 1] = sum;
    }
    return sum;
}
inline float invert_catmull_rom(int n, const float *x, const float *v, float u) {
    if (!(u > v[0])) return x[0];
    if (!(u < v[n - 1])) return x[n - 1];
    int first = 0, len = n;   // find_interval(values[i] <= u)
    while (len > 0) { int half = len >> 1, mid = first + half; if (v[mid] <= u) { first = mid + 1; len -= half + 1; } else len = half; }
    int i = std::min(std::max(first - 1, 0), n - 2);
    float x0 = x[i], x1 = x[i + 1], f0 = v[i], f1 = v[i + 1], width = x1 - x0;
    float d0 = i > 0 ? width * (f1 - v[i - 1]) / (x1 - x[i - 1]) : f1 - f0;
    float d1 = i + 2 < n ? width * (v[i + 2] - f0) / (x[i + 2] - x0) : f1 - f0;
    float a = 0.0f, b = 1.0f, t = 0.5f;
    for (int it = 0; it < 200; ++it) {
        if (!(t > a && t < b)) t = 0.5f * (a + b);
        float t2 = t * t, t3 = t2 * t;
        float Fhat = (2.0f * t3 - 3.0f * t2 + 1.0f) * f0 + (-2.0f * t3 + 3.0f * t2) * f1 + (t3 - 2.0f * t2 + t) * d0 + (t3 - t2) * d1;
        float fhat = (6.0f * t2 - 6.0f * t) * f0 + (-6.0f * t2 + 6.0f * t) * f1 + (3.0f * t2 - 4.0f * t + 1.0f) * d0 + (3.0f * t2 - 2.0f * t) * d1;
        if (std::fabs(Fhat - u) < 1.0e-6f || b - a < 1.0e-6f) break;
        if (Fhat - u < 0.0f) a = t; else b = t;
        t -= (Fhat - u) / fhat;
    }
    return x0 + t * width;
}
inline BssTable compute_beam_diffusion_bssrdf(float g, float eta) {
    BssTable t;
    t.rho_samples.resize(t.n_rho); t.radius_samples.resize(t.n_radius); t.rhoeff.resize(t.n_rho);
    t.profile.resize((size_t)t.n_rho * t.n_radius); t.profile_cdf.resize((size_t)t.n_rho * t.n_radius);
    t.radius_samples[0] = 0.0f; t.radius_samples[1] = 2.5e-3f;
    for (int i = 2; i < t.n_radius; ++i) t.radius_samples[i] = t.radius_samples[i - 1] * 1.2f;
    for (int i = 0; i < t.n_rho; ++i) t.rho_samples[i] = (1.0f - std::exp(-8.0f * (float)i / (float)(t.n_rho - 1))) / (1.0f - std::exp(-8.0f));
    for (int i = 0; i < t.n_rho; ++i)
        for (int j = 0; j < t.n_radius; ++j) {
            float rho = t.rho_samples[i], r = t.radius_samples[j];
            float v = 2.0f * 3.14159265358979323846f * r * (beam_diffusion_ss(rho, 1.0f - rho, g, eta, r) + beam_diffusion_ms(rho, 1.0f - rho, g, eta, r));
            t.profile[(size_t)i * t.n_radius + j] = std::isfinite(v) ? v : 0.0f;   // rho = 0 / r = 0 corners (0 * inf), as the Python twin
        }
    for (int i = 0; i < t.n_rho; ++i) t.rhoeff[i] = integrate_catmull_rom(t.n_radius, t.radius_samples.data(), &t.profile[(size_t)i * t.n_radius], &t.profile_cdf[(size_t)i * t.n_radius]);
    return t;
}
inline void subsurface_from_diffuse(const BssTable &t, const float kd[3], const float mfp[3], float sa[3], float ss[3]) {
    for (int c = 0; c < 3; ++c) { float rho = invert_catmull_rom(t.n_rho, t.rho_samples.data(), t.rhoeff.data(), kd[c]); ss[c] = rho / mfp[c]; sa[c] = (1.0f - rho) / mfp[c]; }
}
struct NamedMedium { float sigma_prime_s[3], sigma_a[3]; };
inline const std::map<std::string, NamedMedium> &named_media() {   // core/medium.rs:20-80 (rows used by the shipped scenes / tests)
    static const std::map<std::string, NamedMedium> m = {
        {"Skin1", {{0.74f, 0.88f, 1.01f}, {0.032f, 0.17f, 0.48f}}}, {"Skin2", {{1.09f, 1.59f, 1.79f}, {0.013f, 0.070f, 0.145f}}},
        {"Marble", {{2.19f, 2.62f, 3.00f}, {0.0021f, 0.0041f, 0.0071f}}}, {"Ketchup", {{0.18f, 0.07f, 0.03f}, {0.061f, 0.97f, 1.45f}}},
        {"Wholemilk", {{2.55f, 3.21f, 3.77f}, {0.0011f, 0.0024f, 0.014f}}}, {"Skimmilk", {{0.70f, 1.22f, 1.90f}, {0.0014f, 0.0025f, 0.0142f}}},
        {"Potato", {{0.68f, 0.70f, 0.55f}, {0.0024f, 0.0090f, 0.12f}}}};
    return m;
}

}  // namespace fe
