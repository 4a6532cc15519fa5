// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_math.h header).
// ref_bssrdf.h: TabulatedBSSRDF restated for the path integrator's subsurface branch.
//   core/interpolation.rs:3-50   catmull_rom_weights          (offset = idx - 1; entries with weight 0 are never read)
//   core/interpolation.rs:133-226 sample_catmull_rom_2d       (Newton-bisection; b starts at 1, test is t >= a && t <= b)
//   core/bssrdf.rs:22-36         fresnel_moment1
//   core/bssrdf.rs:285-321       TabulatedBSSRDF::new
//   core/bssrdf.rs:324-328       sw ; :330-332 sp ; :334-410 sample_sp ; :412-446 pdf_sp
//   core/bssrdf.rs:448-490       sr ; :492-500 sample_sr ; :502-541 pdf_sr
//   core/bssrdf.rs:559-574       sample_s (adapter BSDF at pi, pi.wo = shading.n)
//   core/bssrdf.rs:578-605       SeparableBSSRDFAdapter::f  (Reflection|Diffuse, default cosine sampling/pdf)
// The Newton-bisection loop is capped at kCatmullMaxIter iterations here and on the device (the reference loops until
// convergence; the cap is never reached on the reference's tables, it only bounds the worst case on the GPU).
// phi.cos()/phi.sin() go through the shared deterministic dm_cosf/dm_sinf like every other transcendental.
#pragma once
#include "ref_shading.h"

namespace ref {

constexpr int kCatmullMaxIter = 100;

// interpolation.rs:265-330 invert_catmull_rom: Newton-bisection on one spline segment (the loop is capped like the sampling one above)
inline Float invert_catmull_rom(int n, const Float *x, const Float *values, Float u) {
    if (!(u > values[0])) return x[0];
    else if (!(u < values[n - 1])) return x[n - 1];
    int i = find_interval(n, [&](int k) { return values[k] <= u; });
    Float x0 = x[i], x1 = x[i + 1], f0 = values[i], f1 = values[i + 1], width = x1 - x0;
    Float d0 = (i > 0) ? width * (f1 - values[i - 1]) / (x1 - x[i - 1]) : f1 - f0;
    Float d1 = (i + 2 < n) ? width * (values[i + 2] - f0) / (x[i + 2] - x0) : f1 - f0;
    Float a = 0.0f, b = 1.0f, t = 0.5f, Fhat, fhat;
    for (int it = 0; it < kCatmullMaxIter; ++it) {
        if (!(t > a && t < b)) t = 0.5f * (a + b);
        Float t2 = t * t, t3 = t2 * t;
        Fhat = (2.0f * t3 - 3.0f * t2 + 1.0f) * f0 + (-2.0f * t3 + 3.0f * t2) * f1 + (t3 - 2.0f * t2 + t) * d0 + (t3 - t2) * d1;
        fhat = (6.0f * t2 - 6.0f * t) * f0 + (-6.0f * t2 + 6.0f * t) * f1 + (3.0f * t2 - 4.0f * t + 1.0f) * d0 + (3.0f * t2 - 2.0f * t) * d1;
        if (std::fabs(Fhat - u) < 1.0e-6f || b - a < 1.0e-6f) break;
        if (Fhat - u < 0.0f) a = t; else b = t;
        t -= (Fhat - u) / fhat;
    }
    return x0 + t * width;
}
// bssrdf.rs:186-198 subsurface_from_diffuse
inline void subsurface_from_diffuse(const BssrdfTable &t, RGB rho_eff, RGB mfp, RGB &sigma_a, RGB &sigma_s) {
    for (int c = 0; c < 3; ++c) {
        Float rho = invert_catmull_rom(t.n_rho, t.rho_samples.data(), t.rhoeff.data(), rho_eff.c[c]);
        sigma_s.c[c] = rho / mfp.c[c];
        sigma_a.c[c] = (1.0f - rho) / mfp.c[c];
    }
}

inline bool catmull_rom_weights(int size, const Float *nodes, Float x, int &offset, Float w[4]) {
    if (!(x >= nodes[0] && x < nodes[size - 1])) return false;
    int idx = find_interval(size, [&](int i) { return nodes[i] <= x; });
    offset = idx - 1;
    Float x0 = nodes[idx], x1 = nodes[idx + 1];
    Float t = (x - x0) / (x1 - x0), t2 = t * t, t3 = t2 * t;
    w[1] = 2.0f * t3 - 3.0f * t2 + 1.0f;
    w[2] = -2.0f * t3 + 3.0f * t2;
    if (idx > 0) {
        Float w0 = (t3 - 2.0f * t2 + t) * (x1 - x0) / (x1 - nodes[idx - 1]);
        w[0] = -w0; w[2] += w0;
    } else {
        Float w0 = t3 - 2.0f * t2 + t;
        w[0] = 0.0f; w[1] -= w0; w[2] += w0;
    }
    if (idx + 2 < size) {
        Float w3 = (t3 - t2) * (x1 - x0) / (nodes[idx + 2] - x0);
        w[1] -= w3; w[3] = w3;
    } else {
        Float w3 = t3 - t2;
        w[1] -= w3; w[2] += w3; w[3] = 0.0f;
    }
    return true;
}

inline Float sample_catmull_rom_2d(int size1, int size2, const Float *nodes1, const Float *nodes2, const Float *values, const Float *cdf,
                                   Float alpha, Float u) {
    int offset = 0; Float weights[4] = {0, 0, 0, 0};
    if (!catmull_rom_weights(size1, nodes1, alpha, offset, weights)) return 0.0f;
    auto interpolate = [&](const Float *array, int idx) {
        Float value = 0.0f;
        for (int i = 0; i < 4; ++i)
            if (weights[i] != 0.0f) value += array[(size_t)(offset + i) * size2 + idx] * weights[i];
        return value;
    };
    Float maximum = interpolate(cdf, size2 - 1);
    u *= maximum;
    int idx = find_interval(size2, [&](int i) { return interpolate(cdf, i) <= u; });
    Float f0 = interpolate(values, idx), f1 = interpolate(values, idx + 1);
    Float x0 = nodes2[idx], x1 = nodes2[idx + 1];
    Float width = x1 - x0;
    u = (u - interpolate(cdf, idx)) / width;
    Float d0 = (idx > 0) ? width * (f1 - interpolate(values, idx - 1)) / (x1 - nodes2[idx - 1]) : f1 - f0;
    Float d1 = (idx + 2 < size2) ? width * (interpolate(values, idx + 2) - f0) / (nodes2[idx + 2] - x0) : f1 - f0;
    Float t = (f0 != f1) ? (f0 - std::sqrt(fmax_(f0 * f0 + 2.0f * u * (f1 - f0), 0.0f))) / (f0 - f1) : u / f0;
    Float a = 0.0f, b = 1.0f;
    for (int it = 0; it < kCatmullMaxIter; ++it) {
        if (!(t >= a && t <= b)) t = 0.5f * (a + b);
        Float Fhat = t * (f0 + t * (0.5f * d0 + t * ((1.0f / 3.0f) * (-2.0f * d0 - d1) + f1 - f0 + t * (0.25f * (d0 + d1) + 0.5f * (f0 - f1)))));
        Float fhat = f0 + t * (d0 + t * (-2.0f * d0 - d1 + 3.0f * (f1 - f0) + t * (d0 + d1 + 2.0f * (f0 - f1))));
        if (std::fabs(Fhat - u) < 1.0e-6f || b - a < 1.0e-6f) break;
        if (Fhat - u < 0.0f) a = t; else b = t;
        t -= (Fhat - u) / fhat;
    }
    return x0 + width * t;
}

inline Float fresnel_moment1(Float eta) {
    Float eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta3 * eta, eta5 = eta4 * eta;
    if (eta < 1.0f) return 0.45966f - 1.73965f * eta + 3.37668f * eta2 - 3.904945f * eta3 + 2.49277f * eta4 - 0.68441f * eta5;
    return -4.61686f + 11.1136f * eta - 10.4646f * eta2 + 5.11455f * eta3 - 1.27198f * eta4 + 0.12746f * eta5;
}

// SeparableBSSRDF::sw (bssrdf.rs:324-328) -- also what the adapter BxDF evaluates.
inline Float bssrdf_sw(Float eta, V3 w) {
    Float c = 1.0f - 2.0f * fresnel_moment1(1.0f / eta);
    return (1.0f - fr_dielectric(cos_theta(w), 1.0f, eta)) / (c * PI);
}

struct TabulatedBSSRDF {
    const BssrdfTable *table = nullptr;
    RGB sigma_t, rho;
    V3 ns, ss, ts;
    uint32_t material = PT_NONE;
    Float eta = 1;
    V3 po_p;
    // DisneyBSSRDF (materials/disney.rs:442-704): the same separable machinery with an analytic profile; R = color * diffuse
    // weight, d = scatter distance
    bool disney = false; RGB dR, dD;
    void init_disney(const SurfaceInteraction &s, uint32_t mat, Float eta_, RGB r, RGB d) {
        disney = true; material = mat; eta = eta_; dR = r; dD = d;
        ns = s.sh_n; ss = normalize(s.sh_dpdu); ts = cross(ns, ss); po_p = s.p;
    }
    static Float expf_(Float x) { return (Float)dm_expd((double)x); }

    void init(const SurfaceInteraction &s, uint32_t mat, Float eta_, RGB sigma_a, RGB sigma_s, const BssrdfTable *t) {
        table = t; material = mat; eta = eta_;
        sigma_t = sigma_a + sigma_s;
        ns = s.sh_n; ss = normalize(s.sh_dpdu); ts = cross(ns, ss);
        for (int i = 0; i < 3; ++i) rho.c[i] = (sigma_t.c[i] != 0.0f) ? sigma_s.c[i] / sigma_t.c[i] : 0.0f;
        po_p = s.p;
    }
    RGB sr(Float r) const {
        if (disney) {   // disney.rs:667-671
            if (r < 1.0e-6f) r = 1.0e-6f;
            RGB o;
            for (int i = 0; i < 3; ++i) o.c[i] = dR.c[i] * (expf_(-r / dD.c[i]) + expf_(-r / (dD.c[i] * 3.0f))) / (dD.c[i] * 8.0f * PI * r);
            return o;
        }
        RGB Sr(0.0f);
        for (int ch = 0; ch < 3; ++ch) {
            Float roptical = r * sigma_t.c[ch];
            int rho_off = 0, rad_off = 0; Float rw[4] = {0, 0, 0, 0}, dw[4] = {0, 0, 0, 0};
            if (!catmull_rom_weights(table->n_rho, table->rho_samples.data(), rho.c[ch], rho_off, rw) ||
                !catmull_rom_weights(table->n_radius, table->radius_samples.data(), roptical, rad_off, dw)) continue;
            Float s = 0.0f;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    Float weight = rw[i] * dw[j];
                    if (weight != 0.0f) s += weight * table->eval_profile(rho_off + i, rad_off + j);
                }
            if (roptical != 0.0f) s /= 2.0f * PI * roptical;
            Sr.c[ch] = s;
        }
        Sr = Sr * (sigma_t * sigma_t);
        return Sr.clamps(0.0f, INF);
    }
    Float sample_sr(int ch, Float u) const {
        if (disney) {   // disney.rs:673-681
            if (u < 0.25f) { u = fmin_(u * 4.0f, ONE_MINUS_EPSILON); return dD.c[ch] * dm_logf(1.0f / (1.0f - u)); }
            u = fmin_((u - 0.25f) / 0.75f, ONE_MINUS_EPSILON);
            return 3.0f * dD.c[ch] * dm_logf(1.0f / (1.0f - u));
        }
        if (sigma_t.c[ch] == 0.0f) return -1.0f;
        return sample_catmull_rom_2d(table->n_rho, table->n_radius, table->rho_samples.data(), table->radius_samples.data(),
                                     table->profile.data(), table->profile_cdf.data(), rho.c[ch], u) / sigma_t.c[ch];
    }
    Float pdf_sr(int ch, Float r) const {
        if (disney) {   // disney.rs:683-686
            Float d = dD.c[ch];
            return 0.25f * expf_(-r / d) / (2.0f * PI * d * r) + 0.75f * expf_(-r / (3.0f * d)) / (6.0f * PI * d * r);
        }
        Float roptical = r * sigma_t.c[ch];
        int rho_off = 0, rad_off = 0; Float rw[4] = {0, 0, 0, 0}, dw[4] = {0, 0, 0, 0};
        if (!catmull_rom_weights(table->n_rho, table->rho_samples.data(), rho.c[ch], rho_off, rw) ||
            !catmull_rom_weights(table->n_radius, table->radius_samples.data(), roptical, rad_off, dw)) return 0.0f;
        Float s = 0.0f, rho_eff = 0.0f;
        for (int i = 0; i < 4; ++i) {
            if (rw[i] == 0.0f) continue;
            rho_eff += table->rhoeff[rho_off + i] * rw[i];
            for (int j = 0; j < 4; ++j) {
                if (dw[j] == 0.0f) continue;
                s += table->eval_profile(rho_off + i, rad_off + j) * rw[i] * dw[j];
            }
        }
        if (roptical != 0.0f) s /= 2.0f * PI * roptical;
        return fmax_(s * sigma_t.c[ch] * sigma_t.c[ch] / rho_eff, 0.0f);
    }
    Float pdf_sp(V3 pi_p, V3 pi_n) const {
        V3 d = po_p - pi_p;
        Float dl[3] = {dot(ss, d), dot(ts, d), dot(ns, d)};
        Float nl[3] = {dot(ss, pi_n), dot(ts, pi_n), dot(ns, pi_n)};
        Float rproj[3] = {std::sqrt(dl[1] * dl[1] + dl[2] * dl[2]), std::sqrt(dl[2] * dl[2] + dl[0] * dl[0]), std::sqrt(dl[0] * dl[0] + dl[1] * dl[1])};
        Float pdf = 0.0f;
        const Float axisprob[3] = {0.25f, 0.25f, 0.5f};
        const Float chprob = 1.0f / 3.0f;
        for (int axis = 0; axis < 3; ++axis)
            for (int ch = 0; ch < 3; ++ch) pdf += pdf_sr(ch, rproj[axis]) * std::fabs(nl[axis]) * chprob * axisprob[axis];
        return pdf;
    }
    // sample_sp up to the probe segment: returns false when the sample is rejected before any ray is traced.
    bool probe_segment(Float u1, P2 u2, V3 &start, V3 &target, Float &u1n) const {
        V3 vx, vy, vz;
        if (u1 < 0.5f) { vx = ss; vy = ts; vz = ns; u1n = u1 * 2.0f; }
        else if (u1 < 0.75f) { vx = ts; vy = ns; vz = ss; u1n = (u1 - 0.5f) * 4.0f; }
        else { vx = ns; vy = ss; vz = ts; u1n = (u1 - 0.75f) * 4.0f; }
        int ch = clampv((int)(u1n * 3.0f), 0, 2);
        u1n = u1n * 3.0f - (Float)ch;
        Float r = sample_sr(ch, u2.x);
        if (r < 0.0f) return false;
        Float phi = 2.0f * PI * u2.y;
        Float rmax = sample_sr(ch, 0.999f);
        if (r >= rmax) return false;
        Float l = 2.0f * std::sqrt(rmax * rmax - r * r);
        start = po_p + (vx * dm_cosf(phi) + vy * dm_sinf(phi)) * r - vz * l * 0.5f;
        target = start + vz * l;
        return true;
    }
};

}  // namespace ref
