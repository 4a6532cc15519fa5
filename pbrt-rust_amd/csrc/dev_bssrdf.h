// dev_bssrdf.h -- TabulatedBSSRDF on the device.
//   core/interpolation.rs:3-50    catmull_rom_weights
//   core/interpolation.rs:133-226 sample_catmull_rom_2d (Newton-bisection, capped at kCatmullMaxIter like the oracle)
//   core/bssrdf.rs:22-36 fresnel_moment1 ; :285-321 TabulatedBSSRDF::new ; :324-328 sw ; :334-375 sample_sp (segment)
//   core/bssrdf.rs:412-446 pdf_sp ; :448-490 sr ; :492-500 sample_sr ; :502-541 pdf_sr
//   core/bssrdf.rs:578-605 SeparableBSSRDFAdapter (the one-lobe BSDF at the sampled exit point)
// The table itself (photon beam diffusion, bssrdf.rs:138-188) is material-creation-time work done by the host and
// handed over through PtBSSRDFTable.
#pragma once
#include "dev_bsdf.h"

namespace ptd {

constexpr int kCatmullMaxIter = 100;

template <class Pred> PT_DEV int find_interval_pred(int size, Pred pred) {  // pbrt.rs:184-204
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (pred(middle)) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int r = first - 1;
    return r < 0 ? 0 : (r > size - 2 ? size - 2 : r);
}

PT_DEV bool catmull_rom_weights(int size, const float *nodes, float x, int &offset, float w[4]) {
    if (!(x >= nodes[0] && x < nodes[size - 1])) return false;
    int idx = find_interval_pred(size, [&](int i) { return nodes[i] <= x; });
    offset = idx - 1;
    float x0 = nodes[idx], x1 = nodes[idx + 1];
    float t = (x - x0) / (x1 - x0), t2 = t * t, t3 = t2 * t;
    w[1] = 2.0f * t3 - 3.0f * t2 + 1.0f;
    w[2] = -2.0f * t3 + 3.0f * t2;
    if (idx > 0) {
        float w0 = (t3 - 2.0f * t2 + t) * (x1 - x0) / (x1 - nodes[idx - 1]);
        w[0] = -w0; w[2] += w0;
    } else {
        float w0 = t3 - 2.0f * t2 + t;
        w[0] = 0.0f; w[1] -= w0; w[2] += w0;
    }
    if (idx + 2 < size) {
        float w3 = (t3 - t2) * (x1 - x0) / (nodes[idx + 2] - x0);
        w[1] -= w3; w[3] = w3;
    } else {
        float w3 = t3 - t2;
        w[1] -= w3; w[2] += w3; w[3] = 0.0f;
    }
    return true;
}

PT_DEV float sample_catmull_rom_2d(int size1, int size2, const float *nodes1, const float *nodes2, const float *values, const float *cdf,
                                   float alpha, float u) {
    int offset = 0; float weights[4] = {0, 0, 0, 0};
    if (!catmull_rom_weights(size1, nodes1, alpha, offset, weights)) return 0.0f;
    auto interpolate = [&](const float *array, int idx) {
        float value = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (weights[i] != 0.0f) value += array[(size_t)(offset + i) * size2 + idx] * weights[i];
        return value;
    };
    float maximum = interpolate(cdf, size2 - 1);
    u *= maximum;
    int idx = find_interval_pred(size2, [&](int i) { return interpolate(cdf, i) <= u; });
    float f0 = interpolate(values, idx), f1 = interpolate(values, idx + 1);
    float x0 = nodes2[idx], x1 = nodes2[idx + 1];
    float width = x1 - x0;
    u = (u - interpolate(cdf, idx)) / width;
    float d0 = (idx > 0) ? width * (f1 - interpolate(values, idx - 1)) / (x1 - nodes2[idx - 1]) : f1 - f0;
    float d1 = (idx + 2 < size2) ? width * (interpolate(values, idx + 2) - f0) / (nodes2[idx + 2] - x0) : f1 - f0;
    float t = (f0 != f1) ? (f0 - sqrtf(maxf(f0 * f0 + 2.0f * u * (f1 - f0), 0.0f))) / (f0 - f1) : u / f0;
    float a = 0.0f, b = 1.0f;
    for (int it = 0; it < kCatmullMaxIter; ++it) {
        if (!(t >= a && t <= b)) t = 0.5f * (a + b);
        float Fhat = t * (f0 + t * (0.5f * d0 + t * ((1.0f / 3.0f) * (-2.0f * d0 - d1) + f1 - f0 + t * (0.25f * (d0 + d1) + 0.5f * (f0 - f1)))));
        float fhat = f0 + t * (d0 + t * (-2.0f * d0 - d1 + 3.0f * (f1 - f0) + t * (d0 + d1 + 2.0f * (f0 - f1))));
        if (fabsf(Fhat - u) < 1.0e-6f || b - a < 1.0e-6f) break;
        if (Fhat - u < 0.0f) a = t; else b = t;
        t -= (Fhat - u) / fhat;
    }
    return x0 + width * t;
}

PT_DEV float dm_expf_b(float x) { return (float)dm_expd((double)x); }   // f32::exp through the shared f64 exp
PT_DEV float fresnel_moment1(float eta) {
    float eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta3 * eta, eta5 = eta4 * eta;
    if (eta < 1.0f) return 0.45966f - 1.73965f * eta + 3.37668f * eta2 - 3.904945f * eta3 + 2.49277f * eta4 - 0.68441f * eta5;
    return -4.61686f + 11.1136f * eta - 10.4646f * eta2 + 5.11455f * eta3 - 1.27198f * eta4 + 0.12746f * eta5;
}
PT_DEV float bssrdf_sw(float eta, V3 w) {
    float c = 1.0f - 2.0f * fresnel_moment1(1.0f / eta);
    return (1.0f - fr_dielectric(cos_theta(w), 1.0f, eta)) / (c * kPi);
}

// interpolation.rs:265-330 invert_catmull_rom (Newton-bisection on one spline segment, capped like the sampling loop) and
// bssrdf.rs:186-198 subsurface_from_diffuse: a kdsubsurface material whose Kd / mfp are textures converts at every hit.
PT_DEV float invert_catmull_rom(int n, const float *x, const float *values, float u) {
    if (!(u > values[0])) return x[0];
    else if (!(u < values[n - 1])) return x[n - 1];
    const int i = find_interval_pred(n, [&](int k) { return values[k] <= u; });
    const float x0 = x[i], x1 = x[i + 1], f0 = values[i], f1 = values[i + 1], width = x1 - x0;
    const float d0 = (i > 0) ? width * (f1 - values[i - 1]) / (x1 - x[i - 1]) : f1 - f0;
    const float d1 = (i + 2 < n) ? width * (values[i + 2] - f0) / (x[i + 2] - x0) : f1 - f0;
    float a = 0.0f, b = 1.0f, t = 0.5f;
#pragma unroll 1
    for (int it = 0; it < kCatmullMaxIter; ++it) {
        if (!(t > a && t < b)) t = 0.5f * (a + b);
        const float t2 = t * t, t3 = t2 * t;
        const float Fhat = (2.0f * t3 - 3.0f * t2 + 1.0f) * f0 + (-2.0f * t3 + 3.0f * t2) * f1 + (t3 - 2.0f * t2 + t) * d0 + (t3 - t2) * d1;
        const float fhat = (6.0f * t2 - 6.0f * t) * f0 + (-6.0f * t2 + 6.0f * t) * f1 + (3.0f * t2 - 4.0f * t + 1.0f) * d0 + (3.0f * t2 - 2.0f * t) * d1;
        if (fabsf(Fhat - u) < 1.0e-6f || b - a < 1.0e-6f) break;
        if (Fhat - u < 0.0f) a = t; else b = t;
        t -= (Fhat - u) / fhat;
    }
    return x0 + t * width;
}
PT_DEV void subsurface_from_diffuse(const DevBssTable &t, RGB rho_eff, RGB mfp, RGB &sigma_a, RGB &sigma_s) {
    const float re[3] = {rho_eff.r, rho_eff.g, rho_eff.b}, mf[3] = {mfp.r, mfp.g, mfp.b};
    float sa[3], ss[3];
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        const float rho = invert_catmull_rom(t.n_rho, t.rho_samples, t.rhoeff, re[c]);
        ss[c] = rho / mf[c];
        sa[c] = (1.0f - rho) / mf[c];
    }
    sigma_a = RGB(sa[0], sa[1], sa[2]); sigma_s = RGB(ss[0], ss[1], ss[2]);
}

struct DevBssrdf {
    DevBssTable tb;
    float sigma_t[3], rho[3];
    bool disney = false; float dR[3], dD[3];   // DisneyBSSRDF (disney.rs:442-704): R = color * diffuse weight, d = scatter distance
    V3 ns, ss, ts, po_p;
    float eta;

    // subsurface.rs:100-103 (sigma * scale) + TabulatedBSSRDF::new
    // siga / sigs: the evaluated sigma_a / sigma_s textures (constants: the material fields)
    // m.kd_subsurface: siga / sigs are what subsurface_from_diffuse gave at the hit (kdsubsurface.rs:96-99): taken as they are
    PT_DEV void init_medium(const PtMaterial &m, const DevBssTable *tables, RGB siga, RGB sigs) {
        tb = tables[m.bssrdf_table]; eta = m.eta;
        const float sa3[3] = {siga.r, siga.g, siga.b}, ss3[3] = {sigs.r, sigs.g, sigs.b};
        const bool raw = m.kd_subsurface != 0u;
        for (int i = 0; i < 3; ++i) {
            float sa = raw ? sa3[i] : clampf(sa3[i], 0.0f, PT_INF) * m.scale, ss_ = raw ? ss3[i] : clampf(ss3[i], 0.0f, PT_INF) * m.scale;
            sigma_t[i] = sa + ss_;
            rho[i] = (sigma_t[i] != 0.0f) ? ss_ / sigma_t[i] : 0.0f;
        }
    }
    // DisneyMaterial's BSSRDF (disney.rs:768-776): constant color; the weights are the material's constants
    PT_DEV void init_disney(const PtMaterial &m) {
        disney = true; eta = m.eta;
        const float dw = (1.0f - m.disney[PT_DS_METALLIC]) * (1.0f - m.disney[PT_DS_SPECTRANS]);
        for (int i = 0; i < 3; ++i) { dR[i] = clampf(m.kd[i], 0.0f, PT_INF) * dw; dD[i] = m.disney_scatter[i]; }
    }
    PT_DEV void init_material(const PtMaterial &m, const DevBssTable *tables) {
        if (m.type == PT_MAT_DISNEY) init_disney(m); else init_medium(m, tables, rgb3(m.sigma_a), rgb3(m.sigma_s));
    }
    PT_DEV void init_frame(const SurfaceInteraction &s) { ns = s.sh_n; ss = normalize(s.sh_dpdu); ts = cross(ns, ss); po_p = s.p; }

    PT_DEV RGB sr(float r) const {
        if (disney) {   // disney.rs:667-671
            if (r < 1.0e-6f) r = 1.0e-6f;
            float o[3];
            for (int i = 0; i < 3; ++i) o[i] = dR[i] * (dm_expf_b(-r / dD[i]) + dm_expf_b(-r / (dD[i] * 3.0f))) / (dD[i] * 8.0f * kPi * r);
            return RGB(o[0], o[1], o[2]);
        }
        float out[3] = {0.0f, 0.0f, 0.0f};
        for (int ch = 0; ch < 3; ++ch) {
            float roptical = r * sigma_t[ch];
            int rho_off = 0, rad_off = 0; float rw[4] = {0, 0, 0, 0}, dw[4] = {0, 0, 0, 0};
            if (!catmull_rom_weights(tb.n_rho, tb.rho_samples, rho[ch], rho_off, rw) ||
                !catmull_rom_weights(tb.n_radius, tb.radius_samples, roptical, rad_off, dw)) continue;
            float s = 0.0f;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    float weight = rw[i] * dw[j];
                    if (weight != 0.0f) s += weight * tb.profile[(size_t)(rho_off + i) * tb.n_radius + (rad_off + j)];
                }
            if (roptical != 0.0f) s /= 2.0f * kPi * roptical;
            out[ch] = s;
        }
        RGB Sr(out[0] * (sigma_t[0] * sigma_t[0]), out[1] * (sigma_t[1] * sigma_t[1]), out[2] * (sigma_t[2] * sigma_t[2]));
        return Sr.clamps(0.0f, PT_INF);
    }
    PT_DEV float sample_sr(int ch, float u) const {
        if (disney) {   // disney.rs:673-681
            const float d = ch == 0 ? dD[0] : ch == 1 ? dD[1] : dD[2];
            if (u < 0.25f) { u = minf(u * 4.0f, kOneMinusEps); return d * dm_logf(1.0f / (1.0f - u)); }
            u = minf((u - 0.25f) / 0.75f, kOneMinusEps);
            return 3.0f * d * dm_logf(1.0f / (1.0f - u));
        }
        if (sigma_t[ch] == 0.0f) return -1.0f;
        return sample_catmull_rom_2d(tb.n_rho, tb.n_radius, tb.rho_samples, tb.radius_samples, tb.profile, tb.profile_cdf, rho[ch], u) / sigma_t[ch];
    }
    PT_DEV float pdf_sr(int ch, float r) const {
        if (disney) {   // disney.rs:683-686
            const float d = ch == 0 ? dD[0] : ch == 1 ? dD[1] : dD[2];
            return 0.25f * dm_expf_b(-r / d) / (2.0f * kPi * d * r) + 0.75f * dm_expf_b(-r / (3.0f * d)) / (6.0f * kPi * d * r);
        }
        float roptical = r * sigma_t[ch];
        int rho_off = 0, rad_off = 0; float rw[4] = {0, 0, 0, 0}, dw[4] = {0, 0, 0, 0};
        if (!catmull_rom_weights(tb.n_rho, tb.rho_samples, rho[ch], rho_off, rw) ||
            !catmull_rom_weights(tb.n_radius, tb.radius_samples, roptical, rad_off, dw)) return 0.0f;
        float s = 0.0f, rho_eff = 0.0f;
        for (int i = 0; i < 4; ++i) {
            if (rw[i] == 0.0f) continue;
            rho_eff += tb.rhoeff[rho_off + i] * rw[i];
            for (int j = 0; j < 4; ++j) {
                if (dw[j] == 0.0f) continue;
                s += tb.profile[(size_t)(rho_off + i) * tb.n_radius + (rad_off + j)] * rw[i] * dw[j];
            }
        }
        if (roptical != 0.0f) s /= 2.0f * kPi * roptical;
        return maxf(s * sigma_t[ch] * sigma_t[ch] / rho_eff, 0.0f);
    }
    PT_DEV float pdf_sp(V3 pi_p, V3 pi_n) const {
        V3 d = po_p - pi_p;
        float dl[3] = {dot(ss, d), dot(ts, d), dot(ns, d)};
        float nl[3] = {dot(ss, pi_n), dot(ts, pi_n), dot(ns, pi_n)};
        float rproj[3] = {sqrtf(dl[1] * dl[1] + dl[2] * dl[2]), sqrtf(dl[2] * dl[2] + dl[0] * dl[0]), sqrtf(dl[0] * dl[0] + dl[1] * dl[1])};
        float pdf = 0.0f;
        const float axisprob[3] = {0.25f, 0.25f, 0.5f};
        const float chprob = 1.0f / 3.0f;
        for (int axis = 0; axis < 3; ++axis)
            for (int ch = 0; ch < 3; ++ch) pdf += pdf_sr(ch, rproj[axis]) * fabsf(nl[axis]) * chprob * axisprob[axis];
        return pdf;
    }
    // sample_sp up to the probe segment (bssrdf.rs:337-365); false: S = 0 before any ray is traced.
    PT_DEV bool probe_segment(float u1, P2 u2, V3 &start, V3 &target, float &u1n) const {
        V3 vx, vy, vz;
        if (u1 < 0.5f) { vx = ss; vy = ts; vz = ns; u1n = u1 * 2.0f; }
        else if (u1 < 0.75f) { vx = ts; vy = ns; vz = ss; u1n = (u1 - 0.5f) * 4.0f; }
        else { vx = ns; vy = ss; vz = ts; u1n = (u1 - 0.75f) * 4.0f; }
        int ch = (int)(u1n * 3.0f); ch = ch < 0 ? 0 : (ch > 2 ? 2 : ch);
        u1n = u1n * 3.0f - (float)ch;
        float r = sample_sr(ch, u2.x);
        if (r < 0.0f) return false;
        float phi = 2.0f * kPi * u2.y;
        float rmax = sample_sr(ch, 0.999f);
        if (r >= rmax) return false;
        float l = 2.0f * sqrtf(rmax * rmax - r * r);
        float sn, cs; dm_sincosf(phi, sn, cs);
        start = po_p + (vx * cs + vy * sn) * r - vz * l * 0.5f;
        target = start + vz * l;
        return true;
    }
};

// BSDF::new(pi, 1.0) holding the single SeparableBSSRDFAdapter lobe (Reflection | Diffuse); same interface as Bsdf<MAXL>
// restricted to what estimate_direct / path.rs use. Sampling and pdf are BxDF's defaults (reflection.rs:392-403,439-445).
struct BssrdfAdapterBsdf {
    float eta;       // the BSSRDF's eta (adapter scales by eta^2 in Radiance mode, bssrdf.rs:597-600)
    V3 ns, ng, ss, ts;
    static constexpr int kType = BSDF_REFLECTION | BSDF_DIFFUSE;
    PT_DEV void init(const SurfaceInteraction &si, float bssrdf_eta) {
        eta = bssrdf_eta; ns = si.sh_n; ss = normalize(si.sh_dpdu); ng = si.n; ts = cross(ns, ss);
    }
    PT_DEV static bool matches(int flags) { return (kType & flags) == kType; }
    PT_DEV V3 to_local(V3 v) const { return V3(dot(v, ss), dot(v, ts), dot(v, ns)); }
    PT_DEV V3 to_world(V3 v) const {
        return V3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
    PT_DEV RGB lobe_f(V3 wi) const { return RGB(bssrdf_sw(eta, wi)) * (eta * eta); }
    PT_DEV static float lobe_pdf(V3 wo, V3 wi) { return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * kInvPi : 0.0f; }
    PT_DEV RGB f(V3 wow, V3 wiw, int flags) const {
        V3 wi = to_local(wiw), wo = to_local(wow);
        if (wo.z == 0.0f) return RGB(0.0f);
        bool refl = dot(wiw, ng) * dot(wow, ng) > 0.0f;
        return (matches(flags) && refl) ? lobe_f(wi) : RGB(0.0f);
    }
    PT_DEV float pdf(V3 wow, V3 wiw, int flags) const {
        V3 wo = to_local(wow), wi = to_local(wiw);
        if (wo.z == 0.0f) return 0.0f;
        return matches(flags) ? lobe_pdf(wo, wi) : 0.0f;
    }
    PT_DEV RGB f_pdf(V3 wow, V3 wiw, int flags, float &p) const { p = pdf(wow, wiw, flags); return f(wow, wiw, flags); }
    PT_DEV RGB sample_f(V3 wow, V3 &wiw, P2 u, float &pdf, int ty, int &sampled) const {
        if (!matches(ty)) { pdf = 0.0f; sampled = 0; return RGB(0.0f); }
        P2 ur(minf(u.x, kOneMinusEps), u.y);
        V3 wo = to_local(wow);
        if (wo.z == 0.0f) return RGB(0.0f);
        sampled = kType;
        V3 wi = cosine_sample_hemisphere(ur);
        if (wo.z < 0.0f) wi.z *= -1.0f;
        pdf = lobe_pdf(wo, wi);
        if (pdf == 0.0f) { sampled = 0; return RGB(0.0f); }
        wiw = to_world(wi);
        bool refl = dot(wiw, ng) * dot(wow, ng) > 0.0f;
        return refl ? lobe_f(wi) : RGB(0.0f);
    }
};

}  // namespace ptd
