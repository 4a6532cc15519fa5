// dev_bsdf.h -- BxDF lobes, BSDF frame/evaluation/sampling and material -> BSDF assembly on device.
//   core/reflection.rs:29-190 (Fresnel, refract, trig helpers), :384-446 (BxDF defaults),
//   :612-1222 (lobes), :1506-1689 (BSDF); core/microfacet.rs:249-406 (Trowbridge-Reitz, visible-normal
//   sampling); core/sampling.rs:153-193; materials/{matte,mirror,glass,plastic,metal,uber,substrate}.rs.
#pragma once
#include "dev_scene.h"

namespace ptd {

enum { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };
enum LobeKind : uint8_t { LB_LAMBERT_R, LB_LAMBERT_T, LB_OREN_NAYAR, LB_SPEC_R, LB_SPEC_T, LB_FRESNEL_SPEC, LB_MICRO_R, LB_MICRO_T, LB_FRESNEL_BLEND,
                          LB_DISNEY_DIFFUSE, LB_DISNEY_FAKESS, LB_DISNEY_RETRO, LB_DISNEY_SHEEN, LB_DISNEY_CLEARCOAT };   // materials/disney.rs
enum FresnelKind : uint8_t { FR_NOOP, FR_DIELECTRIC, FR_CONDUCTOR, FR_DISNEY };

#ifdef PT_OUTLINE_BSDF   // experiment: the scalar microfacet / Fresnel helpers as real functions (like the f64 transcendentals, dev_math.h)
#define PT_DEVO __device__ __noinline__ inline
#else
#define PT_DEVO PT_DEV
#endif
PT_DEV V3 cosine_sample_hemisphere(P2 u) {  // sampling.rs:188-193
    P2 d = concentric_sample_disk(u);
    float z = sqrtf(maxf(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return V3(d.x, d.y, z);
}
PT_DEV float power_heuristic(float fpdf, float gpdf) {  // sampling.rs:328-333 with nf = ng = 1
    float f = 1.0f * fpdf, g = 1.0f * gpdf;
    return (f * f) / (f * f + g * g);
}

PT_DEV float cos_theta(V3 w) { return w.z; }
PT_DEV float cos2_theta(V3 w) { return w.z * w.z; }
PT_DEV float abs_cos_theta(V3 w) { return fabsf(w.z); }
PT_DEV float sin2_theta(V3 w) { return maxf(1.0f - cos2_theta(w), 0.0f); }
PT_DEV float sin_theta(V3 w) { return sqrtf(sin2_theta(w)); }
PT_DEV float tan_theta(V3 w) { return sin_theta(w) / cos_theta(w); }
PT_DEV float tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
PT_DEV float cos_phi(V3 w) { float s = sin_theta(w); return (s == 0.0f) ? 1.0f : clampf(w.x / s, -1.0f, 1.0f); }
PT_DEV float sin_phi(V3 w) { float s = sin_theta(w); return (s == 0.0f) ? 0.0f : clampf(w.y / s, -1.0f, 1.0f); }
PT_DEV float cos2_phi(V3 w) { float c = cos_phi(w); return c * c; }
PT_DEV float sin2_phi(V3 w) { float s = sin_phi(w); return s * s; }
PT_DEV bool same_hemisphere(V3 w, V3 wp) { return w.z * wp.z > 0.0f; }
PT_DEV V3 reflect(V3 wo, V3 n) { return -wo + n * 2.0f * dot(wo, n); }
PT_DEV bool refract(V3 wi, V3 n, float eta, V3 &wt) {  // reflection.rs:160-174
    float cos_i = dot(n, wi);
    float sin2_i = maxf(1.0f - cos_i * cos_i, 0.0f);
    float sin2_t = eta * eta * sin2_i;
    if (sin2_t >= 1.0f) return false;
    float cos_t = sqrtf(1.0f - sin2_t);
    wt = n * (eta * cos_i - cos_t) + (-wi) * eta;
    return true;
}
PT_DEVO float fr_dielectric(float cos_i, float etai, float etat) {  // reflection.rs:29-52
    cos_i = clampf(cos_i, -1.0f, 1.0f);
    if (!(cos_i > 0.0f)) { float t = etai; etai = etat; etat = t; cos_i = fabsf(cos_i); }
    float sin_i = sqrtf(maxf(0.0f, 1.0f - cos_i * cos_i));
    float sin_t = etai / etat * sin_i;
    if (sin_t >= 1.0f) return 1.0f;
    float cos_t = sqrtf(maxf(0.0f, 1.0f - sin_t * sin_t));
    float rparl = ((etat * cos_i) - (etai * cos_t)) / ((etat * cos_i) + (etai * cos_t));
    float rperp = ((etai * cos_i) - (etat * cos_t)) / ((etai * cos_i) + (etat * cos_t));
    return (rparl * rparl + rperp * rperp) / 2.0f;
}
PT_DEVO RGB fr_conductor(float cos_i, RGB etai, RGB etat, RGB k) {  // reflection.rs:54-76
    cos_i = clampf(cos_i, -1.0f, 1.0f);
    RGB eta = etat / etai, etak = k / etai;
    float cos2 = cos_i * cos_i, sin2 = 1.0f - cos2;
    RGB eta2 = eta * eta, etak2 = etak * etak;
    RGB t0 = eta2 - etak2 - RGB(sin2);
    RGB a2plusb2 = sqrt_rgb(t0 * t0 + eta2 * etak2 * 4.0f);
    RGB t1 = a2plusb2 + RGB(cos2);
    RGB a = sqrt_rgb((a2plusb2 + t0) * 0.5f);
    RGB t2 = a * cos_i * 2.0f;
    RGB rs = (t1 - t2) / (t1 + t2);
    RGB t3 = a2plusb2 * cos2 + RGB(sin2 * sin2);
    RGB t4 = t2 * sin2;
    RGB rp = rs * (t3 - t4) / (t3 + t4);
    return (rp + rs) * 0.5f;
}

struct Lobe {
    uint8_t kind, type, fresnel;
    uint8_t sepg;    // DisneyMicrofacetDistribution: separable masking-shadowing G = G1(wo) G1(wi) (disney.rs:376-379)
    RGB r, t;        // R | T | Rd ; FresnelBlend: t = Rs ; conductor: r=1, t unused
    float ax, ay;    // Trowbridge-Reitz alphas
    float etaa, etab;  // dielectric etas (etai/etat of the Fresnel term, or lobe etaA/etaB)
    RGB ck, ce;      // conductor k, eta
    float A, B;      // Oren-Nayar; Disney: A = roughness (fake-ss, retro) / clearcoat gloss / Fresnel metallic, B = clearcoat weight
    PT_DEV bool matches(int flags) const { return (type & flags) == type; }
};

// ---- materials/disney.rs:31-52 helpers
PT_DEV float schlick_weight(float c) { const float m = clampf(1.0f - c, 0.0f, 1.0f); return (m * m) * (m * m) * m; }
PT_DEV float lerp_t(float t, float x, float y) { return x * (1.0f - t) + y * t; }                       // pbrt.rs:136-144
PT_DEV RGB lerp_rgb(float t, RGB x, RGB y) { return x * (1.0f - t) + y * t; }
PT_DEV float fr_schlick(float r0, float c) { return lerp_t(schlick_weight(c), r0, 1.0f); }
PT_DEV RGB fr_schlicks(RGB r0, float c) { return lerp_rgb(schlick_weight(c), r0, RGB(1.0f)); }
PT_DEV float gtr1(float c, float alpha) {       // disney.rs:222-226
    const float a2 = alpha * alpha;
    return (a2 - 1.0f) / (kPi * dm_logf(a2) * (1.0f + (a2 - 1.0f) * c * c));
}
PT_DEV float smithg_ggx(float c, float alpha) {  // disney.rs:228-233 (no square root, as written there)
    const float a2 = alpha * alpha, c2 = c * c;
    return 1.0f / (c + (a2 + c2 - a2 * c2));
}

template <bool FULL = true> PT_DEV RGB fresnel_eval(const Lobe &l, float cosi, float fi, float ft) {
    if (l.fresnel == FR_NOOP) return RGB(1.0f);
    if (FULL && l.fresnel == FR_DISNEY) return lerp_rgb(l.A, RGB(fr_dielectric(cosi, 1.0f, l.etab)), fr_schlicks(l.ce, cosi));   // DisneyFresnel (disney.rs:296-303)
    if (l.fresnel == FR_DIELECTRIC) return RGB(fr_dielectric(cosi, fi, ft));
    return fr_conductor(fabsf(cosi), RGB(1.0f), l.ce, l.ck);
}

// ---- Trowbridge-Reitz (microfacet.rs:249-406, samplevis = true) -----------------------------------------
PT_DEV float roughness_to_alpha(float roughness) {  // microfacet.rs:334-340
    roughness = maxf(roughness, 1.0e-3f);
    float x = dm_logf(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
PT_DEVO float tr_d(float ax, float ay, V3 wh) {
    float t2 = tan2_theta(wh);
    if (__builtin_isinf(t2)) return 0.0f;
    float c4 = cos2_theta(wh) * cos2_theta(wh);
    float e = (cos2_phi(wh) / (ax * ax) + sin2_phi(wh) / (ay * ay)) * t2;
    return 1.0f / (kPi * ax * ay * c4 * (1.0f + e) * (1.0f + e));
}
PT_DEVO float tr_lambda(float ax, float ay, V3 w) {
    float abs_tan = fabsf(tan_theta(w));
    if (__builtin_isinf(abs_tan)) return 0.0f;
    float alpha = sqrtf(cos2_phi(w) * ax * ax + sin2_phi(w) * ay * ay);
    float a2t2 = (alpha * abs_tan) * (alpha * abs_tan);
    return (-1.0f + sqrtf(1.0f + a2t2)) / 2.0f;
}
PT_DEV float tr_g1(float ax, float ay, V3 w) { return 1.0f / (1.0f + tr_lambda(ax, ay, w)); }
PT_DEV float tr_g(float ax, float ay, V3 wo, V3 wi) { return 1.0f / (1.0f + tr_lambda(ax, ay, wo) + tr_lambda(ax, ay, wi)); }
PT_DEV float tr_pdf(float ax, float ay, V3 wo, V3 wh) { return tr_d(ax, ay, wh) * tr_g1(ax, ay, wo) * abs_dot(wo, wh) / abs_cos_theta(wo); }
PT_DEV void tr_sample11(float cos_t, float u1, float u2, float &sx, float &sy) {  // microfacet.rs:249-291
    if (cos_t > 0.9999f) {
        float r = sqrtf(u1 / (1.0f - u1));
        float phi = 6.28318530718f * u2;
        float s, c; dm_sincosf(phi, s, c);
        sx = r * c; sy = r * s;
        return;
    }
    float sin_t = sqrtf(maxf(0.0f, 1.0f - cos_t * cos_t));
    float tan_t = sin_t / cos_t;
    float a = 1.0f / tan_t;
    float G1 = 2.0f / (1.0f + sqrtf(1.0f + 1.0f / (a * a)));
    float A = 2.0f * u1 / G1 - 1.0f;
    float tmp = 1.0f / (A * A - 1.0f);
    if (tmp > 1.0e10f) tmp = 1.0e10f;
    float B = tan_t;
    float D = sqrtf(maxf(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.0f));
    float sx1 = B * tmp - D, sx2 = B * tmp + D;
    sx = (A < 0.0f || sx2 > 1.0f / tan_t) ? sx1 : sx2;
    float S;
    if (u2 > 0.5f) { S = 1.0f; u2 = 2.0f * (u2 - 0.5f); }
    else { S = -1.0f; u2 = 2.0f * (0.5f - u2); }
    float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) / (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    sy = S * z * sqrtf(1.0f + sx * sx);
}
PT_DEVO V3 tr_sample_wh(float ax, float ay, V3 wo, P2 u) {  // microfacet.rs:293-316,394-401
    bool flip = wo.z < 0.0f;
    V3 wi = flip ? -wo : wo;
    V3 wis = normalize(V3(ax * wi.x, ay * wi.y, wi.z));
    float sx, sy;
    tr_sample11(cos_theta(wis), u.x, u.y, sx, sy);
    float tmp = cos_phi(wis) * sx - sin_phi(wis) * sy;
    sy = sin_phi(wis) * sx + cos_phi(wis) * sy;
    sx = tmp;
    sx = ax * sx; sy = ay * sy;
    V3 wh = normalize(V3(-sx, -sy, 1.0f));
    return flip ? -wh : wh;
}

PT_DEV float pow5(float v) { return (v * v) * (v * v) * v; }

// DIFF = the material class holds diffuse lobes only (class 0, matte): the value range of `kind` is narrowed so that the
// compiler drops every specular / microfacet case from the matte shade kernel (smaller code, fewer registers).
template <bool FULL = true> PT_DEV float lobe_g(const Lobe &b, V3 wo, V3 wi) { return (FULL && b.sepg) ? tr_g1(b.ax, b.ay, wo) * tr_g1(b.ax, b.ay, wi) : tr_g(b.ax, b.ay, wo, wi); }

// FULL = the five-lobe class, the only one that can hold the Disney lobes (uber, subsurface, translucent, mix, disney materials): the
// one- and two-lobe kernels narrow `kind` to the classic lobes so that they carry none of the Disney code.
// DIFF == 2: the class holds perfectly specular lobes only (class 6: mirror, smooth glass).
template <int DIFF, bool FULL> PT_DEV uint8_t lobe_kind(const Lobe &b) {
    if (DIFF == 1) return b.kind == LB_OREN_NAYAR ? (uint8_t)LB_OREN_NAYAR : (uint8_t)LB_LAMBERT_R;
    if (DIFF == 2) return b.kind == LB_FRESNEL_SPEC ? (uint8_t)LB_FRESNEL_SPEC : (uint8_t)LB_SPEC_R;
    if (DIFF == 3) return (uint8_t)LB_MICRO_R;   // class 1 of a scene whose one-lobe materials are all metals: the conductor microfacet lobe only
    if (DIFF == 4) return b.kind == LB_LAMBERT_R ? (uint8_t)LB_LAMBERT_R : (uint8_t)LB_MICRO_R;
    if (DIFF == 6) return (uint8_t)LB_FRESNEL_SPEC;   // class 3 of a scene whose many-lobe materials are all subsurface materials with the smooth dielectric BSDF
    if (DIFF == 5) return b.kind == LB_LAMBERT_R ? (uint8_t)LB_LAMBERT_R : b.kind == LB_MICRO_R ? (uint8_t)LB_MICRO_R : b.kind == LB_SPEC_R ? (uint8_t)LB_SPEC_R : (uint8_t)LB_SPEC_T;   // class 3 of a scene whose many-lobe materials are all ubers   // class 2 of a scene without rough glass: plastic / opaque uber = Lambert + dielectric microfacet reflection
    return FULL ? b.kind : (b.kind > LB_FRESNEL_BLEND ? (uint8_t)LB_FRESNEL_BLEND : b.kind);
}

template <int DIFF = 0, bool FULL = true> PT_DEV RGB lobe_f(const Lobe &b, V3 wo, V3 wi) {
    switch (lobe_kind<DIFF, FULL>(b)) {
    case LB_LAMBERT_R: return b.r * kInvPi;
    case LB_LAMBERT_T: return b.t * kInvPi;
    case LB_OREN_NAYAR: {  // reflection.rs:926-952
        float sin_i = sin_theta(wi), sin_o = sin_theta(wo);
        float max_cos = 0.0f;
        if (sin_i > 1e-4f && sin_o > 1e-4f) {
            float dcos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
            max_cos = maxf(dcos, 0.0f);
        }
        float sin_alpha, tan_beta;
        if (abs_cos_theta(wi) > abs_cos_theta(wo)) { sin_alpha = sin_o; tan_beta = sin_i / abs_cos_theta(wi); }
        else { sin_alpha = sin_i; tan_beta = sin_o / abs_cos_theta(wo); }
        return b.r * kInvPi * (b.A + b.B * max_cos * sin_alpha * tan_beta);
    }
    case LB_SPEC_R: case LB_SPEC_T: return RGB(0.0f);
    case LB_FRESNEL_SPEC: return RGB(1.0f);
    case LB_MICRO_R: {  // reflection.rs:981-1003
        float cos_o = abs_cos_theta(wo), cos_i = abs_cos_theta(wi);
        V3 wh = wi + wo;
        if (cos_i == 0.0f || cos_o == 0.0f) return RGB(0.0f);
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return RGB(0.0f);
        wh = normalize(wh);
        RGB F = fresnel_eval<FULL>(b, dot(wi, wh), b.etaa, b.etab);
        float d = tr_d(b.ax, b.ay, wh), g = lobe_g<FULL>(b, wo, wi);
        return b.r * d * g * F / (4.0f * cos_i * cos_o);
    }
    case LB_MICRO_T: {  // reflection.rs:1059-1092
        if (same_hemisphere(wo, wi)) return RGB(0.0f);
        float cos_o = cos_theta(wo), cos_i = cos_theta(wi);
        if (cos_i == 0.0f || cos_o == 0.0f) return RGB(0.0f);
        float eta = (cos_theta(wo) > 0.0f) ? b.etab / b.etaa : b.etaa / b.etab;
        V3 wh = normalize(wo + wi * eta);
        if (wh.z < 0.0f) wh = -wh;
        if (dot(wo, wh) * dot(wi, wh) > 0.0f) return RGB(0.0f);
        RGB f = RGB(fr_dielectric(dot(wo, wh), b.etaa, b.etab));
        float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        float factor = 1.0f / eta;
        float s = tr_d(b.ax, b.ay, wh) * lobe_g<FULL>(b, wo, wi) * eta * eta * abs_dot(wi, wh) * abs_dot(wo, wh) * factor * factor /
                  (cos_i * cos_o * sqrt_denom * sqrt_denom);
        return (RGB(1.0f) - f) * b.t * fabsf(s);
    }
    case LB_DISNEY_DIFFUSE: {  // disney.rs:66-74
        const float fo = schlick_weight(abs_cos_theta(wo)), fi = schlick_weight(abs_cos_theta(wi));
        return b.r * kInvPi * (1.0f - fo / 2.0f) * (1.0f - fi / 2.0f);
    }
    case LB_DISNEY_FAKESS: case LB_DISNEY_RETRO: case LB_DISNEY_SHEEN: {  // disney.rs:104-124, :158-172, :202-210
        V3 wh = wi + wo;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return RGB(0.0f);
        wh = normalize(wh);
        const float cos_d = dot(wi, wh);
        if (b.kind == LB_DISNEY_SHEEN) return b.r * schlick_weight(cos_d);
        const float fo = schlick_weight(abs_cos_theta(wo)), fi = schlick_weight(abs_cos_theta(wi));
        if (b.kind == LB_DISNEY_RETRO) {
            const float rr = 2.0f * b.A * cos_d * cos_d;
            return b.r * kInvPi * rr * (fo + fi + fo * fi * (rr - 1.0f));
        }
        const float fss90 = cos_d * cos_d * b.A;
        const float fss = lerp_t(fo, 1.0f, fss90) * lerp_t(fi, 1.0f, fss90);
        const float ss = 1.25f * (fss * (1.0f / (abs_cos_theta(wo) + abs_cos_theta(wi)) - 0.5f) + 0.5f);
        return b.r / kInvPi * ss;   // divides by INV_PI, as the reference does (disney.rs:123)
    }
    case LB_DISNEY_CLEARCOAT: {  // disney.rs:238-255
        V3 wh = wi + wo;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return RGB(0.0f);
        wh = normalize(wh);
        const float dr = gtr1(abs_cos_theta(wh), b.A);
        const float fr = fr_schlick(0.04f, dot(wo, wh));
        const float gr = smithg_ggx(abs_cos_theta(wo), 0.25f) * smithg_ggx(abs_cos_theta(wi), 0.25f);
        return RGB(fr * b.B * gr * dr / 4.0f);
    }
    default: {  // LB_FRESNEL_BLEND, reflection.rs:1165-1182 (r = Rd, t = Rs)
        RGB diffuse = b.r * (RGB(1.0f) - b.t) * (28.0f / (23.0f * kPi)) * (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wi))) *
                      (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wo)));
        V3 wh = wi + wo;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return RGB(0.0f);
        wh = normalize(wh);
        RGB schlick = b.t + (RGB(1.0f) - b.t) * pow5(1.0f - dot(wi, wh));
        RGB specular = schlick * (tr_d(b.ax, b.ay, wh) / (4.0f * abs_dot(wi, wh) * maxf(abs_cos_theta(wi), abs_cos_theta(wo))));
        return diffuse + specular;
    }
    }
}

template <int DIFF = 0, bool FULL = true> PT_DEV float lobe_pdf(const Lobe &b, V3 wo, V3 wi) {
    switch (lobe_kind<DIFF, FULL>(b)) {
    case LB_LAMBERT_R: case LB_OREN_NAYAR: case LB_FRESNEL_SPEC:  // reflection.rs:439-445, :788-794
    case LB_DISNEY_DIFFUSE: case LB_DISNEY_FAKESS: case LB_DISNEY_RETRO: case LB_DISNEY_SHEEN:   // BxDF::pdf default
        return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * kInvPi : 0.0f;
    case LB_DISNEY_CLEARCOAT: {  // disney.rs:276-292: the half vector is built from `wi + wi`, as written there
        if (!same_hemisphere(wo, wi)) return 0.0f;
        V3 wh = wi + wi;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return 0.0f;
        wh = normalize(wh);
        const float dr = gtr1(abs_cos_theta(wh), b.A);
        return dr * abs_cos_theta(wh) / (4.0f * dot(wo, wh));
    }
    case LB_LAMBERT_T: return !same_hemisphere(wo, wi) ? abs_cos_theta(wi) : 0.0f;  // :886-892
    case LB_SPEC_R: case LB_SPEC_T: return 0.0f;
    case LB_MICRO_R: {
        if (!same_hemisphere(wo, wi)) return 0.0f;
        V3 wh = normalize(wo + wi);
        return tr_pdf(b.ax, b.ay, wo, wh) / (4.0f * dot(wo, wh));
    }
    case LB_MICRO_T: {
        if (same_hemisphere(wo, wi)) return 0.0f;
        float eta = (cos_theta(wo) > 0.0f) ? b.etaa / b.etab : b.etab / b.etaa;
        V3 wh = normalize(wo + wi * eta);
        if (dot(wo, wh) * dot(wi, wh) > 0.0f) return 0.0f;
        float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        float dwh_dwi = fabsf(eta * eta * dot(wi, wh)) / (sqrt_denom * sqrt_denom);
        return tr_pdf(b.ax, b.ay, wo, wh) * dwh_dwi;
    }
    default: {  // LB_FRESNEL_BLEND
        if (!same_hemisphere(wo, wi)) return 0.0f;
        V3 wh = normalize(wo + wi);
        float pdf_wh = tr_pdf(b.ax, b.ay, wo, wh);
        return 0.5f * (abs_cos_theta(wi) * kInvPi + pdf_wh / (4.0f * dot(wo, wh)));
    }
    }
}

template <int DIFF = 0, bool FULL = true> PT_DEV RGB lobe_sample_f(const Lobe &b, V3 wo, V3 &wi, P2 u, float &pdf, int &sampled) {
    switch (lobe_kind<DIFF, FULL>(b)) {
    case LB_DISNEY_CLEARCOAT: {  // disney.rs:257-274
        if (wo.z == 0.0f) return RGB(0.0f);
        const float a2 = b.A * b.A;
        const float ct = sqrtf(maxf((1.0f - dm_powf(a2, 1.0f - u.x)) / (1.0f - a2), 0.0f));
        const float st = sqrtf(maxf(1.0f - ct * ct, 0.0f));
        const float phi = 2.0f * kPi * u.y;
        float sp, cp; dm_sincosf(phi, sp, cp);
        V3 wh(st * cp, st * sp, ct);
        if (!same_hemisphere(wo, wh)) wh = -wh;
        wi = reflect(wo, wh);
        if (!same_hemisphere(wo, wh)) return RGB(0.0f);
        pdf = lobe_pdf<DIFF, FULL>(b, wo, wi);
        return lobe_f<DIFF, FULL>(b, wo, wi);
    }
    case LB_LAMBERT_R: case LB_OREN_NAYAR: case LB_DISNEY_DIFFUSE: case LB_DISNEY_FAKESS: case LB_DISNEY_RETRO: case LB_DISNEY_SHEEN: {   // BxDF::sample_f default
        wi = cosine_sample_hemisphere(u);
        if (wo.z < 0.0f) wi.z *= -1.0f;
        pdf = lobe_pdf<DIFF, FULL>(b, wo, wi);
        return lobe_f<DIFF, FULL>(b, wo, wi);
    }
    case LB_LAMBERT_T: {
        wi = cosine_sample_hemisphere(u);
        if (wo.z > 0.0f) wi.z *= -1.0f;
        pdf = lobe_pdf<DIFF, FULL>(b, wo, wi);
        return lobe_f<DIFF, FULL>(b, wo, wi);
    }
    case LB_SPEC_R: {
        wi = V3(-wo.x, -wo.y, wo.z);
        pdf = 1.0f;
        return fresnel_eval<FULL>(b, cos_theta(wi), b.etaa, b.etab) * b.r / abs_cos_theta(wi);
    }
    case LB_SPEC_T: {
        float etai, etat;
        if (cos_theta(wo) > 0.0f) { etai = b.etaa; etat = b.etab; } else { etai = b.etab; etat = b.etaa; }
        if (!refract(wo, face_forward(V3(0.0f, 0.0f, 1.0f), wo), etai / etat, wi)) return RGB(0.0f);
        pdf = 1.0f;
        RGB ft = b.t * (RGB(1.0f) - RGB(fr_dielectric(cos_theta(wi), b.etaa, b.etab)));
        ft = ft * ((etai * etai) / (etat * etat));
        return ft / abs_cos_theta(wi);
    }
    case LB_FRESNEL_SPEC: {
        float f = fr_dielectric(cos_theta(wo), b.etaa, b.etab);
        if (u.x < f) {
            wi = V3(-wo.x, -wo.y, wo.z);
            sampled = BSDF_SPECULAR | BSDF_REFLECTION;
            pdf = f;
            return b.r / abs_cos_theta(wi) * f;
        }
        float etai, etat;
        if (cos_theta(wo) > 0.0f) { etai = b.etaa; etat = b.etab; } else { etai = b.etab; etat = b.etaa; }
        if (!refract(wo, face_forward(V3(0.0f, 0.0f, 1.0f), wo), etai / etat, wi)) return RGB(0.0f);
        RGB ft = b.t * (1.0f - f);
        ft = ft * ((etai * etai) / (etat * etat));
        sampled = BSDF_SPECULAR | BSDF_TRANSMISSION;
        pdf = 1.0f - f;
        return ft / abs_cos_theta(wi);
    }
    case LB_MICRO_R: {
        if (wo.z == 0.0f) return RGB(0.0f);
        V3 wh = tr_sample_wh(b.ax, b.ay, wo, u);
        if (dot(wo, wh) < 0.0f) return RGB(0.0f);
        wi = reflect(wo, wh);
        if (!same_hemisphere(wo, wi)) return RGB(0.0f);
        pdf = tr_pdf(b.ax, b.ay, wo, wh) / (4.0f * dot(wo, wh));
        return lobe_f<DIFF, FULL>(b, wo, wi);
    }
    case LB_MICRO_T: {
        if (wo.z == 0.0f) return RGB(0.0f);
        V3 wh = tr_sample_wh(b.ax, b.ay, wo, u);
        if (dot(wo, wh) < 0.0f) return RGB(0.0f);
        float eta = (cos_theta(wo) > 0.0f) ? b.etaa / b.etab : b.etab / b.etaa;
        if (!refract(wo, wh, eta, wi)) return RGB(0.0f);
        pdf = lobe_pdf<DIFF, FULL>(b, wo, wi);
        return lobe_f<DIFF, FULL>(b, wo, wi);
    }
    default: {  // LB_FRESNEL_BLEND
        P2 uu = u;
        if (uu.x < 0.5f) {
            uu.x = minf(2.0f * uu.x, kOneMinusEps);
            wi = cosine_sample_hemisphere(uu);
            if (wo.z < 0.0f) wi.z *= -1.0f;
        } else {
            uu.x = minf(2.0f * (uu.x - 0.5f), kOneMinusEps);
            V3 wh = tr_sample_wh(b.ax, b.ay, wo, uu);
            wi = reflect(wo, wh);
            if (!same_hemisphere(wo, wi)) return RGB(0.0f);
        }
        pdf = lobe_pdf<DIFF, FULL>(b, wo, wi);
        return lobe_f<DIFF, FULL>(b, wo, wi);
    }
    }
}

// ---- BSDF (reflection.rs:1495-1689). MAXL = compile-time lobe capacity of the shade-queue class -----------
// Lobe storage. A one-lobe BSDF keeps its lobe in registers. The two- and five-lobe classes keep theirs in LDS (12 words per lobe,
// word w of lobe i of thread t at store[(i * 12 + w) * 256 + t]: conflict free) and walk them with rolled loops that hold ONE lobe
// in registers at a time: with the lobes in a register array every loop over them was unrolled MAXL times around the big
// per-kind switches, which cost the five-lobe kernel 256 VGPRs + 73 AGPRs + 632 bytes of scratch per lane at one wave per SIMD.
// The 19 fields of a Lobe are never all live in one lobe, so the stored form overlays them (5 x 19 words x 256 threads = 95 KB kept the
// five-lobe kernel at ONE workgroup per CU; 5 x 12 = 60 KB lets two share the 160 KB):
//   word 0      kind | type << 8 | fresnel << 16 | sepg << 24
//   words 1-3   r
//   words 4-6   ce for the conductor / Disney Fresnel terms (lobes that have no t), else t
//   words 7-9   ck for the conductor Fresnel term, else {etaa, etab, A}
//   word 10     B for Oren-Nayar and the Disney clearcoat (lobes without a microfacet distribution), else ax
//   word 11     ay
// Fields a lobe does not store come back as mk_lobe() leaves them; no lobe kind reads a field it does not store.
constexpr int kLobeWords = 12, kLobeStride = 256;   // (k_shade runs 256-thread blocks)
PT_DEV bool lobe_stores_ce(uint8_t fresnel) { return fresnel == FR_CONDUCTOR || fresnel == FR_DISNEY; }
PT_DEV bool lobe_stores_b(uint8_t kind) { return kind == LB_OREN_NAYAR || kind == LB_DISNEY_CLEARCOAT; }
template <int MAXL> constexpr int lobe_store_words() { return MAXL > 1 ? MAXL * kLobeWords * kLobeStride : 1; }

template <int MAXL, int DIFF = 0> struct Bsdf {
    float eta;
    V3 ns, ng, ss, ts;
    int n;
    static constexpr bool LDS = MAXL > 1;
    Lobe l1;               // MAXL == 1
    float *store;          // MAXL > 1: this thread's column of the block's LDS lobe store
    uint64_t types;        // BxDFType byte of lobe i in bits 8i .. 8i+7 (what `matches` and the reflect / transmit split read)
    // MixMaterial (mix.rs:25-50): lobes [0, n1) are ScaledBxDFs with scale s1, lobes [n1, n) with scale s2 (reflection.rs:466-517);
    // only the five-lobe class carries this. `frozen`: the second material's init() must not reset the frame / eta / lobes.
    static constexpr bool MIX = MAXL == 5 && DIFF != 5, FULL = MAXL == 5 && DIFF != 5;   // (DIFF 5: the uber-only form of the five-lobe class carries no mix / Disney code)
    int n1; RGB s1, s2; bool frozen;

    PT_DEV void bind(float *block_store) { store = LDS ? block_store + threadIdx.x : nullptr; }
    PT_DEV void init(const SurfaceInteraction &si, float eta_) {
        if (MIX && frozen) return;
        eta = eta_; ns = si.sh_n; ss = normalize(si.sh_dpdu); ng = si.n; ts = cross(ns, ss); n = 0; types = 0ull;
    }
    PT_DEV RGB scaled(int i, RGB v) const { if (MIX && n1 >= 0) return (i < n1 ? s1 : s2) * v; return v; }
    PT_DEV int type_of(int i) const { return (int)((types >> (8 * i)) & 0xffull); }
    PT_DEV bool matches(int i, int flags) const { const int t = type_of(i); return (t & flags) == t; }
    PT_DEV Lobe get(int i) const {
        if (!LDS) return l1;
        const float *p = store + (size_t)i * (kLobeWords * kLobeStride);
        Lobe b;
        const uint32_t w0 = __float_as_uint(p[0]);
        b.kind = (uint8_t)(w0 & 0xffu); b.type = (uint8_t)((w0 >> 8) & 0xffu); b.fresnel = (uint8_t)((w0 >> 16) & 0xffu); b.sepg = (uint8_t)(w0 >> 24);
        if (DIFF == 4 || DIFF == 5) { b.fresnel = (uint8_t)FR_DIELECTRIC; b.sepg = 0; }   // (the only Fresnel term a plastic-like lobe set evaluates; the Lambert lobe reads none)
        b.r = RGB(p[1 * kLobeStride], p[2 * kLobeStride], p[3 * kLobeStride]);
        const RGB x4(p[4 * kLobeStride], p[5 * kLobeStride], p[6 * kLobeStride]);
        const float y7 = p[7 * kLobeStride], y8 = p[8 * kLobeStride], y9 = p[9 * kLobeStride], z10 = p[10 * kLobeStride];
        b.t = RGB(0.0f); b.ce = RGB(0.0f); b.ck = RGB(0.0f); b.etaa = b.etab = 1.0f; b.A = b.B = 0.0f; b.ax = 0.001f;
        if (lobe_stores_ce(b.fresnel)) b.ce = x4; else b.t = x4;
        if (b.fresnel == FR_CONDUCTOR) b.ck = RGB(y7, y8, y9); else { b.etaa = y7; b.etab = y8; b.A = y9; }
        if (lobe_stores_b(b.kind)) b.B = z10; else b.ax = z10;
        b.ay = p[11 * kLobeStride];
        return b;
    }
    PT_DEV void add(const Lobe &x) {
        if (n >= MAXL) return;
        if (!LDS) l1 = x;
        else {
            float *p = store + (size_t)n * (kLobeWords * kLobeStride);
            p[0] = __uint_as_float((uint32_t)x.kind | ((uint32_t)x.type << 8) | ((uint32_t)x.fresnel << 16) | ((uint32_t)x.sepg << 24));
            p[1 * kLobeStride] = x.r.r; p[2 * kLobeStride] = x.r.g; p[3 * kLobeStride] = x.r.b;
            const RGB x4 = lobe_stores_ce(x.fresnel) ? x.ce : x.t;
            p[4 * kLobeStride] = x4.r; p[5 * kLobeStride] = x4.g; p[6 * kLobeStride] = x4.b;
            const bool cond = x.fresnel == FR_CONDUCTOR;
            p[7 * kLobeStride] = cond ? x.ck.r : x.etaa; p[8 * kLobeStride] = cond ? x.ck.g : x.etab; p[9 * kLobeStride] = cond ? x.ck.b : x.A;
            p[10 * kLobeStride] = lobe_stores_b(x.kind) ? x.B : x.ax; p[11 * kLobeStride] = x.ay;
        }
        types |= (uint64_t)x.type << (8 * n);
        n++;
    }
    PT_DEV int num_components(int flags) const { int c = 0; for (int i = 0; i < MAXL; ++i) if (i < n && matches(i, flags)) ++c; return c; }
    PT_DEV V3 to_local(V3 v) const { return V3(dot(v, ss), dot(v, ts), dot(v, ns)); }
    PT_DEV V3 to_world(V3 v) const {
        return V3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
    PT_DEV RGB f(V3 wow, V3 wiw, int flags) const {
        V3 wi = to_local(wiw), wo = to_local(wow);
        if (wo.z == 0.0f) return RGB(0.0f);
        bool refl = dot(wiw, ng) * dot(wow, ng) > 0.0f;
        RGB res(0.0f);
#pragma unroll 1
        for (int i = 0; i < n; ++i)
            if (matches(i, flags) && ((refl && (type_of(i) & BSDF_REFLECTION)) || (!refl && (type_of(i) & BSDF_TRANSMISSION))))
                res = res + scaled(i, lobe_f<DIFF, FULL>(get(i), wo, wi));
        return res;
    }
    PT_DEV float pdf(V3 wow, V3 wiw, int flags) const {
        if (n == 0) return 0.0f;
        V3 wo = to_local(wow), wi = to_local(wiw);
        if (wo.z == 0.0f) return 0.0f;
        float p = 0.0f; int matching = 0;
#pragma unroll 1
        for (int i = 0; i < n; ++i) if (matches(i, flags)) { ++matching; p += lobe_pdf<DIFF, FULL>(get(i), wo, wi); }
        return matching > 0 ? p / (float)matching : 0.0f;
    }
    // f(wow, wiw, flags) and pdf(wow, wiw, flags) of the same pair of directions (estimate_direct's light sample, integrator.rs:142-147) in ONE
    // pass over the lobes: each lobe is fetched from the LDS store and decoded once, and what its value and its density share (the half
    // vector, the microfacet distribution's D) is computed once -- the two functions are inlined side by side on the same operands.
#ifndef PT_FUSE_FPDF
#define PT_FUSE_FPDF 1   // experiment hooks: 0 = the two passes as they were
#endif
#ifndef PT_FUSE_TAIL
#define PT_FUSE_TAIL 1
#endif
    PT_DEV RGB f_pdf(V3 wow, V3 wiw, int flags, float &pdf_out) const {
        // (one lobe in registers: nothing to share; the five-lobe class, already spilling, loses more to the longer live ranges than it gains:
        //  C3 two-lobe kernel 130.6 -> 124.7 ms with both fusions, five-lobe kernel 157.8 -> 150.4 ms with only sample_f's)
        if (MAXL != 2 || !PT_FUSE_FPDF) { const RGB r = f(wow, wiw, flags); pdf_out = pdf(wow, wiw, flags); return r; }
        pdf_out = 0.0f;
        V3 wi = to_local(wiw), wo = to_local(wow);
        if (wo.z == 0.0f) return RGB(0.0f);
        bool refl = dot(wiw, ng) * dot(wow, ng) > 0.0f;
        RGB res(0.0f);
        float p = 0.0f; int matching = 0;
#pragma unroll 1
        for (int i = 0; i < n; ++i) {
            if (!matches(i, flags)) continue;
            const Lobe b = get(i);
            ++matching; p += lobe_pdf<DIFF, FULL>(b, wo, wi);
            if ((refl && (type_of(i) & BSDF_REFLECTION)) || (!refl && (type_of(i) & BSDF_TRANSMISSION))) res = res + scaled(i, lobe_f<DIFF, FULL>(b, wo, wi));
        }
        pdf_out = matching > 0 ? p / (float)matching : 0.0f;
        return res;
    }
    // `pdf` must hold the caller's previous value on entry (it is left untouched on the wo.z == 0 exit,
    // reflection.rs:1603-1604).
    PT_DEV RGB sample_f(V3 wow, V3 &wiw, P2 u, float &pdf, int ty, int &sampled) const {
        int matching = num_components(ty);
        if (matching == 0) { pdf = 0.0f; sampled = 0; return RGB(0.0f); }
        int comp = (int)min(f2u32_sat(floorf(u.x * (float)matching)), (uint32_t)(matching - 1));
        int idx = 0, count = comp;
        for (int i = 0; i < MAXL; ++i) {
            if (i >= n) break;
            bool m = matches(i, ty);
            if (m && count == 0) { idx = i; break; }
            else if (m) --count;
        }
        P2 ur(minf(u.x * (float)matching - (float)comp, kOneMinusEps), u.y);
        V3 wo = to_local(wow), wi;
        if (wo.z == 0.0f) return RGB(0.0f);
        pdf = 0.0f;
        RGB fv(0.0f);
        const int btype = type_of(idx);
        sampled = btype;
        fv = scaled(idx, lobe_sample_f<DIFF, FULL>(get(idx), wo, wi, ur, pdf, sampled));
        if (pdf == 0.0f) { sampled = 0; return RGB(0.0f); }
        wiw = to_world(wi);
        if (!LDS || !PT_FUSE_TAIL) {
            if (!(btype & BSDF_SPECULAR) && matching > 1) {
#pragma unroll 1
                for (int i = 0; i < n; ++i) if (i != idx && matches(i, ty)) pdf += lobe_pdf<DIFF, FULL>(get(i), wo, wi);
            }
            if (matching > 1) pdf /= (float)matching;
            if (!(btype & BSDF_SPECULAR)) {
                bool refl = dot(wiw, ng) * dot(wow, ng) > 0.0f;
                fv = RGB(0.0f);
#pragma unroll 1
                for (int i = 0; i < n; ++i)
                    if (matches(i, ty) && ((refl && (type_of(i) & BSDF_REFLECTION)) || (!refl && (type_of(i) & BSDF_TRANSMISSION))))
                        fv = fv + scaled(i, lobe_f<DIFF, FULL>(get(i), wo, wi));
            }
            return fv;
        }
        if (!(btype & BSDF_SPECULAR)) {   // the other lobes' densities (reflection.rs:1627-1633) and every matching lobe's value (:1639-1650), one pass
            const bool refl = dot(wiw, ng) * dot(wow, ng) > 0.0f;
            fv = RGB(0.0f);
#pragma unroll 1
            for (int i = 0; i < n; ++i) {
                if (!matches(i, ty)) continue;
                const Lobe b = get(i);
                if (i != idx && matching > 1) pdf += lobe_pdf<DIFF, FULL>(b, wo, wi);
                if ((refl && (type_of(i) & BSDF_REFLECTION)) || (!refl && (type_of(i) & BSDF_TRANSMISSION))) fv = fv + scaled(i, lobe_f<DIFF, FULL>(b, wo, wi));
            }
        }
        if (matching > 1) pdf /= (float)matching;
        return fv;
    }
};

PT_DEV RGB rgb3(const float *p) { return RGB(p[0], p[1], p[2]); }
// DisneyMaterial attaches a BSSRDF when it is not thin, has diffuse weight and a non-black scatter distance (disney.rs:755-776)
PT_HD bool disney_has_bssrdf(const PtMaterial &m) {
    return m.type == PT_MAT_DISNEY && !m.disney_thin && (1.0f - m.disney[PT_DS_METALLIC]) * (1.0f - m.disney[PT_DS_SPECTRANS]) > 0.0f &&
           (m.disney_scatter[0] != 0.0f || m.disney_scatter[1] != 0.0f || m.disney_scatter[2] != 0.0f);
}
PT_DEV Lobe mk_lobe(uint8_t kind, uint8_t type) { Lobe b; b.kind = kind; b.type = type; b.fresnel = FR_NOOP; b.sepg = 0; b.ax = b.ay = 0.001f; b.etaa = b.etab = 1.0f; b.A = b.B = 0.0f; return b; }
PT_DEV void set_dist(Lobe &b, float ax, float ay) { b.ax = maxf(ax, 0.001f); b.ay = maxf(ay, 0.001f); }  // microfacet.rs:325-331

// Texture::evaluate of a material parameter when no texture can be bound (scenes without textures): the constant field.
struct ConstMatEval {
    PT_DEV RGB spec(const PtMaterial &, int, const float *field) const { return RGB(field[0], field[1], field[2]); }
    PT_DEV float flt(const PtMaterial &, int, float field) const { return field; }
    PT_DEV bool bound(const PtMaterial &, int) const { return false; }
    PT_DEV ConstMatEval plain() const { return *this; }
};

// Material::compute_scattering_functions. Returns false when the reference leaves si.bsdf == None. `E` evaluates the
// (possibly textured) parameters: ConstMatEval above, or the texture evaluator of kernels.hip.
template <int MAXL, class ME, int DIFF> PT_DEV bool build_bsdf_leaf(const PtMaterial &m, const SurfaceInteraction &si, Bsdf<MAXL, DIFF> &bsdf, const ME &E) {
    // class 0 holds matte materials only, class 6 mirrors and smooth glass: the other cases drop out of those kernels
    switch (DIFF == 1 ? (uint32_t)PT_MAT_MATTE : DIFF == 2 ? (m.type == PT_MAT_MIRROR ? (uint32_t)PT_MAT_MIRROR : (uint32_t)PT_MAT_GLASS) : DIFF == 3 ? (uint32_t)PT_MAT_METAL : DIFF == 4 ? (m.type == PT_MAT_PLASTIC ? (uint32_t)PT_MAT_PLASTIC : (uint32_t)PT_MAT_UBER) : DIFF == 5 ? (uint32_t)PT_MAT_UBER : DIFF == 6 ? (uint32_t)PT_MAT_SUBSURFACE : m.type) {
    case PT_MAT_MATTE: {  // matte.rs:28-53
        bsdf.init(si, 1.0f);
        RGB r = E.spec(m, PT_MP_KD, m.kd).clamps(0.0f, PT_INF);
        float sig = clampf(E.flt(m, PT_MP_SIGMA, m.sigma), 0.0f, 90.0f);
        if (!r.is_black()) {
            Lobe b = mk_lobe(sig == 0.0f ? LB_LAMBERT_R : LB_OREN_NAYAR, BSDF_REFLECTION | BSDF_DIFFUSE);
            b.r = r;
            if (sig != 0.0f) {  // reflection.rs:909-921
                float sigma = (kPi / 180.0f) * sig;
                float sigma2 = sigma * sigma;
                b.A = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
                b.B = 0.45f * sigma2 / (sigma2 + 0.09f);
            }
            bsdf.add(b);
        }
        return true;
    }
    case PT_MAT_MIRROR: {  // mirror.rs:23-42
        bsdf.init(si, 1.0f);
        RGB R = E.spec(m, PT_MP_KR, m.kr).clamps(0.0f, PT_INF);
        if (!R.is_black()) { Lobe b = mk_lobe(LB_SPEC_R, BSDF_REFLECTION | BSDF_SPECULAR); b.r = R; bsdf.add(b); }
        return true;
    }
    case PT_MAT_SUBSURFACE:  // subsurface.rs:56-98 / kdsubsurface.rs:53-93: the same dielectric BSDF as glass
    case PT_MAT_GLASS: {  // glass.rs:35-92
        float eta = E.flt(m, PT_MP_ETA, m.eta), ur = E.flt(m, PT_MP_U_ROUGHNESS, m.u_roughness), vr = E.flt(m, PT_MP_V_ROUGHNESS, m.v_roughness);
        RGB R = E.spec(m, PT_MP_KR, m.kr).clamps(0.0f, PT_INF), T = E.spec(m, PT_MP_KT, m.kt).clamps(0.0f, PT_INF);
        bsdf.init(si, eta);
        if (R.is_black() && T.is_black()) return false;
        bool is_specular = ur == 0.0f && vr == 0.0f;
        if (is_specular) {
            Lobe b = mk_lobe(LB_FRESNEL_SPEC, BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR);
            b.r = R; b.t = T; b.etaa = 1.0f; b.etab = eta; bsdf.add(b);
        } else {
            if (m.remap_roughness) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
            if (!R.is_black()) {
                Lobe b = mk_lobe(LB_MICRO_R, BSDF_REFLECTION | BSDF_GLOSSY); b.r = R; set_dist(b, ur, vr);
                b.fresnel = FR_DIELECTRIC; b.etaa = 1.0f; b.etab = eta; bsdf.add(b);
            }
            if (!T.is_black()) {
                Lobe b = mk_lobe(LB_MICRO_T, BSDF_TRANSMISSION | BSDF_GLOSSY); b.t = T; set_dist(b, ur, vr);
                b.fresnel = FR_DIELECTRIC; b.etaa = 1.0f; b.etab = eta; bsdf.add(b);
            }
        }
        return true;
    }
    case PT_MAT_PLASTIC: {  // plastic.rs:34-70
        bsdf.init(si, 1.0f);
        RGB kd = E.spec(m, PT_MP_KD, m.kd).clamps(0.0f, PT_INF);
        if (!kd.is_black()) { Lobe b = mk_lobe(LB_LAMBERT_R, BSDF_REFLECTION | BSDF_DIFFUSE); b.r = kd; bsdf.add(b); }
        RGB ks = E.spec(m, PT_MP_KS, m.ks).clamps(0.0f, PT_INF);
        if (!ks.is_black()) {
            float rough = E.flt(m, PT_MP_ROUGHNESS, m.roughness);
            if (m.remap_roughness) rough = roughness_to_alpha(rough);
            Lobe b = mk_lobe(LB_MICRO_R, BSDF_REFLECTION | BSDF_GLOSSY); b.r = ks; set_dist(b, rough, rough);
            b.fresnel = FR_DIELECTRIC; b.etaa = 1.5f; b.etab = 1.0f; bsdf.add(b);
        }
        return true;
    }
    case PT_MAT_METAL: {  // metal.rs:78-112
        bsdf.init(si, 1.0f);
        // metal.rs:88-96: the uroughness / vroughness textures fall back to the `roughness` texture when absent
        float ur = (E.bound(m, PT_MP_U_ROUGHNESS) || m.u_roughness >= 0.0f) ? E.flt(m, PT_MP_U_ROUGHNESS, m.u_roughness) : E.flt(m, PT_MP_ROUGHNESS, m.roughness);
        float vr = (E.bound(m, PT_MP_V_ROUGHNESS) || m.v_roughness >= 0.0f) ? E.flt(m, PT_MP_V_ROUGHNESS, m.v_roughness) : E.flt(m, PT_MP_ROUGHNESS, m.roughness);
        if (m.remap_roughness) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
        Lobe b = mk_lobe(LB_MICRO_R, BSDF_REFLECTION | BSDF_GLOSSY); b.r = RGB(1.0f); set_dist(b, ur, vr);
        b.fresnel = FR_CONDUCTOR; b.ce = E.spec(m, PT_MP_ETA_RGB, m.eta_rgb); b.ck = E.spec(m, PT_MP_K_RGB, m.k_rgb);
        bsdf.add(b);
        return true;
    }
    case PT_MAT_UBER: {  // uber.rs:40-106
        float e = E.flt(m, PT_MP_ETA, m.eta);
        RGB op = E.spec(m, PT_MP_OPACITY, m.opacity).clamps(0.0f, PT_INF);
        RGB t = RGB(-op.r + 1.0f, -op.g + 1.0f, -op.b + 1.0f).clamps(0.0f, PT_INF);
        if (!t.is_black()) {
            bsdf.init(si, 1.0f);
            Lobe b = mk_lobe(LB_SPEC_T, BSDF_TRANSMISSION | BSDF_SPECULAR); b.t = t; b.etaa = 1.0f; b.etab = 1.0f; bsdf.add(b);
        } else bsdf.init(si, e);
        RGB kd = op * E.spec(m, PT_MP_KD, m.kd).clamps(0.0f, PT_INF);
        if (!kd.is_black()) { Lobe b = mk_lobe(LB_LAMBERT_R, BSDF_REFLECTION | BSDF_DIFFUSE); b.r = kd; bsdf.add(b); }
        RGB ks = op * E.spec(m, PT_MP_KS, m.ks).clamps(0.0f, PT_INF);
        if (!ks.is_black()) {
            float ru = (E.bound(m, PT_MP_U_ROUGHNESS) || m.u_roughness >= 0.0f) ? E.flt(m, PT_MP_U_ROUGHNESS, m.u_roughness) : E.flt(m, PT_MP_ROUGHNESS, m.roughness);
            float rv = (E.bound(m, PT_MP_V_ROUGHNESS) || m.v_roughness >= 0.0f) ? E.flt(m, PT_MP_V_ROUGHNESS, m.v_roughness) : E.flt(m, PT_MP_ROUGHNESS, m.roughness);
            if (m.remap_roughness) { ru = roughness_to_alpha(ru); rv = roughness_to_alpha(rv); }
            Lobe b = mk_lobe(LB_MICRO_R, BSDF_REFLECTION | BSDF_GLOSSY); b.r = ks; set_dist(b, ru, rv);
            b.fresnel = FR_DIELECTRIC; b.etaa = 1.0f; b.etab = e; bsdf.add(b);
        }
        RGB kr = op * E.spec(m, PT_MP_KR, m.kr).clamps(0.0f, PT_INF);
        if (!kr.is_black()) {
            Lobe b = mk_lobe(LB_SPEC_R, BSDF_REFLECTION | BSDF_SPECULAR); b.r = kr; b.fresnel = FR_DIELECTRIC; b.etaa = 1.0f; b.etab = e; bsdf.add(b);
        }
        RGB kt = op * E.spec(m, PT_MP_KT, m.kt).clamps(0.0f, PT_INF);
        if (!kt.is_black()) { Lobe b = mk_lobe(LB_SPEC_T, BSDF_TRANSMISSION | BSDF_SPECULAR); b.t = kt; b.etaa = 1.0f; b.etab = e; bsdf.add(b); }
        return true;
    }
    case PT_MAT_TRANSLUCENT: {  // translucent.rs:36-78 (kr = "reflect", kt = "transmit")
        const float eta = 1.5f;
        bsdf.init(si, eta);
        RGB r = E.spec(m, PT_MP_KR, m.kr).clamps(0.0f, PT_INF), t = E.spec(m, PT_MP_KT, m.kt).clamps(0.0f, PT_INF);
        if (r.is_black() && t.is_black()) return false;
        RGB kd = E.spec(m, PT_MP_KD, m.kd).clamps(0.0f, PT_INF);
        if (!kd.is_black()) {
            if (!r.is_black()) { Lobe b = mk_lobe(LB_LAMBERT_R, BSDF_REFLECTION | BSDF_DIFFUSE); b.r = r * kd; bsdf.add(b); }
            if (!t.is_black()) { Lobe b = mk_lobe(LB_LAMBERT_T, BSDF_TRANSMISSION | BSDF_DIFFUSE); b.t = t * kd; bsdf.add(b); }
        }
        RGB ks = E.spec(m, PT_MP_KS, m.ks).clamps(0.0f, PT_INF);
        if (!ks.is_black() && (!r.is_black() || !t.is_black())) {
            float rough = E.flt(m, PT_MP_ROUGHNESS, m.roughness);
            if (m.remap_roughness) rough = roughness_to_alpha(rough);
            if (!r.is_black()) {
                Lobe b = mk_lobe(LB_MICRO_R, BSDF_REFLECTION | BSDF_GLOSSY); b.r = r * ks; set_dist(b, rough, rough);
                b.fresnel = FR_DIELECTRIC; b.etaa = 1.0f; b.etab = eta; bsdf.add(b);
            }
            if (!t.is_black()) {
                Lobe b = mk_lobe(LB_MICRO_T, BSDF_TRANSMISSION | BSDF_GLOSSY); b.t = t * ks; set_dist(b, rough, rough);
                b.fresnel = FR_DIELECTRIC; b.etaa = 1.0f; b.etab = eta; bsdf.add(b);
            }
        }
        return true;
    }
    case PT_MAT_DISNEY: {  // disney.rs:719-840 (scatterdistance == 0: the BSSRDF branch is refused at scene creation)
        bsdf.init(si, 1.0f);
        const RGB c = E.spec(m, PT_MP_KD, m.kd).clamps(0.0f, PT_INF);   // "color"
        const float metallic = m.disney[PT_DS_METALLIC], e = E.flt(m, PT_MP_ETA, m.eta), strans = m.disney[PT_DS_SPECTRANS];
        const float dweight = (1.0f - metallic) * (1.0f - strans);
        const float dt = m.disney[PT_DS_DIFFTRANS] / 2.0f;
        const float rough = E.flt(m, PT_MP_ROUGHNESS, m.roughness);
        const float lum = 0.212671f * c.r + 0.715160f * c.g + 0.072169f * c.b;   // Spectrum::y (spectrum.rs:123-127)
        const RGB ctint = lum > 0.0f ? c / lum : RGB(1.0f);
        const float sheen_weight = m.disney[PT_DS_SHEEN];
        RGB csheen(0.0f);
        if (sheen_weight > 0.0f) csheen = lerp_rgb(m.disney[PT_DS_SHEENTINT], RGB(1.0f), ctint);
        const bool thin = m.disney_thin != 0;
        if (dweight > 0.0f) {
            if (thin) {
                const float flat = m.disney[PT_DS_FLATNESS];
                { Lobe b = mk_lobe(LB_DISNEY_DIFFUSE, BSDF_REFLECTION | BSDF_DIFFUSE); b.r = c * dweight * flat * (1.0f - dt); bsdf.add(b); }   // `flat`, not 1 - flat (disney.rs:759)
                { Lobe b = mk_lobe(LB_DISNEY_FAKESS, BSDF_REFLECTION | BSDF_DIFFUSE); b.r = c * (1.0f - dt) * flat * dweight; b.A = rough; bsdf.add(b); }
            } else if (m.disney_scatter[0] == 0.0f && m.disney_scatter[1] == 0.0f && m.disney_scatter[2] == 0.0f) { Lobe b = mk_lobe(LB_DISNEY_DIFFUSE, BSDF_REFLECTION | BSDF_DIFFUSE); b.r = c * dweight; bsdf.add(b); }
            else { Lobe b = mk_lobe(LB_SPEC_T, BSDF_TRANSMISSION | BSDF_SPECULAR); b.t = RGB(1.0f); b.etaa = 1.0f; b.etab = e; bsdf.add(b); }   // + DisneyBSSRDF (disney.rs:768-776), sampled by the subsurface branch of k_shade
            { Lobe b = mk_lobe(LB_DISNEY_RETRO, BSDF_REFLECTION | BSDF_DIFFUSE); b.r = c * dweight; b.A = rough; bsdf.add(b); }
            if (sheen_weight > 0.0f) { Lobe b = mk_lobe(LB_DISNEY_SHEEN, BSDF_REFLECTION | BSDF_DIFFUSE); b.r = csheen * sheen_weight * dweight; bsdf.add(b); }
        }
        const float aspect = sqrtf(1.0f - m.disney[PT_DS_ANISOTROPIC] * 0.9f);
        const float ax = maxf(rough * rough / aspect, 0.001f), ay = maxf(rough * rough * aspect, 0.001f);
        const RGB cspec0 = lerp_rgb(metallic, lerp_rgb(m.disney[PT_DS_SPECULARTINT], RGB(1.0f) * (((e - 1.0f) * (e - 1.0f)) / ((e + 1.0f) * (e + 1.0f))), ctint), c);
        {
            Lobe b = mk_lobe(LB_MICRO_R, BSDF_REFLECTION | BSDF_GLOSSY); b.r = RGB(1.0f); b.ax = ax; b.ay = ay; b.sepg = 1;
            b.fresnel = FR_DISNEY; b.ce = cspec0; b.A = metallic; b.etaa = 1.0f; b.etab = e; bsdf.add(b);
        }
        const float cc = m.disney[PT_DS_CLEARCOAT];
        if (cc > 0.0f) { Lobe b = mk_lobe(LB_DISNEY_CLEARCOAT, BSDF_REFLECTION | BSDF_GLOSSY); b.B = cc; b.A = lerp_t(m.disney[PT_DS_CLEARCOATGLOSS], 0.1f, 0.001f); bsdf.add(b); }
        if (strans > 0.0f) {
            const RGB T = sqrt_rgb(c) * strans;
            Lobe b = mk_lobe(LB_MICRO_T, BSDF_TRANSMISSION | BSDF_GLOSSY); b.t = T; b.fresnel = FR_DIELECTRIC; b.etaa = 1.0f; b.etab = e;
            if (thin) {   // a Trowbridge-Reitz distribution with the roughness scaled by the index (disney.rs:812-821)
                const float rscaled = (0.65f * e - 0.35f) * rough;
                b.ax = maxf(rscaled * rscaled / aspect, 0.001f); b.ay = maxf(rscaled * rscaled * aspect, 0.001f);
            } else { b.ax = ax; b.ay = ay; b.sepg = 1; }
            bsdf.add(b);
        }
        if (thin) { Lobe b = mk_lobe(LB_LAMBERT_T, BSDF_TRANSMISSION | BSDF_DIFFUSE); b.t = c * dt; bsdf.add(b); }
        return true;
    }
    default: {  // PT_MAT_SUBSTRATE, substrate.rs:34-60
        bsdf.init(si, 1.0f);
        RGB d = E.spec(m, PT_MP_KD, m.kd).clamps(0.0f, PT_INF), s = E.spec(m, PT_MP_KS, m.ks).clamps(0.0f, PT_INF);
        float ru = E.flt(m, PT_MP_U_ROUGHNESS, m.u_roughness), rv = E.flt(m, PT_MP_V_ROUGHNESS, m.v_roughness);
        if (!d.is_black() || !s.is_black()) {
            if (m.remap_roughness) { ru = roughness_to_alpha(ru); rv = roughness_to_alpha(rv); }
            Lobe b = mk_lobe(LB_FRESNEL_BLEND, BSDF_REFLECTION | BSDF_GLOSSY); b.r = d; b.t = s; set_dist(b, ru, rv);
            bsdf.add(b);
            return true;
        }
        return false;
    }
    }
}

// Material::compute_scattering_functions incl. MixMaterial (mix.rs:25-50): both materials' BxDFs in one BSDF (frame and eta of
// the first), wrapped in ScaledBxDFs with scale = "amount" and 1 - "amount". `mats` = the scene's material array.
template <int MAXL, class ME, int DIFF> PT_DEV bool build_bsdf(const PtMaterial &m, const SurfaceInteraction &si, Bsdf<MAXL, DIFF> &bsdf, const ME &E, const PtMaterial *mats) {
    if constexpr (MAXL == 5) {
        bsdf.n1 = -1; bsdf.frozen = false;
        if (m.type == PT_MAT_MIX) {
            const RGB a = E.spec(m, PT_MP_KD, m.kd).clamps(0.0f, PT_INF);
            const RGB c = RGB(1.0f - a.r, 1.0f - a.g, 1.0f - a.b).clamps(0.0f, PT_INF);
            if (!build_bsdf_leaf(mats[m.mix[0]], si, bsdf, E)) return false;   // the reference unwraps si.bsdf here
            const int first = bsdf.n;
            bsdf.frozen = true;
            build_bsdf_leaf(mats[m.mix[1]], si, bsdf, E.plain());   // evaluated on the reference's `si2`: no ray differentials
            bsdf.frozen = false;
            bsdf.n1 = first; bsdf.s1 = a; bsdf.s2 = c;
            return true;
        }
    }
    return build_bsdf_leaf(m, si, bsdf, E);
}

}  // namespace ptd
