// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_math.h header).
// Implementations for ref_shading.h. Each function cites the reference lines it restates.
#include "ref_bssrdf.h"

namespace ref {

// ---- BxDF evaluation --------------------------------------------------------------------------
static Float shape_area(const Scene &s, uint32_t sh);
static Float default_pdf(V3 wo, V3 wi) {  // reflection.rs:439-445
    return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * INV_PI : 0.0f;
}
static inline Float pow5(Float v) { return (v * v) * (v * v) * v; }
static inline Float disney_gtr1(Float c, Float alpha) {   // disney.rs:222-226
    Float a2 = alpha * alpha;
    return (a2 - 1.0f) / (PI * dm_logf(a2) * (1.0f + (a2 - 1.0f) * c * c));
}
static inline Float disney_smithg_ggx(Float c, Float alpha) {   // disney.rs:228-233: no square root there
    Float a2 = alpha * alpha, c2 = c * c;
    return 1.0f / (c + (a2 + c2 - a2 * c2));
}

static RGB bxdf_f_unscaled(const Bxdf &b, V3 wo, V3 wi);
RGB bxdf_f(const Bxdf &b, V3 wo, V3 wi) { RGB f = bxdf_f_unscaled(b, wo, wi); return b.scaled ? b.scale * f : f; }
static RGB bxdf_f_unscaled(const Bxdf &b, V3 wo, V3 wi) {
    switch (b.kind) {
    case BX_LAMBERT_R: return b.r * INV_PI;  // reflection.rs:822-824
    case BX_BSSRDF: return RGB(bssrdf_sw(b.etab, wi)) * (b.etab * b.etab);  // bssrdf.rs:594-602 (mode == Radiance)
    case BX_LAMBERT_T: return b.t * INV_PI;  // :861-863
    case BX_OREN_NAYAR: {                    // :926-952
        Float sin_i = sin_theta(wi), sin_o = sin_theta(wo);
        Float max_cos = 0.0f;
        if (sin_i > 1e-4f && sin_o > 1e-4f) {
            Float sin_phii = sin_phi(wi), cos_phii = cos_phi(wi), sin_phio = sin_phi(wo), cos_phio = cos_phi(wo);
            Float dcos = cos_phii * cos_phio + sin_phii * sin_phio;
            max_cos = fmax_(dcos, 0.0f);
        }
        Float sin_alpha, tan_beta;
        if (abs_cos_theta(wi) > abs_cos_theta(wo)) { sin_alpha = sin_o; tan_beta = sin_i / abs_cos_theta(wi); }
        else { sin_alpha = sin_i; tan_beta = sin_o / abs_cos_theta(wo); }
        return b.r * INV_PI * (b.A + b.B * max_cos * sin_alpha * tan_beta);
    }
    case BX_DISNEY_DIFFUSE: {                // disney.rs:66-74
        Float fo = schlick_weight(abs_cos_theta(wo)), fi = schlick_weight(abs_cos_theta(wi));
        return b.r * INV_PI * (1.0f - fo / 2.0f) * (1.0f - fi / 2.0f);
    }
    case BX_DISNEY_FAKESS: {                 // disney.rs:104-124
        V3 wh = wi + wo;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return RGB(0.0f);
        wh = normalize(wh);
        Float cos_d = dot(wi, wh);
        Float fss90 = cos_d * cos_d * b.A;
        Float fo = schlick_weight(abs_cos_theta(wo)), fi = schlick_weight(abs_cos_theta(wi));
        Float fss = lerp_t(fo, 1.0f, fss90) * lerp_t(fi, 1.0f, fss90);
        Float ss = 1.25f * (fss * (1.0f / (abs_cos_theta(wo) + abs_cos_theta(wi)) - 0.5f) + 0.5f);
        return b.r / INV_PI * ss;            // `/ INV_PI` as written at disney.rs:123
    }
    case BX_DISNEY_RETRO: {                  // disney.rs:158-172
        V3 wh = wi + wo;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return RGB(0.0f);
        wh = normalize(wh);
        Float cos_d = dot(wi, wh);
        Float fo = schlick_weight(abs_cos_theta(wo)), fi = schlick_weight(abs_cos_theta(wi));
        Float rr = 2.0f * b.A * cos_d * cos_d;
        return b.r * INV_PI * rr * (fo + fi + fo * fi * (rr - 1.0f));
    }
    case BX_DISNEY_SHEEN: {                  // disney.rs:202-210
        V3 wh = wi + wo;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return RGB(0.0f);
        wh = normalize(wh);
        return b.r * schlick_weight(dot(wi, wh));
    }
    case BX_DISNEY_CLEARCOAT: {              // disney.rs:238-255 (A = gloss, B = weight)
        V3 wh = wi + wo;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return RGB(0.0f);
        wh = normalize(wh);
        Float dr = disney_gtr1(abs_cos_theta(wh), b.A);
        Float fr = fr_schlick(0.04f, dot(wo, wh));
        Float gr = disney_smithg_ggx(abs_cos_theta(wo), 0.25f) * disney_smithg_ggx(abs_cos_theta(wi), 0.25f);
        return RGB(fr * b.B * gr * dr / 4.0f);
    }
    case BX_SPEC_R: case BX_SPEC_T: return RGB(0.0f);
    case BX_FRESNEL_SPEC: return RGB(1.0f);  // App. A #9, reflection.rs:745-747
    case BX_MICRO_R: {                       // :981-1003
        Float cos_o = abs_cos_theta(wo), cos_i = abs_cos_theta(wi);
        V3 wh = wi + wo;
        if (cos_i == 0.0f || cos_o == 0.0f) return RGB(0.0f);
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return RGB(0.0f);
        wh = normalize(wh);
        RGB F = b.fresnel.evaluate(dot(wi, wh));
        Float d = b.dist.d(wh), g = b.dist.g(wo, wi);
        return b.r * d * g * F / (4.0f * cos_i * cos_o);
    }
    case BX_MICRO_T: {                       // :1059-1092
        if (same_hemisphere(wo, wi)) return RGB(0.0f);
        Float cos_o = cos_theta(wo), cos_i = cos_theta(wi);
        if (cos_i == 0.0f || cos_o == 0.0f) return RGB(0.0f);
        Float eta = (cos_theta(wo) > 0.0f) ? b.etab / b.etaa : b.etaa / b.etab;
        V3 wh = normalize(wo + wi * eta);
        if (wh.z < 0.0f) wh = -wh;
        if (dot(wo, wh) * dot(wi, wh) > 0.0f) return RGB(0.0f);
        RGB f = b.fresnel.evaluate(dot(wo, wh));
        Float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        Float factor = 1.0f / eta;  // TransportMode::Radiance
        Float s = b.dist.d(wh) * b.dist.g(wo, wi) * eta * eta * abs_dot(wi, wh) * abs_dot(wo, wh) * factor * factor /
                  (cos_i * cos_o * sqrt_denom * sqrt_denom);
        return (RGB(1.0f) - f) * b.t * std::fabs(s);
    }
    case BX_FRESNEL_BLEND: {                 // :1165-1182
        RGB diffuse = b.r * (RGB(1.0f) - b.rs) * (28.0f / (23.0f * PI)) *
                      (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wi))) * (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wo)));
        V3 wh = wi + wo;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return RGB(0.0f);
        wh = normalize(wh);
        RGB schlick = b.rs + (RGB(1.0f) - b.rs) * pow5(1.0f - dot(wi, wh));
        RGB specular = schlick * (b.dist.d(wh) / (4.0f * abs_dot(wi, wh) * fmax_(abs_cos_theta(wi), abs_cos_theta(wo))));
        return diffuse + specular;
    }
    }
    return RGB(0.0f);
}

Float bxdf_pdf(const Bxdf &b, V3 wo, V3 wi) {
    switch (b.kind) {
    case BX_LAMBERT_R: case BX_OREN_NAYAR: case BX_FRESNEL_SPEC: case BX_BSSRDF: return default_pdf(wo, wi);  // FresnelSpecular: :788-794
    case BX_DISNEY_DIFFUSE: case BX_DISNEY_FAKESS: case BX_DISNEY_RETRO: case BX_DISNEY_SHEEN: return default_pdf(wo, wi);
    case BX_DISNEY_CLEARCOAT: {  // disney.rs:276-292 (`wh = wi + wi` as written there)
        if (!same_hemisphere(wo, wi)) return 0.0f;
        V3 wh = wi + wi;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return 0.0f;
        wh = normalize(wh);
        Float dr = disney_gtr1(abs_cos_theta(wh), b.A);
        return dr * abs_cos_theta(wh) / (4.0f * dot(wo, wh));
    }
    case BX_LAMBERT_T: return !same_hemisphere(wo, wi) ? abs_cos_theta(wi) : 0.0f;              // :886-892 (no INV_PI, App. A #10)
    case BX_SPEC_R: case BX_SPEC_T: return 0.0f;
    case BX_MICRO_R: {  // :1021-1027
        if (!same_hemisphere(wo, wi)) return 0.0f;
        V3 wh = normalize(wo + wi);
        return b.dist.pdf(wo, wh) / (4.0f * dot(wo, wh));
    }
    case BX_MICRO_T: {  // :1112-1129
        if (same_hemisphere(wo, wi)) return 0.0f;
        Float eta = (cos_theta(wo) > 0.0f) ? b.etaa / b.etab : b.etab / b.etaa;
        V3 wh = normalize(wo + wi * eta);
        if (dot(wo, wh) * dot(wi, wh) > 0.0f) return 0.0f;
        Float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        Float dwh_dwi = std::fabs(eta * eta * dot(wi, wh)) / (sqrt_denom * sqrt_denom);
        return b.dist.pdf(wo, wh) * dwh_dwi;
    }
    case BX_FRESNEL_BLEND: {  // :1210-1221
        if (!same_hemisphere(wo, wi)) return 0.0f;
        V3 wh = normalize(wo + wi);
        Float pdf_wh = b.dist.pdf(wo, wh);
        return 0.5f * (abs_cos_theta(wi) * INV_PI + pdf_wh / (4.0f * dot(wo, wh)));
    }
    }
    return 0.0f;
}

static RGB bxdf_sample_f_unscaled(const Bxdf &b, V3 wo, V3 &wi, P2 u, Float &pdf, int &sampled);
RGB bxdf_sample_f(const Bxdf &b, V3 wo, V3 &wi, P2 u, Float &pdf, int &sampled) { RGB f = bxdf_sample_f_unscaled(b, wo, wi, u, pdf, sampled); return b.scaled ? b.scale * f : f; }
static RGB bxdf_sample_f_unscaled(const Bxdf &b, V3 wo, V3 &wi, P2 u, Float &pdf, int &sampled) {
    switch (b.kind) {
    case BX_DISNEY_CLEARCOAT: {  // disney.rs:257-274
        if (wo.z == 0.0f) return RGB(0.0f);
        Float a2 = b.A * b.A;
        Float ct = std::sqrt(fmax_((1.0f - dm_powf(a2, 1.0f - u.x)) / (1.0f - a2), 0.0f));
        Float st = std::sqrt(fmax_(1.0f - ct * ct, 0.0f));
        Float phi = 2.0f * PI * u.y;
        V3 wh(st * dm_cosf(phi), st * dm_sinf(phi), ct);   // spherical_direction (geometry.rs:27-33)
        if (!same_hemisphere(wo, wh)) wh = -wh;
        wi = reflect(wo, wh);
        if (!same_hemisphere(wo, wh)) return RGB(0.0f);
        pdf = bxdf_pdf(b, wo, wi);
        return bxdf_f_unscaled(b, wo, wi);
    }
    case BX_DISNEY_DIFFUSE: case BX_DISNEY_FAKESS: case BX_DISNEY_RETRO: case BX_DISNEY_SHEEN:
    case BX_LAMBERT_R: case BX_OREN_NAYAR: case BX_BSSRDF: {  // default BxDF::sample_f :392-403
        wi = cosine_sample_hemisphere(u);
        if (wo.z < 0.0f) wi.z *= -1.0f;
        pdf = bxdf_pdf(b, wo, wi);
        return bxdf_f_unscaled(b, wo, wi);
    }
    case BX_LAMBERT_T: {  // :865-876
        wi = cosine_sample_hemisphere(u);
        if (wo.z > 0.0f) wi.z *= -1.0f;
        pdf = bxdf_pdf(b, wo, wi);
        return bxdf_f_unscaled(b, wo, wi);
    }
    case BX_SPEC_R: {  // :636-643
        wi = V3(-wo.x, -wo.y, wo.z);
        pdf = 1.0f;
        return b.fresnel.evaluate(cos_theta(wi)) * b.r / abs_cos_theta(wi);
    }
    case BX_SPEC_T: {  // :687-708
        Float etai, etat;
        if (cos_theta(wo) > 0.0f) { etai = b.etaa; etat = b.etab; } else { etai = b.etab; etat = b.etaa; }
        if (!refract(wo, face_forward(V3(0, 0, 1), wo), etai / etat, wi)) return RGB(0.0f);
        pdf = 1.0f;
        RGB ft = b.t * (RGB(1.0f) - b.fresnel.evaluate(cos_theta(wi)));
        ft = ft * ((etai * etai) / (etat * etat));
        return ft / abs_cos_theta(wi);
    }
    case BX_FRESNEL_SPEC: {  // :749-786
        Float f = fr_dielectric(cos_theta(wo), b.etaa, b.etab);
        if (u.x < f) {
            wi = V3(-wo.x, -wo.y, wo.z);
            sampled = BSDF_SPECULAR | BSDF_REFLECTION;
            pdf = f;
            return b.r / abs_cos_theta(wi) * f;
        }
        Float etai, etat;
        if (cos_theta(wo) > 0.0f) { etai = b.etaa; etat = b.etab; } else { etai = b.etab; etat = b.etaa; }
        if (!refract(wo, face_forward(V3(0, 0, 1), wo), etai / etat, wi)) return RGB(0.0f);
        RGB ft = b.t * (1.0f - f);
        ft = ft * ((etai * etai) / (etat * etat));
        sampled = BSDF_SPECULAR | BSDF_TRANSMISSION;
        pdf = 1.0f - f;
        return ft / abs_cos_theta(wi);
    }
    case BX_MICRO_R: {  // :1005-1019
        if (wo.z == 0.0f) return RGB(0.0f);
        V3 wh = b.dist.sample_wh(wo, u);
        if (dot(wo, wh) < 0.0f) return RGB(0.0f);
        wi = reflect(wo, wh);
        if (!same_hemisphere(wo, wi)) return RGB(0.0f);
        pdf = b.dist.pdf(wo, wh) / (4.0f * dot(wo, wh));
        return bxdf_f_unscaled(b, wo, wi);
    }
    case BX_MICRO_T: {  // :1094-1110
        if (wo.z == 0.0f) return RGB(0.0f);
        V3 wh = b.dist.sample_wh(wo, u);
        if (dot(wo, wh) < 0.0f) return RGB(0.0f);
        Float eta = (cos_theta(wo) > 0.0f) ? b.etaa / b.etab : b.etab / b.etaa;
        if (!refract(wo, wh, eta, wi)) return RGB(0.0f);
        pdf = bxdf_pdf(b, wo, wi);
        return bxdf_f_unscaled(b, wo, wi);
    }
    case BX_FRESNEL_BLEND: {  // :1184-1208
        P2 uu = u;
        if (uu.x < 0.5f) {
            uu.x = fmin_(2.0f * uu.x, ONE_MINUS_EPSILON);
            wi = cosine_sample_hemisphere(uu);
            if (wo.z < 0.0f) wi.z *= -1.0f;
        } else {
            uu.x = fmin_(2.0f * (uu.x - 0.5f), ONE_MINUS_EPSILON);
            V3 wh = b.dist.sample_wh(wo, uu);
            wi = reflect(wo, wh);
            if (!same_hemisphere(wo, wi)) return RGB(0.0f);
        }
        pdf = bxdf_pdf(b, wo, wi);
        return bxdf_f_unscaled(b, wo, wi);
    }
    }
    return RGB(0.0f);
}

// ---- BSDF (reflection.rs:1527-1689) ------------------------------------------------------------
RGB BSDF::f(V3 wow, V3 wiw, int flags) const {
    V3 wi = world_to_local(wiw), wo = world_to_local(wow);
    if (wo.z == 0.0f) return RGB(0.0f);
    bool reflect_ = dot(wiw, ng) * dot(wow, ng) > 0.0f;
    RGB res(0.0f);
    for (int i = 0; i < n; ++i)
        if (b[i].matches(flags) && ((reflect_ && (b[i].type & BSDF_REFLECTION)) || (!reflect_ && (b[i].type & BSDF_TRANSMISSION))))
            res += bxdf_f(b[i], wo, wi);
    return res;
}
Float BSDF::pdf(V3 wow, V3 wiw, int flags) const {
    if (n == 0) return 0.0f;
    V3 wo = world_to_local(wow), wi = world_to_local(wiw);
    if (wo.z == 0.0f) return 0.0f;
    Float p = 0.0f; int matching = 0;
    for (int i = 0; i < n; ++i) if (b[i].matches(flags)) { ++matching; p += bxdf_pdf(b[i], wo, wi); }
    return matching > 0 ? p / (Float)matching : 0.0f;
}
RGB BSDF::sample_f(V3 wow, V3 &wiw, P2 u, Float &pdf, int ty, int &sampled) const {
    int matching = num_components(ty);
    if (matching == 0) { pdf = 0.0f; sampled = 0; return RGB(0.0f); }
    int comp = (int)std::min<uint64_t>(f2u_sat(std::floor(u.x * (Float)matching)), (uint64_t)(matching - 1));
    int idx = -1, count = comp;
    for (int i = 0; i < n; ++i) {
        bool m = b[i].matches(ty);
        if (m && count == 0) { idx = i; break; }
        else if (m) --count;
    }
    const Bxdf &bx = b[idx];
    P2 ur(fmin_(u.x * (Float)matching - (Float)comp, ONE_MINUS_EPSILON), u.y);
    V3 wo = world_to_local(wow), wi;
    if (wo.z == 0.0f) return RGB(0.0f);
    pdf = 0.0f;
    sampled = bx.type;
    RGB f = bxdf_sample_f(bx, wo, wi, ur, pdf, sampled);
    if (pdf == 0.0f) { sampled = 0; return RGB(0.0f); }
    wiw = local_to_world(wi);
    if (!(bx.type & BSDF_SPECULAR) && matching > 1)
        for (int i = 0; i < n; ++i) if (i != idx && b[i].matches(ty)) pdf += bxdf_pdf(b[i], wo, wi);
    if (matching > 1) pdf /= (Float)matching;
    if (!(bx.type & BSDF_SPECULAR)) {
        bool reflect_ = dot(wiw, ng) * dot(wow, ng) > 0.0f;
        f = RGB(0.0f);
        for (int i = 0; i < n; ++i)
            if (b[i].matches(ty) && ((reflect_ && (b[i].type & BSDF_REFLECTION)) || (!reflect_ && (b[i].type & BSDF_TRANSMISSION))))
                f += bxdf_f(b[i], wo, wi);
    }
    return f;
}

// ---- lights --------------------------------------------------------------------------------------
static void bounding_sphere(const Bounds3 &b, V3 &c, Float &rad) {  // bounds.rs:516-524
    c = (b.pmin + b.pmax) / 2.0f;
    rad = b.inside(c) ? length(b.pmax - c) : 0.0f;
}

RGB LightSampler::env_lookup(P2 st) const {  // mipmap.rs:202-223 (width 0 -> triangle(0)), :295-327, Repeat wrap
    const Scene &s = *scene;
    int w = (int)s.env_w, h = (int)s.env_h;
    Float sf = st.x * (Float)w - 0.5f, tf = st.y * (Float)h - 0.5f;
    int64_t s0 = f2i_sat(std::floor(sf)), t0 = f2i_sat(std::floor(tf));
    Float ds = sf - (Float)s0, dt = tf - (Float)t0;
    auto texel = [&](int64_t ss, int64_t tt) -> RGB {
        int64_t si = ss % w; if (si < 0) si += w;
        int64_t ti = tt % h; if (ti < 0) ti += h;
        return s.env_texels[(size_t)ti * w + si];
    };
    RGB tmp1 = texel(s0 + 1, t0 + 1) * (ds * dt);
    RGB tmp2 = texel(s0 + 1, t0) * (ds * (1.0f - dt));
    RGB tmp3 = texel(s0, t0 + 1) * ((1.0f - ds) * dt);
    RGB tmp4 = texel(s0, t0) * ((1.0f - ds) * (1.0f - dt));
    return tmp4 + tmp3 + tmp2 + tmp1;
}

RGB LightSampler::area_l(uint32_t li, V3 n, V3 w) const {  // diffuse.rs:71-79
    const PtLight &L = scene->lights[li];
    if (L.two_sided || dot(n, w) > 0.0f) return RGB(L.L[0], L.L[1], L.L[2]);
    return RGB(0.0f);
}

RGB LightSampler::light_le(uint32_t li, const Ray &r) const {  // infinite.rs:118-126; others 0
    const PtLight &L = scene->lights[li];
    if (L.type != PT_LIGHT_INFINITE) return RGB(0.0f);
    V3 w = normalize(xf_vector(m4_from(L.world_to_light), r.d));
    P2 st(spherical_phi(w) * INV2_PI, spherical_theta(w) * INV_PI);
    return env_lookup(st);
}

RGB LightSampler::power(uint32_t li) const {
    const PtLight &L = scene->lights[li];
    RGB c(L.L[0], L.L[1], L.L[2]);
    switch (L.type) {
    case PT_LIGHT_DIFFUSE_AREA: {  // diffuse.rs:82-84
        return c * shape_area(*scene, scene->prim_shape[L.prim]) * PI;
    }
    case PT_LIGHT_DISTANT: return c * PI * world_radius * world_radius;  // distant.rs:47-50
    case PT_LIGHT_POINT: return c * 4.0f * PI;                            // point.rs:44-46
    case PT_LIGHT_SPOT: return c * 2.0f * PI * (1.0f - 0.5f * (L.cos_falloff_start + L.cos_total_width));  // spot.rs:64-66
    case PT_LIGHT_INFINITE: {                                             // infinite.rs:103-109 (lookup width .5 => top level)
        RGB v = scene->env_power_lookup;   // map.lookup((.5,.5), .5): MIP level `levels - 2`, computed by the host
        return v * world_radius * world_radius * PI;
    }
    }
    return RGB(0.0f);
}

// Shape::area: triangle.rs:550-554, sphere.rs:291-293
static Float shape_area(const Scene &s, uint32_t sh) {
    if ((sh >> 30) == PT_SHAPE_SPHERE) {
        const PtSphere &S = s.spheres[sh & 0x3fffffffu];
        if (S.kind == PT_QUADRIC_DISK) return S.phi_max * 0.5f * (S.radius * S.radius - S.inner_radius * S.inner_radius);   // disk.rs:120-122
        return S.phi_max * S.radius * (S.z_max - S.z_min);
    }
    return s.tri_area(sh & 0x3fffffffu);
}

// Sphere::sample_interaction (sphere.rs:313-378) incl. Sphere::sample (:295-311). App. A #7: the cone branch leaves
// it.n = (0,0,0) (`it.n *= 1.0`), so one-sided sphere lights return L = 0 from sample_li and the shadow ray's far end is
// not offset.
static IData sphere_sample_interaction(const PtSphere &S, const IData &ref, P2 u, Float &pdf) {
    M4 o2w = m4_from(S.object_to_world), w2o = m4_from(S.world_to_object);
    V3 pcenter = xf_point(o2w, V3(0.0f, 0.0f, 0.0f));
    V3 porigin = offset_ray_origin(ref.p, ref.p_error, ref.n, pcenter - ref.p);
    IData it;
    if (distance_squared(porigin, pcenter) <= S.radius * S.radius) {
        V3 pobj = V3(0.0f, 0.0f, 0.0f) + uniform_sample_sphere(u) * S.radius;
        it.n = normalize(xf_normal_inv(w2o, pobj));
        if (S.reverse_orientation) it.n = it.n * -1.0f;
        pobj = pobj * (S.radius / length(pobj));
        V3 perr = vabs(pobj) * gamma(5);
        it.p = xf_point_abs_err(o2w, pobj, perr, it.p_error);
        pdf = 1.0f / (S.phi_max * S.radius * (S.z_max - S.z_min));
        V3 wi = it.p - ref.p;
        if (length_squared(wi) == 0.0f) pdf = 0.0f;
        else { wi = normalize(wi); pdf *= distance_squared(ref.p, it.p) / abs_dot(it.n, -wi); }
        if (std::isinf(pdf)) pdf = 0.0f;
        return it;
    }
    Float dc = length(ref.p - pcenter);
    Float invdc = 1.0f / dc;
    V3 wc = (pcenter - ref.p) * invdc, wcx, wcy;
    coordinate_system(wc, wcx, wcy);
    Float sin_thetamax = S.radius * invdc;
    Float sin_thetamax2 = sin_thetamax * sin_thetamax;
    Float inv_sin_thetamax = 1.0f / sin_thetamax;
    Float cos_thetamax = std::sqrt(fmax_(1.0f - sin_thetamax2, 0.0f));
    Float cos_theta = (cos_thetamax - 1.0f) * u.x + 1.0f;
    Float sin_theta2 = 1.0f - cos_theta * cos_theta;
    if (sin_thetamax2 < 0.00068523f) { sin_theta2 = sin_thetamax2 * u.x; cos_theta = std::sqrt(1.0f - sin_theta2); }
    Float cos_alpha = sin_theta2 * inv_sin_thetamax + cos_theta * std::sqrt(fmax_(1.0f - sin_theta2 * inv_sin_thetamax * inv_sin_thetamax, 0.0f));
    Float sin_alpha = std::sqrt(fmax_(1.0f - cos_alpha * cos_alpha, 0.0f));
    Float phi = u.y * 2.0f * PI;
    // spherical_direction_basis(sin_alpha, cos_alpha, phi, -wcx, -wcy, -wc) (geometry.rs:36-38)
    V3 nworld = (-wcx) * sin_alpha * dm_cosf(phi) + (-wcy) * sin_alpha * dm_sinf(phi) + (-wc) * cos_alpha;
    V3 pworld = pcenter + V3(nworld.x, nworld.y, nworld.z) * S.radius;
    it.p = pworld; it.p_error = vabs(pworld) * gamma(5); it.n = V3(0.0f, 0.0f, 0.0f);
    pdf = 1.0f / (2.0f * PI * (1.0f - cos_thetamax));
    return it;
}

RGB LightSampler::sample_li(uint32_t li, const IData &ref, P2 u, V3 &wi, Float &pdf, IData &p1) const {
    const Scene &s = *scene;
    const PtLight &L = s.lights[li];
    p1 = IData();
    switch (L.type) {
    case PT_LIGHT_DIFFUSE_AREA: {  // diffuse.rs:95-112 + shape.rs:40-58 + triangle.rs:556-584
        uint32_t sh = s.prim_shape[L.prim];
        if ((sh >> 30) == PT_SHAPE_SPHERE && s.spheres[sh & 0x3fffffffu].kind == PT_QUADRIC_DISK) {
            // Disk::sample (disk.rs:124-139) + the default Shape::sample_interaction (shape.rs:40-52)
            const PtSphere &S = s.spheres[sh & 0x3fffffffu];
            P2 pd = concentric_sample_disk(u);
            M4 o2w = m4_from(S.object_to_world), w2o = m4_from(S.world_to_object);
            IData it;
            it.n = normalize(xf_normal_inv(w2o, V3(0.0f, 0.0f, 0.1f)));
            if (S.reverse_orientation) it.n = it.n * -1.0f;
            it.p = xf_point_abs_err(o2w, V3(pd.x * S.radius, pd.y * S.radius, S.z_min), V3(0.0f, 0.0f, 0.0f), it.p_error);
            pdf = 1.0f / shape_area(s, sh);
            V3 w = it.p - ref.p;
            if (length_squared(w) == 0.0f) pdf = 0.0f;
            else {
                w = normalize(w);
                pdf *= distance_squared(ref.p, it.p) / abs_dot(it.n, -w);
                if (std::isinf(pdf)) pdf = 0.0f;
            }
            if (pdf == 0.0f || length_squared(it.p - ref.p) == 0.0f) { pdf = 0.0f; return RGB(0.0f); }
            wi = normalize(it.p - ref.p);
            p1 = it;
            return area_l(li, it.n, -wi);
        }
        if ((sh >> 30) == PT_SHAPE_SPHERE) {
            IData it = sphere_sample_interaction(s.spheres[sh & 0x3fffffffu], ref, u, pdf);
            if (pdf == 0.0f || length_squared(it.p - ref.p) == 0.0f) { pdf = 0.0f; return RGB(0.0f); }
            wi = normalize(it.p - ref.p);
            p1 = it;
            return area_l(li, it.n, -wi);
        }
        uint32_t tri = sh & 0x3fffffffu;
        P2 b = uniform_sample_triangle(u);
        V3 p0, p1v, p2; s.tri_positions(tri, p0, p1v, p2);
        IData it;
        it.p = p0 * b.x + p1v * b.y + p2 * (1.0f - b.x - b.y);
        it.n = normalize(cross(p1v - p0, p2 - p0));
        uint8_t fl = s.tri_flags[tri];
        if (fl & PT_TRI_HAS_N) {
            uint32_t i0 = s.idx[3 * tri], i1 = s.idx[3 * tri + 1], i2 = s.idx[3 * tri + 2];
            V3 ns = s.N[i0] * b.x + s.N[i1] * b.y + s.N[i2] * (1.0f - b.x - b.y);
            it.n = face_forward(it.n, ns);
        } else if (((fl & PT_TRI_REVERSE_ORIENTATION) != 0) ^ ((fl & PT_TRI_SWAPS_HANDEDNESS) != 0)) {
            it.n = it.n * -1.0f;
        }
        V3 pabs = vabs(p0 * b.x) + vabs(p1v * b.y) + vabs(p2 * (1.0f - b.x - b.y));
        it.p_error = pabs * gamma(6);
        pdf = 1.0f / s.tri_area(tri);
        // Shape::sample_interaction
        V3 w = it.p - ref.p;
        if (length_squared(w) == 0.0f) pdf = 0.0f;
        else {
            w = normalize(w);
            pdf *= distance_squared(ref.p, it.p) / abs_dot(it.n, -w);
            if (std::isinf(pdf)) pdf = 0.0f;
        }
        if (pdf == 0.0f || length_squared(it.p - ref.p) == 0.0f) { pdf = 0.0f; return RGB(0.0f); }
        wi = normalize(it.p - ref.p);
        p1 = it;
        return area_l(li, it.n, -wi);
    }
    case PT_LIGHT_DISTANT: {  // distant.rs:64-84
        V3 wl(L.dir[0], L.dir[1], L.dir[2]);
        wi = wl; pdf = 1.0f;
        p1.p = ref.p + wl * (2.0f * world_radius);
        return RGB(L.L[0], L.L[1], L.L[2]);
    }
    case PT_LIGHT_POINT: {  // point.rs:52-69
        V3 pl(L.pos[0], L.pos[1], L.pos[2]);
        wi = normalize(pl - ref.p); pdf = 1.0f;
        p1.p = pl;
        return RGB(L.L[0], L.L[1], L.L[2]) / distance_squared(pl, ref.p);
    }
    case PT_LIGHT_SPOT: {  // spot.rs:48-56,71-87
        V3 pl(L.pos[0], L.pos[1], L.pos[2]);
        wi = normalize(pl - ref.p); pdf = 1.0f;
        p1.p = pl;
        V3 wl = normalize(xf_vector(m4_from(L.world_to_light), -wi));
        Float cos_theta = wl.z, fall;
        if (cos_theta < L.cos_total_width) fall = 0.0f;
        else if (cos_theta >= L.cos_falloff_start) fall = 1.0f;
        else { Float delta = (cos_theta - L.cos_total_width) / (L.cos_falloff_start - L.cos_total_width); fall = (delta * delta) * (delta * delta); }
        return RGB(L.L[0], L.L[1], L.L[2]) * fall / distance_squared(pl, ref.p);
    }
    case PT_LIGHT_INFINITE: {  // infinite.rs:140-177
        Float map_pdf = 0.0f;
        P2 uv = env_dist.sample_continuous(u, map_pdf);
        if (map_pdf == 0.0f) return RGB(0.0f);
        Float theta = uv.y * PI, phi = uv.x * 2.0f * PI;
        Float cos_t = dm_cosf(theta), sin_t = dm_sinf(theta);
        Float sin_p = dm_sinf(phi), cos_p = dm_cosf(phi);
        V3 v(sin_t * cos_p, sin_t * sin_p, cos_t);
        wi = xf_vector(m4_from(L.light_to_world), v);
        pdf = map_pdf / (2.0f * PI * PI * sin_t);
        if (sin_t == 0.0f) pdf = 0.0f;
        p1.p = ref.p + wi * (2.0f * world_radius);
        return env_lookup(uv);
    }
    }
    pdf = 0.0f;
    return RGB(0.0f);
}

Float LightSampler::pdf_li(uint32_t li, const IData &ref, V3 wi) const {
    const Scene &s = *scene;
    const PtLight &L = s.lights[li];
    switch (L.type) {
    case PT_LIGHT_DIFFUSE_AREA: {  // shape.rs:63-82 (intersect with s = None)
        if ((s.prim_shape[L.prim] >> 30) == PT_SHAPE_SPHERE && s.spheres[s.prim_shape[L.prim] & 0x3fffffffu].kind == PT_QUADRIC_DISK) {   // Shape::pdf_wi default
            Ray ray = spawn_ray(ref, wi);
            Float thit; SurfaceInteraction il;
            if (!s.sphere_intersect(s.prim_shape[L.prim] & 0x3fffffffu, ray, thit, il, false)) return 0.0f;
            Float pdf = distance_squared(ref.p, il.p) / (dot(il.n, -wi) * shape_area(s, s.prim_shape[L.prim]));
            if (std::isinf(pdf)) pdf = 0.0f;
            return pdf;
        }
        if ((s.prim_shape[L.prim] >> 30) == PT_SHAPE_SPHERE) {  // Sphere::pdf_wi (sphere.rs:380-395)
            uint32_t si_ = s.prim_shape[L.prim] & 0x3fffffffu;
            const PtSphere &S = s.spheres[si_];
            V3 pcenter = xf_point(m4_from(S.object_to_world), V3(0.0f, 0.0f, 0.0f));
            V3 porigin = offset_ray_origin(ref.p, ref.p_error, ref.n, pcenter - ref.p);
            if (distance_squared(porigin, pcenter) <= S.radius * S.radius) {  // shape_pdfwi (shape.rs:117-136)
                Ray ray = spawn_ray(ref, wi);
                Float thit; SurfaceInteraction il;
                if (!s.sphere_intersect(si_, ray, thit, il, false)) return 0.0f;
                Float pdf = distance_squared(ref.p, il.p) / (dot(il.n, -wi) * shape_area(s, s.prim_shape[L.prim]));
                if (std::isinf(pdf)) pdf = 0.0f;
                return pdf;
            }
            Float sin_thetamax2 = S.radius * S.radius / distance_squared(ref.p, pcenter);
            Float cos_thetamax = std::sqrt(fmax_(1.0f - sin_thetamax2, 0.0f));
            return 1.0f / (2.0f * PI * (1.0f - cos_thetamax));
        }
        uint32_t tri = s.prim_shape[L.prim] & 0x3fffffffu;
        Ray ray = spawn_ray(ref, wi);
        Float t, b[3];
        if (!s.tri_intersect(tri, ray, t, b)) return 0.0f;
        SurfaceInteraction il;
        s.tri_fill_interaction(tri, ray, t, b, false, il);
        Float pdf = distance_squared(ref.p, il.p) / (dot(il.n, -wi) * s.tri_area(tri));
        if (std::isinf(pdf)) pdf = 0.0f;
        return pdf;
    }
    case PT_LIGHT_INFINITE: {  // infinite.rs:128-138
        V3 w = xf_vector(m4_from(L.world_to_light), wi);
        Float theta = spherical_theta(w), phi = spherical_phi(w);
        Float sin_t = dm_sinf(theta);
        if (sin_t == 0.0f) return 0.0f;
        return env_dist.pdf(P2(phi * INV2_PI, theta * INV_PI)) / (2.0f * PI * PI * sin_t);
    }
    default: return 0.0f;
    }
}

// ---- light distributions -----------------------------------------------------------------------
void LightSampler::init(const Scene &s, int requested) {
    scene = &s;
    bounding_sphere(s.wb, world_center, world_radius);  // Light::preprocess (distant.rs:52-58, infinite.rs:111-116)
    if (s.env_w > 0) env_dist = Distribution2D(s.env_importance.data(), 2 * s.env_w, 2 * s.env_h);
    size_t nl = s.lights.size();
    strategy = requested >= PT_LS_SPATIAL ? (int)PT_LS_SPATIAL : requested;   // (PT_LS_SPATIAL_EAGER / _LAZY: how the DEVICE fills its voxels; the same distribution)
    if (requested == PT_LS_UNIFORM || nl == 1) strategy = PT_LS_UNIFORM;  // lightdistrib.rs:21
    if (nl == 0) { strategy = PT_LS_UNIFORM; fixed = std::make_shared<Distribution1D>(std::vector<Float>()); return; }  // Distribution1D::new(vec![]) (lightdistrib.rs:79)
    if (strategy == PT_LS_UNIFORM) fixed = std::make_shared<Distribution1D>(std::vector<Float>(nl, 1.0f));
    else if (strategy == PT_LS_POWER) {  // integrator.rs:239-247
        std::vector<Float> p;
        for (size_t i = 0; i < nl; ++i) p.push_back(power((uint32_t)i).y());
        fixed = std::make_shared<Distribution1D>(p);
    } else {  // spatial, lightdistrib.rs:112-128
        V3 diag = s.wb.diagonal();
        Float bmax = diag[s.wb.maximum_extent()];
        for (int i = 0; i < 3; ++i) nvox[i] = std::max<size_t>(1, (size_t)f2u_sat(std::round(diag[i] / bmax * 64.0f)));
        grid = std::vector<std::atomic<const Distribution1D *>>(nvox[0] * nvox[1] * nvox[2]);
        for (auto &g : grid) g.store(nullptr);
    }
}

const Distribution1D *LightSampler::compute_distribution(const int64_t pi[3]) const {  // lightdistrib.rs:151-228
    const Scene &s = *scene;
    V3 p0((Float)pi[0] / (Float)nvox[0], (Float)pi[1] / (Float)nvox[1], (Float)pi[2] / (Float)nvox[2]);
    V3 p1((Float)(pi[0] + 1) / (Float)nvox[0], (Float)(pi[1] + 1) / (Float)nvox[1], (Float)(pi[2] + 1) / (Float)nvox[2]);
    Bounds3 vb(s.wb.lerp3(p0), s.wb.lerp3(p1));
    const int nsamples = 128;
    size_t nl = s.lights.size();
    std::vector<Float> contrib(nl, 0.0f);
    for (int i = 0; i < nsamples; ++i) {
        V3 p(radical_inverse(0, (uint64_t)i), radical_inverse(1, (uint64_t)i), radical_inverse(2, (uint64_t)i));
        IData intr; intr.p = vb.lerp3(p); intr.wo = V3(1, 0, 0);
        P2 u(radical_inverse(3, (uint64_t)i), radical_inverse(4, (uint64_t)i));
        for (size_t j = 0; j < nl; ++j) {
            Float pdf = 0.0f; V3 wi; IData vis;
            RGB Li = sample_li((uint32_t)j, intr, u, wi, pdf, vis);
            if (pdf > 0.0f) contrib[j] += Li.y() / pdf;
        }
    }
    Float sum = 0.0f;
    for (size_t j = 0; j < nl; ++j) sum += contrib[j];
    Float avg = sum / ((Float)nsamples * (Float)nl);
    Float min_contrib = (avg > 0.0f) ? 0.001f * avg : 1.0f;
    for (size_t j = 0; j < nl; ++j) contrib[j] = fmax_(contrib[j], min_contrib);
    return new Distribution1D(contrib);
}

const Distribution1D *LightSampler::lookup(V3 p) const {  // lightdistrib.rs:233-339
    if (strategy != PT_LS_SPATIAL) return fixed.get();
    V3 off = scene->wb.offset(p);
    int64_t pi[3];
    for (int i = 0; i < 3; ++i) pi[i] = clampv<int64_t>(f2i_sat(off[i] * (Float)nvox[i]), 0, (int64_t)nvox[i] - 1);
    size_t cell = ((size_t)pi[2] * nvox[1] + (size_t)pi[1]) * nvox[0] + (size_t)pi[0];
    const Distribution1D *d = grid[cell].load(std::memory_order_acquire);
    if (d) return d;
    std::lock_guard<std::mutex> g(mu);
    d = grid[cell].load(std::memory_order_acquire);
    if (d) return d;
    d = compute_distribution(pi);
    grid[cell].store(d, std::memory_order_release);
    return d;
}

}  // namespace ref
