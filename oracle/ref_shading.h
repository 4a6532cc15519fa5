// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_math.h header).
// ref_shading.h: BSDF/BxDFs, materials, lights, light distributions, one-light MIS, PathIntegrator::li.
//   core/reflection.rs:27-190,384-1229,1495-1717; materials/*.rs; core/sampling.rs;
//   core/light.rs; lights/{diffuse,distant,point,infinite}.rs; core/shape.rs:40-82;
//   core/lightdistrib.rs; core/integrator.rs:81-237; integrators/path.rs:79-222.
#pragma once
#include "ref_scene.h"
#include "ref_sampler.h"
#include <memory>
#include <unordered_map>
#include <mutex>
#include <atomic>

namespace ref {

// ---- sampling (core/sampling.rs) -----------------------------------------------------------
struct Distribution1D {
    std::vector<Float> func, cdf;
    Float func_int = 0;
    Distribution1D() {}
    explicit Distribution1D(const std::vector<Float> &f) : func(f) {  // sampling.rs:12-34
        size_t n = func.size();
        cdf.assign(n + 1, 0.0f);
        for (size_t i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / (Float)n;
        func_int = cdf[n];
        if (func_int == 0.0f) { for (size_t i = 1; i < n + 1; ++i) cdf[i] = (Float)i / (Float)n; }
        else { for (size_t i = 1; i < n + 1; ++i) cdf[i] /= func_int; }
    }
    size_t count() const { return func.size(); }
    Float sample_continuous(Float u, Float *pdf, size_t *off) const {  // sampling.rs:38-64
        size_t offset = (size_t)find_interval((int)cdf.size(), [&](int i) { return cdf[i] <= u; });
        if (off) *off = offset;
        Float du = u - cdf[offset];
        Float diff = cdf[offset + 1] - cdf[offset];
        if (diff > 0.0f) du /= diff;
        if (pdf) *pdf = (func_int > 0.0f) ? func[offset] / func_int : 0.0f;
        return ((Float)offset + du) / (Float)count();
    }
    size_t sample_discrete(Float u, Float *pdf, Float *uremapped = nullptr) const {  // sampling.rs:66-85
        size_t offset = (size_t)find_interval((int)cdf.size(), [&](int i) { return cdf[i] <= u; });
        if (pdf) *pdf = (func_int > 0.0f) ? func[offset] / (func_int * (Float)count()) : 0.0f;
        if (uremapped) *uremapped = (u - cdf[offset]) / (cdf[offset + 1] - cdf[offset]);
        return offset;
    }
    Float discrete_pdf(size_t index) const { return func[index] / (func_int * (Float)count()); }
};
struct Distribution2D {  // sampling.rs:94-145
    std::vector<Distribution1D> cond;
    Distribution1D marginal;
    Distribution2D() {}
    Distribution2D(const Float *func, size_t nu, size_t nv) {
        for (size_t v = 0; v < nv; ++v) cond.emplace_back(std::vector<Float>(func + v * nu, func + (v + 1) * nu));
        std::vector<Float> mf;
        for (size_t v = 0; v < nv; ++v) mf.push_back(cond[v].func_int);
        marginal = Distribution1D(mf);
    }
    P2 sample_continuous(P2 u, Float &pdf) const {
        Float pdfs[2]; size_t v;
        Float d1 = marginal.sample_continuous(u.y, &pdfs[1], &v);
        Float d0 = cond[v].sample_continuous(u.x, &pdfs[0], nullptr);
        pdf = pdfs[0] * pdfs[1];
        return P2(d0, d1);
    }
    Float pdf(P2 p) const {
        size_t nu = cond[0].count(), nv = marginal.count();
        size_t iu = clampv<size_t>((size_t)f2u_sat(p.x * (Float)nu), 0, nu - 1);
        size_t iv = clampv<size_t>((size_t)f2u_sat(p.y * (Float)nv), 0, nv - 1);
        return cond[iv].func[iu] / marginal.func_int;
    }
};

inline P2 concentric_sample_disk(P2 u) {  // sampling.rs:153-176
    Float ox = u.x * 2.0f - 1.0f, oy = u.y * 2.0f - 1.0f;
    if (ox == 0.0f && oy == 0.0f) return P2(0.0f, 0.0f);
    Float theta, r;
    if (std::fabs(ox) > std::fabs(oy)) { r = ox; theta = PI_OVER4 * (oy / ox); }
    else { r = oy; theta = PI_OVER2 - PI_OVER4 * (ox / oy); }
    return P2(dm_cosf(theta) * r, dm_sinf(theta) * r);
}
inline V3 cosine_sample_hemisphere(P2 u) {  // sampling.rs:188-193
    P2 d = concentric_sample_disk(u);
    Float z = std::sqrt(fmax_(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return V3(d.x, d.y, z);
}
inline P2 uniform_sample_triangle(P2 u) { Float su0 = std::sqrt(u.x); return P2(1.0f - su0, u.y * su0); }
inline V3 uniform_sample_sphere(P2 u) {  // sampling.rs:212-218
    Float z = 1.0f - 2.0f * u.x;
    Float r = std::sqrt(fmax_(1.0f - z * z, 0.0f));
    Float phi = 2.0f * PI * u.y;
    return V3(r * dm_cosf(phi), r * dm_sinf(phi), z);
}
inline Float power_heuristic(int nf, Float fpdf, int ng, Float gpdf) {  // sampling.rs:328-333
    Float f = (Float)nf * fpdf, g = (Float)ng * gpdf;
    return (f * f) / (f * f + g * g);
}

// ---- BxDFs (core/reflection.rs) ---------------------------------------------------------------
enum { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };
enum BxdfKind { BX_LAMBERT_R, BX_LAMBERT_T, BX_OREN_NAYAR, BX_SPEC_R, BX_SPEC_T, BX_FRESNEL_SPEC, BX_MICRO_R, BX_MICRO_T, BX_FRESNEL_BLEND, BX_BSSRDF,
                BX_DISNEY_DIFFUSE, BX_DISNEY_FAKESS, BX_DISNEY_RETRO, BX_DISNEY_SHEEN, BX_DISNEY_CLEARCOAT };   // materials/disney.rs
enum FresnelKind { FR_NOOP, FR_DIELECTRIC, FR_CONDUCTOR, FR_DISNEY };

inline Float cos_theta(V3 w) { return w.z; }
inline Float cos2_theta(V3 w) { return w.z * w.z; }
inline Float abs_cos_theta(V3 w) { return std::fabs(w.z); }
inline Float sin2_theta(V3 w) { return fmax_(1.0f - cos2_theta(w), 0.0f); }
inline Float sin_theta(V3 w) { return std::sqrt(sin2_theta(w)); }
inline Float tan_theta(V3 w) { return sin_theta(w) / cos_theta(w); }
inline Float tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
inline Float cos_phi(V3 w) { Float s = sin_theta(w); return (s == 0.0f) ? 1.0f : clampv(w.x / s, -1.0f, 1.0f); }
inline Float sin_phi(V3 w) { Float s = sin_theta(w); return (s == 0.0f) ? 0.0f : clampv(w.y / s, -1.0f, 1.0f); }
inline Float cos2_phi(V3 w) { return cos_phi(w) * cos_phi(w); }
inline Float sin2_phi(V3 w) { return sin_phi(w) * sin_phi(w); }
inline bool same_hemisphere(V3 w, V3 wp) { return w.z * wp.z > 0.0f; }
inline V3 reflect(V3 wo, V3 n) { return -wo + n * 2.0f * dot(wo, n); }
inline bool refract(V3 wi, V3 n, Float eta, V3 &wt) {  // reflection.rs:160-174
    Float cos_thetai = dot(n, wi);
    Float sin2_thetai = fmax_(1.0f - cos_thetai * cos_thetai, 0.0f);
    Float sin2_thetat = eta * eta * sin2_thetai;
    if (sin2_thetat >= 1.0f) return false;
    Float cos_thetat = std::sqrt(1.0f - sin2_thetat);
    wt = n * (eta * cos_thetai - cos_thetat) + (-wi) * eta;
    return true;
}
// reflection.rs:29-52
inline Float fr_dielectric(Float cos_thetai, Float etai, Float etat) {
    cos_thetai = clampv(cos_thetai, -1.0f, 1.0f);
    bool entering = cos_thetai > 0.0f;
    if (!entering) { std::swap(etai, etat); cos_thetai = std::fabs(cos_thetai); }
    Float sin_thetai = std::sqrt(fmax_(0.0f, 1.0f - cos_thetai * cos_thetai));
    Float sin_thetat = etai / etat * sin_thetai;
    if (sin_thetat >= 1.0f) return 1.0f;
    Float cos_thetat = std::sqrt(fmax_(0.0f, 1.0f - sin_thetat * sin_thetat));
    Float rparl = ((etat * cos_thetai) - (etai * cos_thetat)) / ((etat * cos_thetai) + (etai * cos_thetat));
    Float rperp = ((etai * cos_thetai) - (etat * cos_thetat)) / ((etai * cos_thetai) + (etat * cos_thetat));
    return (rparl * rparl + rperp * rperp) / 2.0f;
}
// reflection.rs:54-76
inline RGB fr_conductor(Float cos_thetai, RGB etai, RGB etat, RGB k) {
    cos_thetai = clampv(cos_thetai, -1.0f, 1.0f);
    RGB eta = etat / etai, etak = k / etai;
    Float cos2 = cos_thetai * cos_thetai;
    Float sin2 = 1.0f - cos2;
    RGB eta2 = eta * eta, etak2 = etak * etak;
    RGB t0 = eta2 - etak2 - RGB(sin2);
    RGB a2plusb2 = sqrt_rgb(t0 * t0 + eta2 * etak2 * 4.0f);
    RGB t1 = a2plusb2 + RGB(cos2);
    RGB a = sqrt_rgb((a2plusb2 + t0) * 0.5f);
    RGB t2 = a * cos_thetai * 2.0f;
    RGB rs = (t1 - t2) / (t1 + t2);
    RGB t3 = a2plusb2 * cos2 + RGB(sin2 * sin2);
    RGB t4 = t2 * sin2;
    RGB rp = rs * (t3 - t4) / (t3 + t4);
    return (rp + rs) * 0.5f;
}

// materials/disney.rs:31-52
inline Float schlick_weight(Float c) { Float m = clampv(1.0f - c, 0.0f, 1.0f); return (m * m) * (m * m) * m; }
inline Float lerp_t(Float t, Float x, Float y) { return x * (1.0f - t) + y * t; }   // pbrt.rs:136-144
inline RGB lerp_t(Float t, RGB x, RGB y) { return x * (1.0f - t) + y * t; }
inline Float fr_schlick(Float r0, Float c) { return lerp_t(schlick_weight(c), r0, 1.0f); }
inline RGB fr_schlicks(RGB r0, Float c) { return lerp_t(schlick_weight(c), r0, RGB(1.0f)); }

struct Fresnel {
    int kind = FR_NOOP;
    Float etai = 1, etat = 1;
    RGB ci, ct, k;
    RGB r0; Float metallic = 0;   // DisneyFresnel (disney.rs:283-303), eta in `etat`
    RGB evaluate(Float cosi) const {
        if (kind == FR_NOOP) return RGB(1.0f);
        if (kind == FR_DISNEY) return lerp_t(metallic, RGB(fr_dielectric(cosi, 1.0f, etat)), fr_schlicks(r0, cosi));
        if (kind == FR_DIELECTRIC) return RGB(fr_dielectric(cosi, etai, etat));
        return fr_conductor(std::fabs(cosi), ci, ct, k);
    }
};

// TrowbridgeReitzDistribution (core/microfacet.rs:249-406), sample_visible_area = true
struct TRDist {
    Float ax = 0, ay = 0;
    static Float roughness_to_alpha(Float roughness) {  // microfacet.rs:334-340
        roughness = fmax_(roughness, 1e-3f);
        Float x = dm_logf(roughness);
        return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
    }
    Float d(V3 wh) const {  // microfacet.rs:343-353
        Float t2 = tan2_theta(wh);
        if (std::isinf(t2)) return 0.0f;
        Float c4 = cos2_theta(wh) * cos2_theta(wh);
        Float e = (cos2_phi(wh) / (ax * ax) + sin2_phi(wh) / (ay * ay)) * t2;
        return 1.0f / (PI * ax * ay * c4 * (1.0f + e) * (1.0f + e));
    }
    Float lambda(V3 w) const {  // microfacet.rs:355-367
        Float abs_tan = std::fabs(tan_theta(w));
        if (std::isinf(abs_tan)) return 0.0f;
        Float alpha = std::sqrt(cos2_phi(w) * ax * ax + sin2_phi(w) * ay * ay);
        Float a2t2 = (alpha * abs_tan) * (alpha * abs_tan);
        return (-1.0f + std::sqrt(1.0f + a2t2)) / 2.0f;
    }
    Float g1(V3 w) const { return 1.0f / (1.0f + lambda(w)); }
    bool separable = false;   // DisneyMicrofacetDistribution::g (disney.rs:376-379)
    Float g(V3 wo, V3 wi) const { return separable ? g1(wo) * g1(wi) : 1.0f / (1.0f + lambda(wo) + lambda(wi)); }
    Float pdf(V3 wo, V3 wh) const {  // microfacet.rs:120-131 (samplevis)
        return d(wh) * g1(wo) * abs_dot(wo, wh) / abs_cos_theta(wo);
    }
    static void sample11(Float cos_t, Float u1, Float u2, Float &sx, Float &sy) {  // microfacet.rs:249-291
        if (cos_t > 0.9999f) {
            Float r = std::sqrt(u1 / (1.0f - u1));
            Float phi = 6.28318530718f * u2;
            sx = r * dm_cosf(phi); sy = r * dm_sinf(phi);
            return;
        }
        Float sin_t = std::sqrt(fmax_(0.0f, 1.0f - cos_t * cos_t));
        Float tan_t = sin_t / cos_t;
        Float a = 1.0f / tan_t;
        Float G1 = 2.0f / (1.0f + std::sqrt(1.0f + 1.0f / (a * a)));
        Float A = 2.0f * u1 / G1 - 1.0f;
        Float tmp = 1.0f / (A * A - 1.0f);
        if (tmp > 1e10f) tmp = 1e10f;
        Float B = tan_t;
        Float D = std::sqrt(fmax_(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.0f));
        Float sx1 = B * tmp - D, sx2 = B * tmp + D;
        sx = (A < 0.0f || sx2 > 1.0f / tan_t) ? sx1 : sx2;
        Float S;
        if (u2 > 0.5f) { S = 1.0f; u2 = 2.0f * (u2 - 0.5f); }
        else { S = -1.0f; u2 = 2.0f * (0.5f - u2); }
        Float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) /
                  (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
        sy = S * z * std::sqrt(1.0f + sx * sx);
    }
    static V3 sample_stretched(V3 wi, Float ax, Float ay, Float u1, Float u2) {  // microfacet.rs:293-316
        V3 wis = normalize(V3(ax * wi.x, ay * wi.y, wi.z));
        Float sx, sy;
        sample11(cos_theta(wis), u1, u2, sx, sy);
        Float tmp = cos_phi(wis) * sx - sin_phi(wis) * sy;
        sy = sin_phi(wis) * sx + cos_phi(wis) * sy;
        sx = tmp;
        sx = ax * sx; sy = ay * sy;
        return normalize(V3(-sx, -sy, 1.0f));
    }
    V3 sample_wh(V3 wo, P2 u) const {  // microfacet.rs:369-406, samplevis branch
        bool flip = wo.z < 0.0f;
        V3 wh = sample_stretched(flip ? -wo : wo, ax, ay, u.x, u.y);
        if (flip) wh = -wh;
        return wh;
    }
};

struct Bxdf {
    int kind = BX_LAMBERT_R;
    int type = 0;
    RGB r, t;            // R / T / Rd
    RGB rs;              // FresnelBlend
    Float A = 0, B = 0;  // Oren-Nayar
    Float etaa = 1, etab = 1;
    Fresnel fresnel;
    TRDist dist;
    bool scaled = false; RGB scale;   // ScaledBxDF (reflection.rs:466-517): f and sample_f times `scale`, pdf unchanged
    bool matches(int flags) const { return (type & flags) == type; }
};

struct BSDF {
    Float eta = 1;
    V3 ns, ng, ss, ts;
    int n = 0;
    Bxdf b[8];
    void init(const SurfaceInteraction &si, Float eta_) {  // reflection.rs:1506-1518
        eta = eta_; ns = si.sh_n; ss = normalize(si.sh_dpdu); ng = si.n; ts = cross(ns, ss); n = 0;
    }
    void add(const Bxdf &x) { b[n++] = x; }
    int num_components(int flags) const { int c = 0; for (int i = 0; i < n; ++i) if (b[i].matches(flags)) ++c; return c; }
    V3 world_to_local(V3 v) const { return V3(dot(v, ss), dot(v, ts), dot(v, ns)); }
    V3 local_to_world(V3 v) const {
        return V3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
    RGB f(V3 wow, V3 wiw, int flags) const;
    Float pdf(V3 wow, V3 wiw, int flags) const;
    RGB sample_f(V3 wow, V3 &wiw, P2 u, Float &pdf, int ty, int &sampled) const;
};

RGB bxdf_f(const Bxdf &b, V3 wo, V3 wi);
Float bxdf_pdf(const Bxdf &b, V3 wo, V3 wi);
RGB bxdf_sample_f(const Bxdf &b, V3 wo, V3 &wi, P2 u, Float &pdf, int &sampled);

// ---- interaction data for light sampling (core/interaction.rs InteractionData) -----------------------
struct IData { V3 p, p_error, n, wo; };
inline Ray spawn_ray(const IData &it, V3 d) {  // interaction.rs:32-36
    return Ray(offset_ray_origin(it.p, it.p_error, it.n, d), d, INF, 0.0f);
}
inline Ray spawn_ray_to(const IData &a, const IData &b) {  // interaction.rs:45-52
    V3 o = offset_ray_origin(a.p, a.p_error, a.n, b.p - a.p);
    V3 t = offset_ray_origin(b.p, b.p_error, b.n, o - b.p);
    return Ray(o, t - o, 1.0f - SHADOW_EPSILON, 0.0f);
}

struct RenderCtx;  // fwd

// ---- light distributions (core/lightdistrib.rs) ------------------------------------------------------
struct LightSampler {
    const Scene *scene = nullptr;
    int strategy = PT_LS_SPATIAL;  // after the `lights.len() == 1 -> uniform` rule (lightdistrib.rs:21)
    std::shared_ptr<Distribution1D> fixed;  // uniform / power
    size_t nvox[3] = {1, 1, 1};
    Float world_radius = 0; V3 world_center;
    Distribution2D env_dist;
    // Per-voxel distributions. The reference keeps them in a lock-free open-addressing hash keyed by
    // the packed voxel coordinates (lightdistrib.rs:249-337); content per voxel is deterministic, so a
    // dense lazily-filled array is result-identical.
    mutable std::vector<std::atomic<const Distribution1D *>> grid;
    mutable std::mutex mu;
    void init(const Scene &s, int requested);
    const Distribution1D *compute_distribution(const int64_t pi[3]) const;
    const Distribution1D *lookup(V3 p) const;
    ~LightSampler() { for (auto &g : grid) delete g.load(); }
    // lights
    RGB sample_li(uint32_t li, const IData &ref, P2 u, V3 &wi, Float &pdf, IData &p1) const;
    Float pdf_li(uint32_t li, const IData &ref, V3 wi) const;
    RGB light_le(uint32_t li, const Ray &r) const;
    RGB area_l(uint32_t li, V3 n, V3 w) const;
    RGB power(uint32_t li) const;
    RGB env_lookup(P2 st) const;
    bool is_delta(uint32_t li) const { uint32_t t = scene->lights[li].type; return t == PT_LIGHT_DISTANT || t == PT_LIGHT_POINT || t == PT_LIGHT_SPOT; }
};

}  // namespace ref
