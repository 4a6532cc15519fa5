// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_math.h header).
// ref_render.cpp: materials -> BSDF, one-light MIS, PathIntegrator::li, perspective camera, film,
// the tile render loop, and the oracle's C ABI (driven from tests/ and bench.py's cpu_baseline leg).
//   materials/{matte,mirror,glass,plastic,metal,uber,substrate}.rs; core/integrator.rs:81-237,263-403;
//   integrators/path.rs:79-222; cameras/perspective.rs:120-179; core/film.rs:104-161,217-258,292-331.
#include "ref_bssrdf.h"
#include "ref_texture.h"
#include "ref_hlbvh.h"
#include <thread>
#include <atomic>
#include <chrono>

namespace ref {

// ---- materials ------------------------------------------------------------------------------
static RGB rgb3(const float *p) { return RGB(p[0], p[1], p[2]); }
static TRDist make_dist(Float ax, Float ay) { TRDist d; d.ax = fmax_(ax, 0.001f); d.ay = fmax_(ay, 0.001f); return d; }  // microfacet.rs:325-331

bool ref::Scene::tri_alpha_rejects(uint32_t tri, const Float b[3], bool shadow) const {
    // isectl = SurfaceInteraction::new(phit, 0, uvhit, -r.d, dpdu, dpdv, ..): no differentials (dudx.. = 0, dpdx = dpdy = 0)
    uint32_t i0 = idx[3 * tri], i1 = idx[3 * tri + 1], i2 = idx[3 * tri + 2];
    P2 uv[3]; tri_uvs(tri, uv);
    TexCtx c; c.dpdx = V3(0, 0, 0); c.dpdy = V3(0, 0, 0);
    c.p = P[i0] * b[0] + P[i1] * b[1] + P[i2] * b[2];
    c.uv = P2(uv[0].x * b[0] + uv[1].x * b[1] + uv[2].x * b[2], uv[0].y * b[0] + uv[1].y * b[1] + uv[2].y * b[2]);
    if (!tri_alpha.empty() && tri_alpha[tri] >= 0 && textures->eval(tri_alpha[tri], c).c[0] == 0.0f) return true;
    if (shadow && !tri_shadow_alpha.empty() && tri_shadow_alpha[tri] >= 0 && textures->eval(tri_shadow_alpha[tri], c).c[0] == 0.0f) return true;
    return false;
}

// Texture::evaluate of a material parameter: a ConstantTexture (the field) unless tex[slot] names a texture node.
struct MatEval {
    const Scene &scene; const PtMaterial &m; const TexCtx *ctx;
    RGB spec(int slot, const float *field) const {
        if (ctx && scene.textures && m.tex[slot] >= 0) return scene.textures->eval(m.tex[slot], *ctx);
        return RGB(field[0], field[1], field[2]);
    }
    Float flt(int slot, Float field) const {
        if (ctx && scene.textures && m.tex[slot] >= 0) return scene.textures->eval(m.tex[slot], *ctx).c[0];
        return field;
    }
};

// bump() (core/material.rs:46-87): displacement texture `d` evaluated at the hit and at two shifted points; the shading
// geometry is rebuilt with set_shading_geometry(.., false) (interaction.rs:228-249).
static void bump_shading(const Scene &scene, int d, const TexCtx &ctx, SurfaceInteraction &si) {
    TexCtx e = ctx;
    Float du = 0.5f * (std::fabs(ctx.dudx) + std::fabs(ctx.dudy));
    if (du == 0.0f) du = 0.0005f;
    e.p = si.p + si.sh_dpdu * du; e.uv = P2(si.uv.x + du, si.uv.y + 0.0f);
    Float udisplace = scene.textures->eval(d, e).c[0];
    Float dv = 0.5f * (std::fabs(ctx.dvdx) + std::fabs(ctx.dvdy));
    if (dv == 0.0f) dv = 0.0005f;
    e.p = si.p + si.sh_dpdv * dv; e.uv = P2(si.uv.x + 0.0f, si.uv.y + dv);
    Float vdisplace = scene.textures->eval(d, e).c[0];
    Float displace = scene.textures->eval(d, ctx).c[0];
    V3 dpdu = si.sh_dpdu + si.sh_n * ((udisplace - displace) / du) + si.sh_dndu * displace;
    V3 dpdv = si.sh_dpdv + si.sh_n * ((vdisplace - displace) / dv) + si.sh_dndv * displace;
    si.sh_n = normalize(cross(dpdu, dpdv));
    if (si.has_shape) {
        if (si.shape_flip) si.sh_n = -si.sh_n;
        si.sh_n = face_forward(si.sh_n, si.n);   // orientation_is_authoritative = false
    }
    si.sh_dpdu = dpdu; si.sh_dpdv = dpdv;
}

// Returns false when the material leaves `si.bsdf == None` (null surface, path.rs:124-129).
static bool material_scattering_functions(const Scene &scene, uint32_t mi, SurfaceInteraction &si, BSDF &bsdf,
                                          TabulatedBSSRDF *bssrdf, bool *has_bssrdf, const TexCtx *tctx);
static bool compute_scattering_functions(const Scene &scene, SurfaceInteraction &si, BSDF &bsdf,
                                         TabulatedBSSRDF *bssrdf = nullptr, bool *has_bssrdf = nullptr, const TexCtx *tctx = nullptr) {
    uint32_t mi = scene.prim_material[si.prim];
    if (mi == PT_NONE) return false;  // primitive.rs:168-170: no material => no bsdf
    const PtMaterial &m = scene.materials[mi];
    if (m.type == PT_MAT_MIX) {   // MixMaterial::compute_scattering_functions (mix.rs:25-50)
        const MatEval E{scene, m, tctx};
        RGB s1 = E.spec(PT_MP_KD, m.kd).clamps(0.0f, INF);   // "amount"
        RGB s2 = (RGB(1.0f) - s1).clamps(0.0f, INF);
        SurfaceInteraction si2 = si;                          // a fresh interaction: same point and uv, no ray differentials;
        TexCtx plain; const TexCtx *tctx2 = nullptr;          // its own shading frame is discarded with its BSDF
        if (tctx) { plain = *tctx; plain.dpdx = plain.dpdy = V3(0.0f, 0.0f, 0.0f); plain.dudx = plain.dvdx = plain.dudy = plain.dvdy = 0.0f; tctx2 = &plain; }
        if (!material_scattering_functions(scene, m.mix[0], si, bsdf, nullptr, nullptr, tctx)) return false;   // the reference unwraps
        BSDF b2;
        bool ok2 = material_scattering_functions(scene, m.mix[1], si2, b2, nullptr, nullptr, tctx2);
        for (int i = 0; i < bsdf.n; ++i) { bsdf.b[i].scaled = true; bsdf.b[i].scale = s1; }
        if (ok2) for (int i = 0; i < b2.n; ++i) { Bxdf x = b2.b[i]; x.scaled = true; x.scale = s2; bsdf.add(x); }
        return true;
    }
    return material_scattering_functions(scene, mi, si, bsdf, bssrdf, has_bssrdf, tctx);
}
static bool material_scattering_functions(const Scene &scene, uint32_t mi, SurfaceInteraction &si, BSDF &bsdf,
                                          TabulatedBSSRDF *bssrdf, bool *has_bssrdf, const TexCtx *tctx) {
    const PtMaterial &m = scene.materials[mi];
    const MatEval E{scene, m, tctx};
    // bump() modifies the caller's interaction in place (shading.n is what path.rs / estimate_direct read afterwards)
    if (tctx && scene.textures && m.tex[PT_MP_BUMP] >= 0) bump_shading(scene, m.tex[PT_MP_BUMP], *tctx, si);   // every material: `if let Some(map) = bumpmap { bump(map, si) }`
    switch (m.type) {
    case PT_MAT_MATTE: {  // matte.rs:28-53
        bsdf.init(si, 1.0f);
        RGB r = E.spec(PT_MP_KD, m.kd).clamps(0.0f, INF);
        Float sig = clampv(E.flt(PT_MP_SIGMA, m.sigma), 0.0f, 90.0f);
        if (!r.is_black()) {
            Bxdf b; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.r = r;
            if (sig == 0.0f) b.kind = BX_LAMBERT_R;
            else {  // reflection.rs:909-921
                b.kind = BX_OREN_NAYAR;
                Float sigma = (PI / 180.0f) * sig;
                Float sigma2 = sigma * sigma;
                b.A = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
                b.B = 0.45f * sigma2 / (sigma2 + 0.09f);
            }
            bsdf.add(b);
        }
        return true;
    }
    case PT_MAT_MIRROR: {  // mirror.rs:23-42
        bsdf.init(si, 1.0f);
        RGB R = E.spec(PT_MP_KR, m.kr).clamps(0.0f, INF);
        if (!R.is_black()) { Bxdf b; b.kind = BX_SPEC_R; b.type = BSDF_REFLECTION | BSDF_SPECULAR; b.r = R; b.fresnel.kind = FR_NOOP; bsdf.add(b); }
        return true;
    }
    case PT_MAT_GLASS: {  // glass.rs:35-92 (allow_multiple_lobes = true, mode = Radiance)
        Float eta = E.flt(PT_MP_ETA, m.eta), urough = E.flt(PT_MP_U_ROUGHNESS, m.u_roughness), vrough = E.flt(PT_MP_V_ROUGHNESS, m.v_roughness);
        RGB R = E.spec(PT_MP_KR, m.kr).clamps(0.0f, INF), T = E.spec(PT_MP_KT, m.kt).clamps(0.0f, INF);
        bsdf.init(si, eta);
        if (R.is_black() && T.is_black()) return false;  // App. A #14
        bool is_specular = urough == 0.0f && vrough == 0.0f;
        if (is_specular) {
            Bxdf b; b.kind = BX_FRESNEL_SPEC; b.type = BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR;
            b.r = R; b.t = T; b.etaa = 1.0f; b.etab = eta; bsdf.add(b);
        } else {
            if (m.remap_roughness) { urough = TRDist::roughness_to_alpha(urough); vrough = TRDist::roughness_to_alpha(vrough); }
            TRDist d = make_dist(urough, vrough);
            if (!R.is_black()) {
                Bxdf b; b.kind = BX_MICRO_R; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.r = R; b.dist = d;
                b.fresnel.kind = FR_DIELECTRIC; b.fresnel.etai = 1.0f; b.fresnel.etat = eta; bsdf.add(b);
            }
            if (!T.is_black()) {
                Bxdf b; b.kind = BX_MICRO_T; b.type = BSDF_TRANSMISSION | BSDF_GLOSSY; b.t = T; b.dist = d; b.etaa = 1.0f; b.etab = eta;
                b.fresnel.kind = FR_DIELECTRIC; b.fresnel.etai = 1.0f; b.fresnel.etat = eta; bsdf.add(b);
            }
        }
        return true;
    }
    case PT_MAT_TRANSLUCENT: {  // translucent.rs:36-78
        const Float eta = 1.5f;
        bsdf.init(si, eta);
        RGB r = E.spec(PT_MP_KR, m.kr).clamps(0.0f, INF), t = E.spec(PT_MP_KT, m.kt).clamps(0.0f, INF);   // "reflect", "transmit"
        if (r.is_black() && t.is_black()) return false;  // App. A #14
        RGB kd = E.spec(PT_MP_KD, m.kd).clamps(0.0f, INF);
        if (!kd.is_black()) {
            if (!r.is_black()) { Bxdf b; b.kind = BX_LAMBERT_R; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.r = r * kd; bsdf.add(b); }
            if (!t.is_black()) { Bxdf b; b.kind = BX_LAMBERT_T; b.type = BSDF_TRANSMISSION | BSDF_DIFFUSE; b.t = t * kd; bsdf.add(b); }
        }
        RGB ks = E.spec(PT_MP_KS, m.ks).clamps(0.0f, INF);
        if (!ks.is_black() && (!r.is_black() || !t.is_black())) {
            Float rough = E.flt(PT_MP_ROUGHNESS, m.roughness);
            if (m.remap_roughness) rough = TRDist::roughness_to_alpha(rough);
            TRDist d = make_dist(rough, rough);
            if (!r.is_black()) {
                Bxdf b; b.kind = BX_MICRO_R; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.r = r * ks; b.dist = d;
                b.fresnel.kind = FR_DIELECTRIC; b.fresnel.etai = 1.0f; b.fresnel.etat = eta; bsdf.add(b);
            }
            if (!t.is_black()) {
                Bxdf b; b.kind = BX_MICRO_T; b.type = BSDF_TRANSMISSION | BSDF_GLOSSY; b.t = t * ks; b.dist = d; b.etaa = 1.0f; b.etab = eta;
                b.fresnel.kind = FR_DIELECTRIC; b.fresnel.etai = 1.0f; b.fresnel.etat = eta; bsdf.add(b);
            }
        }
        return true;
    }
    case PT_MAT_PLASTIC: {  // plastic.rs:34-70
        bsdf.init(si, 1.0f);
        RGB kd = E.spec(PT_MP_KD, m.kd).clamps(0.0f, INF);
        if (!kd.is_black()) { Bxdf b; b.kind = BX_LAMBERT_R; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.r = kd; bsdf.add(b); }
        RGB ks = E.spec(PT_MP_KS, m.ks).clamps(0.0f, INF);
        if (!ks.is_black()) {
            Float rough = E.flt(PT_MP_ROUGHNESS, m.roughness);
            if (m.remap_roughness) rough = TRDist::roughness_to_alpha(rough);
            Bxdf b; b.kind = BX_MICRO_R; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.r = ks; b.dist = make_dist(rough, rough);
            b.fresnel.kind = FR_DIELECTRIC; b.fresnel.etai = 1.5f; b.fresnel.etat = 1.0f; bsdf.add(b);
        }
        return true;
    }
    case PT_MAT_METAL: {  // metal.rs:78-112
        bsdf.init(si, 1.0f);
        // metal.rs:88-96: uroughness / vroughness textures fall back to the `roughness` texture when absent
        Float urough = (m.tex[PT_MP_U_ROUGHNESS] >= 0 || m.u_roughness >= 0.0f) ? E.flt(PT_MP_U_ROUGHNESS, m.u_roughness) : E.flt(PT_MP_ROUGHNESS, m.roughness);
        Float vrough = (m.tex[PT_MP_V_ROUGHNESS] >= 0 || m.v_roughness >= 0.0f) ? E.flt(PT_MP_V_ROUGHNESS, m.v_roughness) : E.flt(PT_MP_ROUGHNESS, m.roughness);
        if (m.remap_roughness) { urough = TRDist::roughness_to_alpha(urough); vrough = TRDist::roughness_to_alpha(vrough); }
        Bxdf b; b.kind = BX_MICRO_R; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.r = RGB(1.0f); b.dist = make_dist(urough, vrough);
        b.fresnel.kind = FR_CONDUCTOR; b.fresnel.ci = RGB(1.0f); b.fresnel.ct = E.spec(PT_MP_ETA_RGB, m.eta_rgb); b.fresnel.k = E.spec(PT_MP_K_RGB, m.k_rgb);
        bsdf.add(b);
        return true;
    }
    case PT_MAT_UBER: {  // uber.rs:40-106
        Float e = E.flt(PT_MP_ETA, m.eta);
        RGB op = E.spec(PT_MP_OPACITY, m.opacity).clamps(0.0f, INF);
        RGB t = RGB(-op.c[0] + 1.0f, -op.c[1] + 1.0f, -op.c[2] + 1.0f).clamps(0.0f, INF);
        if (!t.is_black()) {
            bsdf.init(si, 1.0f);
            Bxdf b; b.kind = BX_SPEC_T; b.type = BSDF_TRANSMISSION | BSDF_SPECULAR; b.t = t; b.etaa = 1.0f; b.etab = 1.0f;
            b.fresnel.kind = FR_DIELECTRIC; b.fresnel.etai = 1.0f; b.fresnel.etat = 1.0f; bsdf.add(b);
        } else bsdf.init(si, e);
        RGB kd = op * E.spec(PT_MP_KD, m.kd).clamps(0.0f, INF);
        if (!kd.is_black()) { Bxdf b; b.kind = BX_LAMBERT_R; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.r = kd; bsdf.add(b); }
        RGB ks = op * E.spec(PT_MP_KS, m.ks).clamps(0.0f, INF);
        if (!ks.is_black()) {
            Float ru = (m.tex[PT_MP_U_ROUGHNESS] >= 0 || m.u_roughness >= 0.0f) ? E.flt(PT_MP_U_ROUGHNESS, m.u_roughness) : E.flt(PT_MP_ROUGHNESS, m.roughness);
            Float rv = (m.tex[PT_MP_V_ROUGHNESS] >= 0 || m.v_roughness >= 0.0f) ? E.flt(PT_MP_V_ROUGHNESS, m.v_roughness) : E.flt(PT_MP_ROUGHNESS, m.roughness);
            if (m.remap_roughness) { ru = TRDist::roughness_to_alpha(ru); rv = TRDist::roughness_to_alpha(rv); }
            Bxdf b; b.kind = BX_MICRO_R; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.r = ks; b.dist = make_dist(ru, rv);
            b.fresnel.kind = FR_DIELECTRIC; b.fresnel.etai = 1.0f; b.fresnel.etat = e; bsdf.add(b);
        }
        RGB kr = op * E.spec(PT_MP_KR, m.kr).clamps(0.0f, INF);
        if (!kr.is_black()) {
            Bxdf b; b.kind = BX_SPEC_R; b.type = BSDF_REFLECTION | BSDF_SPECULAR; b.r = kr;
            b.fresnel.kind = FR_DIELECTRIC; b.fresnel.etai = 1.0f; b.fresnel.etat = e; bsdf.add(b);
        }
        RGB kt = op * E.spec(PT_MP_KT, m.kt).clamps(0.0f, INF);
        if (!kt.is_black()) {
            Bxdf b; b.kind = BX_SPEC_T; b.type = BSDF_TRANSMISSION | BSDF_SPECULAR; b.t = kt; b.etaa = 1.0f; b.etab = e;
            b.fresnel.kind = FR_DIELECTRIC; b.fresnel.etai = 1.0f; b.fresnel.etat = e; bsdf.add(b);
        }
        return true;
    }
    case PT_MAT_DISNEY: {  // disney.rs:719-840; "scatterdistance" is zero here (the BSSRDF branch is refused at scene creation)
        bsdf.init(si, 1.0f);
        RGB c = E.spec(PT_MP_KD, m.kd).clamps(0.0f, INF);   // "color"
        Float mweight = m.disney[PT_DS_METALLIC], e = E.flt(PT_MP_ETA, m.eta), strans = m.disney[PT_DS_SPECTRANS];
        Float dweight = (1.0f - mweight) * (1.0f - strans);
        Float dt = m.disney[PT_DS_DIFFTRANS] / 2.0f;
        Float rough = E.flt(PT_MP_ROUGHNESS, m.roughness);
        Float lum = 0.212671f * c.c[0] + 0.715160f * c.c[1] + 0.072169f * c.c[2];
        RGB ctint = lum > 0.0f ? c / lum : RGB(1.0f);
        Float sheen_weight = m.disney[PT_DS_SHEEN];
        RGB csheen(0.0f);
        if (sheen_weight > 0.0f) csheen = lerp_t(m.disney[PT_DS_SHEENTINT], RGB(1.0f), ctint);
        bool thin = m.disney_thin != 0;
        if (dweight > 0.0f) {
            if (thin) {
                Float flat = m.disney[PT_DS_FLATNESS];
                { Bxdf b; b.kind = BX_DISNEY_DIFFUSE; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.r = c * dweight * flat * (1.0f - dt); bsdf.add(b); }
                { Bxdf b; b.kind = BX_DISNEY_FAKESS; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.r = c * (1.0f - dt) * flat * dweight; b.A = rough; bsdf.add(b); }
            } else {
                RGB sd(m.disney_scatter[0], m.disney_scatter[1], m.disney_scatter[2]);
                if (sd.is_black()) { Bxdf b; b.kind = BX_DISNEY_DIFFUSE; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.r = c * dweight; bsdf.add(b); }
                else {   // a BSSRDF instead (disney.rs:768-776)
                    Bxdf b; b.kind = BX_SPEC_T; b.type = BSDF_TRANSMISSION | BSDF_SPECULAR; b.t = RGB(1.0f); b.etaa = 1.0f; b.etab = e;
                    b.fresnel.kind = FR_DIELECTRIC; b.fresnel.etai = 1.0f; b.fresnel.etat = e; bsdf.add(b);
                    if (bssrdf) { bssrdf->init_disney(si, mi, e, c * dweight, sd); *has_bssrdf = true; }
                }
            }
            { Bxdf b; b.kind = BX_DISNEY_RETRO; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.r = c * dweight; b.A = rough; bsdf.add(b); }
            if (sheen_weight > 0.0f) { Bxdf b; b.kind = BX_DISNEY_SHEEN; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.r = csheen * sheen_weight * dweight; bsdf.add(b); }
        }
        Float aspect = std::sqrt(1.0f - m.disney[PT_DS_ANISOTROPIC] * 0.9f);
        TRDist dis; dis.ax = fmax_(rough * rough / aspect, 0.001f); dis.ay = fmax_(rough * rough * aspect, 0.001f); dis.separable = true;
        RGB cspec0 = lerp_t(mweight, lerp_t(m.disney[PT_DS_SPECULARTINT], RGB(1.0f) * (((e - 1.0f) * (e - 1.0f)) / ((e + 1.0f) * (e + 1.0f))), ctint), c);
        {
            Bxdf b; b.kind = BX_MICRO_R; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.r = RGB(1.0f); b.dist = dis;
            b.fresnel.kind = FR_DISNEY; b.fresnel.r0 = cspec0; b.fresnel.metallic = mweight; b.fresnel.etat = e; bsdf.add(b);
        }
        Float cc = m.disney[PT_DS_CLEARCOAT];
        if (cc > 0.0f) { Bxdf b; b.kind = BX_DISNEY_CLEARCOAT; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.B = cc; b.A = lerp_t(m.disney[PT_DS_CLEARCOATGLOSS], 0.1f, 0.001f); bsdf.add(b); }
        if (strans > 0.0f) {
            RGB T = sqrt_rgb(c) * strans;
            Bxdf b; b.kind = BX_MICRO_T; b.type = BSDF_TRANSMISSION | BSDF_GLOSSY; b.t = T; b.etaa = 1.0f; b.etab = e;
            b.fresnel.kind = FR_DIELECTRIC; b.fresnel.etai = 1.0f; b.fresnel.etat = e;
            if (thin) {
                Float rscaled = (0.65f * e - 0.35f) * rough;
                b.dist.ax = fmax_(rscaled * rscaled / aspect, 0.001f); b.dist.ay = fmax_(rscaled * rscaled * aspect, 0.001f);
            } else b.dist = dis;
            bsdf.add(b);
        }
        if (thin) { Bxdf b; b.kind = BX_LAMBERT_T; b.type = BSDF_TRANSMISSION | BSDF_DIFFUSE; b.t = c * dt; bsdf.add(b); }
        return true;
    }
    case PT_MAT_SUBSTRATE: {  // substrate.rs:34-60
        bsdf.init(si, 1.0f);
        RGB d = E.spec(PT_MP_KD, m.kd).clamps(0.0f, INF), s = E.spec(PT_MP_KS, m.ks).clamps(0.0f, INF);
        Float ru = E.flt(PT_MP_U_ROUGHNESS, m.u_roughness), rv = E.flt(PT_MP_V_ROUGHNESS, m.v_roughness);
        if (!d.is_black() || !s.is_black()) {
            if (m.remap_roughness) { ru = TRDist::roughness_to_alpha(ru); rv = TRDist::roughness_to_alpha(rv); }
            Bxdf b; b.kind = BX_FRESNEL_BLEND; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.r = d; b.rs = s; b.dist = make_dist(ru, rv);
            bsdf.add(b);
            return true;
        }
        return false;  // App. A #14
    }
    case PT_MAT_SUBSURFACE: {  // subsurface.rs:47-106 (kdsubsurface.rs:45-103 after the host-side subsurface_from_diffuse)
        Float eta = E.flt(PT_MP_ETA, m.eta), urough = E.flt(PT_MP_U_ROUGHNESS, m.u_roughness), vrough = E.flt(PT_MP_V_ROUGHNESS, m.v_roughness);
        RGB R = E.spec(PT_MP_KR, m.kr).clamps(0.0f, INF), T = E.spec(PT_MP_KT, m.kt).clamps(0.0f, INF);
        bsdf.init(si, eta);
        if (R.is_black() && T.is_black()) return false;
        bool is_specular = urough == 0.0f && vrough == 0.0f;
        if (is_specular) {
            Bxdf b; b.kind = BX_FRESNEL_SPEC; b.type = BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR;
            b.r = R; b.t = T; b.etaa = 1.0f; b.etab = eta; bsdf.add(b);
        } else {
            if (m.remap_roughness) { urough = TRDist::roughness_to_alpha(urough); vrough = TRDist::roughness_to_alpha(vrough); }
            TRDist d = make_dist(urough, vrough);
            if (!R.is_black()) {
                Bxdf b; b.kind = BX_MICRO_R; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.r = R; b.dist = d;
                b.fresnel.kind = FR_DIELECTRIC; b.fresnel.etai = 1.0f; b.fresnel.etat = eta; bsdf.add(b);
            }
            if (!T.is_black()) {
                Bxdf b; b.kind = BX_MICRO_T; b.type = BSDF_TRANSMISSION | BSDF_GLOSSY; b.t = T; b.dist = d; b.etaa = 1.0f; b.etab = eta;
                b.fresnel.kind = FR_DIELECTRIC; b.fresnel.etai = 1.0f; b.fresnel.etat = eta; bsdf.add(b);
            }
        }
        if (bssrdf) {
            RGB siga = E.spec(PT_MP_SIGMA_A, m.sigma_a).clamps(0.0f, INF) * m.scale, sigs = E.spec(PT_MP_SIGMA_S, m.sigma_s).clamps(0.0f, INF) * m.scale;
            if (m.kd_subsurface) {   // kdsubsurface.rs:96-99: textured Kd / mfp, converted at this hit
                RGB mfree = E.spec(PT_MP_MFP, m.mfp).clamps(0.0f, INF) * m.scale, kd = E.spec(PT_MP_KD, m.kd).clamps(0.0f, INF).clamps(0.0f, INF);
                subsurface_from_diffuse(scene.bssrdf_tables[m.bssrdf_table], kd, mfree, siga, sigs);
            }
            bssrdf->init(si, mi, eta, siga, sigs, &scene.bssrdf_tables[m.bssrdf_table]);
            *has_bssrdf = true;
        }
        return true;
    }
    }
    return false;
}

// ---- direct lighting (core/integrator.rs:81-237) ---------------------------------------------------
struct RenderCtx {
    const Scene *scene;
    const LightSampler *lights;
    Counters *c;
};

static RGB isect_le(const RenderCtx &ctx, const SurfaceInteraction &si, V3 w) {  // interaction.rs:344-349
    uint32_t li = ctx.scene->prim_light[si.prim];
    if (li == PT_NONE) return RGB(0.0f);
    return ctx.lights->area_l(li, si.n, w);
}

static RGB estimate_direct(const RenderCtx &ctx, const SurfaceInteraction &si, const BSDF &bsdf, P2 uscatt, uint32_t li, P2 ulight) {
    const int flags = BSDF_ALL & ~BSDF_SPECULAR;
    RGB Ld(0.0f);
    IData it; it.p = si.p; it.p_error = si.p_error; it.n = si.n; it.wo = si.wo;
    V3 wi; Float lightpdf = 0.0f, scattpdf = 0.0f; IData p1;
    RGB Li = ctx.lights->sample_li(li, it, ulight, wi, lightpdf, p1);
    bool delta = ctx.lights->is_delta(li);
    if (lightpdf > 0.0f && !Li.is_black()) {
        RGB f = bsdf.f(si.wo, wi, flags) * abs_dot(wi, si.sh_n);
        scattpdf = bsdf.pdf(si.wo, wi, flags);
        if (!f.is_black()) {
            Ray sr = spawn_ray_to(it, p1);  // VisibilityTester::unoccluded (light.rs:120-123)
            if (ctx.scene->intersect_p(sr, *ctx.c)) Li = RGB(0.0f);
            if (!Li.is_black()) {
                if (delta) Ld += f * Li / lightpdf;
                else {
                    Float weight = power_heuristic(1, lightpdf, 1, scattpdf);
                    Ld += f * Li * weight / lightpdf;
                }
            }
        }
    }
    if (!delta) {
        int sampled_type = 0;
        RGB f = bsdf.sample_f(si.wo, wi, uscatt, scattpdf, flags, sampled_type);
        f = f * abs_dot(wi, si.sh_n);
        bool sampled_specular = (sampled_type & BSDF_SPECULAR) != 0;
        if (!f.is_black() && scattpdf > 0.0f) {
            Float weight = 1.0f;
            if (!sampled_specular) {
                lightpdf = ctx.lights->pdf_li(li, it, wi);
                if (lightpdf == 0.0f) return Ld;
                weight = power_heuristic(1, scattpdf, 1, lightpdf);
            }
            SurfaceInteraction lisect;
            Ray ray = spawn_ray(it, wi);
            bool found = ctx.scene->intersect(ray, lisect, *ctx.c);
            RGB li_(0.0f);
            if (found) {
                if (ctx.scene->prim_light[lisect.prim] == li) li_ = isect_le(ctx, lisect, -wi);
            } else li_ = ctx.lights->light_le(li, ray);
            if (!li_.is_black()) Ld += f * li_ * RGB(1.0f) * weight / scattpdf;
        }
    }
    return Ld;
}

static RGB uniform_sample_onelight(const RenderCtx &ctx, const SurfaceInteraction &si, const BSDF &bsdf, SobolSampler &sampler,
                                   const Distribution1D *distrib) {
    size_t nlights = ctx.scene->lights.size();
    if (nlights == 0) return RGB(0.0f);
    Float lightpdf = 0.0f;
    size_t lightnum = distrib->sample_discrete(sampler.get_1d(), &lightpdf);
    if (lightpdf == 0.0f) return RGB(0.0f);
    P2 ulight = sampler.get_2d();
    P2 uscatt = sampler.get_2d();
    return estimate_direct(ctx, si, bsdf, uscatt, (uint32_t)lightnum, ulight) / lightpdf;
}

// ---- PathIntegrator::li (integrators/path.rs:79-222) ------------------------------------------------
struct PathParams { uint32_t max_depth; Float rr_threshold; };

static RGB path_li(const RenderCtx &ctx, const PathParams &pp, Ray ray, SobolSampler &sampler, RayDiff rdiff = RayDiff()) {
    RGB L(0.0f), beta(1.0f);
    bool specular_bounce = false;
    uint32_t bounces = 0;
    Float etascale = 1.0f;
    for (;;) {
        SurfaceInteraction isect;
        bool found = ctx.scene->intersect(ray, isect, *ctx.c);
        if (bounces == 0 || specular_bounce) {
            if (found) L += isect_le(ctx, isect, -ray.d) * beta;
            else for (uint32_t li : ctx.scene->infinite_lights) L += ctx.lights->light_le(li, ray) * beta;
        }
        if (!found || bounces >= pp.max_depth) break;
        BSDF bsdf;
        TabulatedBSSRDF bssrdf; bool has_bssrdf = false;
        // SurfaceInteraction::compute_scattering_functions (interaction.rs:262-267): differentials of THIS ray first;
        // only the camera ray carries them (every spawn_ray below creates a ray without differentials)
        TexCtx tctx;
        const bool textured = (bool)ctx.scene->textures;
        if (textured) tctx = compute_differentials(isect, rdiff);
        rdiff.has = false;
        if (!compute_scattering_functions(*ctx.scene, isect, bsdf, &bssrdf, &has_bssrdf, textured ? &tctx : nullptr)) {
            IData it; it.p = isect.p; it.p_error = isect.p_error; it.n = isect.n;
            ray = spawn_ray(it, ray.d);
            continue;
        }
        const Distribution1D *distrib = ctx.lights->lookup(isect.p);
        if (bsdf.num_components(BSDF_ALL & ~BSDF_SPECULAR) > 0) {
            ctx.c->zero_den++;
            RGB Ld = beta * uniform_sample_onelight(ctx, isect, bsdf, sampler, distrib);
            if (Ld.is_black()) ctx.c->zero_num++;
            if (!(Ld.y() >= 0.0f)) ctx.c->ref_asserts++;   // path.rs:143 assert!(Ld.y() >= 0.0)
            L += Ld;
        }
        V3 wo = -ray.d, wi;
        Float pdf = 0.0f; int flags = 0;
        RGB f = bsdf.sample_f(wo, wi, sampler.get_2d(), pdf, BSDF_ALL, flags);
        if (f.is_black() || pdf == 0.0f) break;
        beta *= f * abs_dot(wi, isect.sh_n) / pdf;
        if (!(beta.y() >= 0.0f)) ctx.c->ref_asserts++;       // path.rs:162
        if (std::isinf(beta.y())) ctx.c->ref_asserts++;      // path.rs:163
        specular_bounce = (flags & BSDF_SPECULAR) != 0;
        if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
            Float eta = bsdf.eta;
            etascale *= (dot(wo, isect.n) > 0.0f) ? eta * eta : 1.0f / (eta * eta);
        }
        IData it; it.p = isect.p; it.p_error = isect.p_error; it.n = isect.n;
        ray = spawn_ray(it, wi);
        if (has_bssrdf && (flags & BSDF_TRANSMISSION)) {  // path.rs:177-204
            P2 s2 = sampler.get_2d();
            Float s1 = sampler.get_1d();
            if (std::isinf(beta.y())) ctx.c->ref_asserts++;  // path.rs:184 (evaluated after sample_s whatever it returned)
            // TabulatedBSSRDF::sample_s -> sample_sp (bssrdf.rs:334-410)
            V3 start, target; Float u1n = 0.0f;
            if (!bssrdf.probe_segment(s1, s2, start, target, u1n)) break;   // S black
            IData base; base.p = start; base.p_error = V3(0, 0, 0); base.n = V3(0, 0, 0);
            std::vector<SurfaceInteraction> chain;
            for (;;) {
                // spawn_rayto_point (interaction.rs:38-43): d = p2 - p measured from the un-offset point
                V3 d = target - base.p;
                Ray r(offset_ray_origin(base.p, base.p_error, base.n, d), d, 1.0f - SHADOW_EPSILON, 0.0f);
                SurfaceInteraction si2;
                if ((d.x == 0.0f && d.y == 0.0f && d.z == 0.0f) || !ctx.scene->intersect(r, si2, *ctx.c)) break;
                base.p = si2.p; base.p_error = si2.p_error; base.n = si2.n;
                if (ctx.scene->prim_material[si2.prim] == bssrdf.material) chain.push_back(si2);
            }
            size_t nfound = chain.size();
            if (nfound == 0) break;
            size_t selected = (size_t)clampv((int64_t)(u1n * (Float)nfound), (int64_t)0, (int64_t)nfound - 1);
            SurfaceInteraction pi = chain[selected];
            pdf = bssrdf.pdf_sp(pi.p, pi.n) / (Float)nfound;
            RGB S = bssrdf.sr(length(bssrdf.po_p - pi.p));
            if (S.is_black() || pdf == 0.0f) break;
            // sample_s (bssrdf.rs:559-574): BSDF::new(pi, 1.0) + SeparableBSSRDFAdapter, pi.wo = shading.n
            BSDF pibsdf; pibsdf.init(pi, 1.0f);
            { Bxdf b; b.kind = BX_BSSRDF; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.etab = bssrdf.eta; pibsdf.add(b); }
            pi.wo = pi.sh_n;
            beta *= S / pdf;
            const Distribution1D *d2 = ctx.lights->lookup(pi.p);
            L += beta * uniform_sample_onelight(ctx, pi, pibsdf, sampler, d2);
            RGB ff = pibsdf.sample_f(pi.wo, wi, sampler.get_2d(), pdf, BSDF_ALL, flags);
            if (ff.is_black() || pdf == 0.0f) break;
            beta *= ff * abs_dot(wi, pi.sh_n) / pdf;
            if (std::isinf(beta.y())) ctx.c->ref_asserts++;  // path.rs:201
            specular_bounce = (flags & BSDF_SPECULAR) != 0;
            IData pit; pit.p = pi.p; pit.p_error = pi.p_error; pit.n = pi.n;
            ray = spawn_ray(pit, wi);
        }
        RGB rrbeta = beta * etascale;
        if (rrbeta.max_component_value() < pp.rr_threshold && bounces > 3) {
            Float q = fmax_(1.0f - rrbeta.max_component_value(), 0.05f);
            if (sampler.get_1d() < q) break;
            beta = beta / (1.0f - q);
            if (std::isinf(beta.y())) ctx.c->ref_asserts++;  // path.rs:213 / volpath.rs:223
        }
        bounces += 1;
    }
    ctx.c->path_len[std::min<uint32_t>(bounces, 15)]++;
    return L;
}


// ---- VolPathIntegrator (integrators/volpath.rs:76-186) with HomogeneousMedium (media/homogeneous.rs) and the Henyey-Greenstein
//      phase function (core/medium.rs:149-194). A primitive without a material is a medium-interface shell (api.rs:597): the path
//      steps over it with `bounces -= 1; continue` (volpath.rs:152-156: the increment at the loop's end is skipped, so the count drops
//      and, at 0, wraps -- the release build has no overflow checks), and the transmittance loops of VisibilityTester::tr
//      (light.rs:125-150) and Scene::intersect_tr (scene.rs:68-87) walk on behind it.
static inline Float dm_expf_(Float x) { return (Float)dm_expd((double)x); }   // f32::exp through the shared f64 exp
static RGB medium_tr(const PtMedium &m, Float t_max, V3 d) {   // homogeneous.rs:32-35
    Float l = fmin_(t_max * length(d), std::numeric_limits<Float>::max());
    RGB r;
    for (int i = 0; i < 3; ++i) r.c[i] = dm_expf_(-(m.sigma_a[i] + m.sigma_s[i]) * l);
    return r;
}
struct MediumVertex { bool valid = false; V3 p, wo; uint32_t medium = PT_NONE; Float g = 0; };

// ---- GridDensityMedium (media/grid.rs) -----------------------------------------------------------------------------------
static Float grid_d(const PtMedium &m, const Scene::GridAux &g, int64_t x, int64_t y, int64_t z) {   // grid.rs:102-110
    if (x < 0 || y < 0 || z < 0 || x >= (int64_t)m.nx || y >= (int64_t)m.ny || z >= (int64_t)m.nz) return 0.0f;
    return g.density[((size_t)z * m.ny + (size_t)y) * m.nx + (size_t)x];
}
static Float grid_density(const PtMedium &m, const Scene::GridAux &g, V3 p) {   // grid.rs:77-100
    const Float sx = p.x * (Float)m.nx - 0.5f, sy = p.y * (Float)m.ny - 0.5f, sz = p.z * (Float)m.nz - 0.5f;
    const int64_t ix = f2i_sat(sx), iy = f2i_sat(sy), iz = f2i_sat(sz);   // Point3i::from(Point3f) is `x as isize` (point.rs:618-626): truncation towards zero, not floor
    const Float dx = sx - (Float)ix, dy = sy - (Float)iy, dz = sz - (Float)iz;
    auto lerp_ = [](Float t, Float a, Float b) { return a * (1.0f - t) + b * t; };   // pbrt.rs:136-144
    const Float d00 = lerp_(dx, grid_d(m, g, ix, iy, iz), grid_d(m, g, ix + 1, iy, iz));
    const Float d10 = lerp_(dx, grid_d(m, g, ix, iy + 1, iz), grid_d(m, g, ix + 1, iy + 1, iz));
    const Float d01 = lerp_(dx, grid_d(m, g, ix, iy, iz + 1), grid_d(m, g, ix + 1, iy, iz + 1));
    const Float d11 = lerp_(dx, grid_d(m, g, ix, iy + 1, iz + 1), grid_d(m, g, ix + 1, iy + 1, iz + 1));
    return lerp_(dz, lerp_(dy, d00, d10), lerp_(dy, d01, d11));
}
// The ray of grid.rs:115-117 / :155-158 in medium space and its overlap [tmin, tmax] with the unit cube (Bounds3f::intersect_p,
// bounds.rs:533-557). Returns false when the ray misses the medium's bounds.
static bool grid_ray(const PtMedium &m, const Ray &ray, Ray &r, Float &tmin, Float &tmax) {
    M4 w2m = m4_from(m.world_to_medium);
    r = xf_ray(w2m, Ray(ray.o, normalize(ray.d), ray.t_max * length(ray.d), 0.0f));
    Float t0 = 0.0f, t1 = r.t_max;
    const Float o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
    for (int i = 0; i < 3; ++i) {
        const Float inv = 1.0f / d[i];
        Float tnear = (0.0f - o[i]) * inv, tfar = (1.0f - o[i]) * inv;
        if (tnear > tfar) std::swap(tnear, tfar);
        tfar *= 1.0f + 2.0f * gamma(3);
        t0 = tnear > t0 ? tnear : t0;
        t1 = tfar < t1 ? tfar : t1;
        if (t0 > t1) return false;
    }
    tmin = t0; tmax = t1;
    return true;
}
template <class S> static RGB grid_tr(const PtMedium &m, const Scene::GridAux &g, const Ray &ray, S &sampler) {   // grid.rs:113-147: ratio tracking
    Ray r; Float tmin, tmax;
    if (!grid_ray(m, ray, r, tmin, tmax)) return RGB(1.0f);
    Float tr = 1.0f, t = tmin;
    for (;;) {
        t -= dm_logf(1.0f - sampler.get_1d()) * g.inv_max_density / g.sigma_t;
        if (t >= tmax) break;
        if (sampler.dim_overflow) return RGB(0.0f);   // past the sampler's last dimension the reference panics; the render reports the error
        const Float density = grid_density(m, g, r.o + r.d * t);
        tr *= 1.0f - fmax_(density * g.inv_max_density, 0.0f);
        const Float rr_threshold = 0.1f;   // grid.rs:136-143: roulette on low transmittance
        if (tr < rr_threshold) {
            const Float q = fmax_(1.0f - tr, 0.05f);
            if (sampler.get_1d() < q) return RGB(0.0f);
            tr /= 1.0f - q;
        }
    }
    return RGB(tr);
}
template <class S> static RGB grid_sample(const PtMedium &m, const Scene::GridAux &g, uint32_t mid, const Ray &ray, S &sampler, MediumVertex &mi) {   // grid.rs:149-182: delta tracking
    Ray r; Float tmin, tmax;
    if (!grid_ray(m, ray, r, tmin, tmax)) return RGB(1.0f);
    Float t = tmin;
    for (;;) {
        t -= dm_logf(1.0f - sampler.get_1d()) * g.inv_max_density / g.sigma_t;
        if (t >= tmax) break;
        if (sampler.dim_overflow) break;
        if (grid_density(m, g, r.o + r.d * t) * g.inv_max_density > sampler.get_1d()) {
            // (the reference evaluates `ray.find_point(t)` with the medium-space parameter on the WORLD ray, as written in grid.rs:173)
            mi.valid = true; mi.p = ray.o + ray.d * t; mi.wo = -ray.d; mi.medium = mid; mi.g = m.g;
            return RGB(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]) / g.sigma_t;
        }
    }
    return RGB(1.0f);
}

static RGB medium_sample(const PtMedium &m, uint32_t mid, const Ray &ray, SobolSampler &sampler, MediumVertex &mi) {   // homogeneous.rs:37-68
    Float sigma_t[3] = {m.sigma_a[0] + m.sigma_s[0], m.sigma_a[1] + m.sigma_s[1], m.sigma_a[2] + m.sigma_s[2]};
    Float uc = sampler.get_1d() * 3.0f;
    size_t channel = std::min<size_t>(uc > 0.0f ? (size_t)uc : 0, 2);
    Float dist = -dm_logf(1.0f - sampler.get_1d()) / sigma_t[channel];
    Float dl = length(ray.d);
    Float t = fmin_(dist / dl, ray.t_max);
    bool sampled = t < ray.t_max;
    if (sampled) { mi.valid = true; mi.p = ray.o + ray.d * t; mi.wo = -ray.d; mi.medium = mid; mi.g = m.g; }
    RGB Tr, density;
    Float pdf = 0.0f;
    for (int i = 0; i < 3; ++i) {
        Tr.c[i] = dm_expf_(-sigma_t[i] * fmin_(t, std::numeric_limits<Float>::max()) * dl);
        density.c[i] = sampled ? sigma_t[i] * Tr.c[i] : Tr.c[i];
        pdf += density.c[i];
    }
    pdf *= 1.0f / 3.0f;
    if (pdf == 0.0f) pdf = 1.0f;
    RGB ss(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]);
    return sampled ? Tr * ss / pdf : Tr / pdf;
}
static inline Float phase_hg(Float cos_theta, Float g) {   // medium.rs:149-154
    Float denom = 1.0f + g * g + 2.0f * g * cos_theta;
    return INV4_PI * (1.0f - g * g) / (denom * std::sqrt(denom));
}
static Float hg_sample_p(Float g, V3 wo, V3 &wi, P2 u) {   // medium.rs:173-193
    Float cos_theta;
    if (std::fabs(g) < 1.0e-3f) cos_theta = 1.0f - 2.0f * u.x;
    else {
        Float sqr_term = (1.0f - g * g) / (1.0f + g - 2.0f * g * u.x);
        cos_theta = -(1.0f + g * g - sqr_term * sqr_term) / (2.0f * g);
    }
    Float sin_theta = std::sqrt(fmax_(1.0f - cos_theta * cos_theta, 0.0f));
    Float phi = 2.0f * PI * u.y;
    V3 v1, v2; coordinate_system(wo, v1, v2);
    wi = v1 * sin_theta * dm_cosf(phi) + v2 * sin_theta * dm_sinf(phi) + wo * cos_theta;   // spherical_direction_basis (geometry.rs:36-38)
    return phase_hg(cos_theta, g);
}
// Medium::tr / Medium::sample by medium type
template <class S> static RGB medium_tr_any(const Scene &sc, uint32_t mid, const Ray &ray, S &sampler) {
    const PtMedium &m = sc.media[mid];
    return m.type == PT_MEDIUM_GRID ? grid_tr(m, sc.grid_aux[mid], ray, sampler) : medium_tr(m, ray.t_max, ray.d);
}
template <class S> static RGB medium_sample_any(const Scene &sc, uint32_t mid, const Ray &ray, S &sampler, MediumVertex &mi) {
    const PtMedium &m = sc.media[mid];
    return m.type == PT_MEDIUM_GRID ? grid_sample(m, sc.grid_aux[mid], mid, ray, sampler, mi) : medium_sample(m, mid, ray, sampler, mi);
}
// the interaction's MediumInterface: the primitive's own when it is a transition, else the ray's medium on both sides
// (primitive.rs:139-145); get_medium_vec (interaction.rs:54-66)
struct MedIface { uint32_t inside = PT_NONE, outside = PT_NONE; };
static MedIface surface_iface(const Scene &s, uint32_t prim, uint32_t ray_medium) {
    MedIface m; m.inside = m.outside = ray_medium;
    if (!s.prim_med_in.empty() && s.prim_med_in[prim] != s.prim_med_out[prim]) { m.inside = s.prim_med_in[prim]; m.outside = s.prim_med_out[prim]; }
    return m;
}
static inline uint32_t medium_toward(const MedIface &m, V3 n, V3 w) { return dot(w, n) > 0.0f ? m.outside : m.inside; }

// estimate_direct with handle_media = true (integrator.rs:109-237) for a surface (bsdf != nullptr) or a medium vertex
static RGB vol_estimate_direct(const RenderCtx &ctx, const IData &it, const MedIface &mif, const SurfaceInteraction *si, const BSDF *bsdf, Float g,
                               P2 uscatt, uint32_t li, P2 ulight, SobolSampler &sampler) {   // the sampler: grid media draw their tracking steps from it (grid.rs)
    const Scene &S = *ctx.scene;
    const int flags = BSDF_ALL & ~BSDF_SPECULAR;
    RGB Ld(0.0f);
    V3 wi; Float lightpdf = 0.0f, scattpdf = 0.0f; IData p1;
    RGB Li = ctx.lights->sample_li(li, it, ulight, wi, lightpdf, p1);
    bool delta = ctx.lights->is_delta(li);
    if (lightpdf > 0.0f && !Li.is_black()) {
        RGB f;
        if (bsdf) { f = bsdf->f(si->wo, wi, flags) * abs_dot(wi, si->sh_n); scattpdf = bsdf->pdf(si->wo, wi, flags); }
        else { Float p = phase_hg(dot(it.wo, wi), g); f = RGB(p); scattpdf = p; }
        if (!f.is_black()) {
            // VisibilityTester::tr (light.rs:125-150): a loop over the segments between material-less surfaces (medium-interface shells)
            Ray sr = spawn_ray_to(it, p1);
            sr.medium = medium_toward(mif, it.n, sr.d);
            RGB Tr(1.0f);
            for (;;) {
                SurfaceInteraction tmp;
                const bool hitt = S.intersect(sr, tmp, *ctx.c);
                if (hitt && S.prim_material[tmp.prim] != PT_NONE) { Tr = RGB(0.0f); break; }   // an opaque surface along the ray's path
                if (sr.medium != PT_NONE) Tr = Tr * medium_tr_any(S, sr.medium, sr, sampler);   // the current segment (ray.t_max = the hit)
                if (!hitt) break;
                IData a; a.p = tmp.p; a.p_error = tmp.p_error; a.n = tmp.n;
                const uint32_t seg_medium = sr.medium;
                sr = spawn_ray_to(a, p1);                                                     // isect.spawn_rayto_interaction(&self.p1)
                sr.medium = medium_toward(surface_iface(S, tmp.prim, seg_medium), tmp.n, sr.d);   // get_medium(d) of the shell's interface (interaction.rs:54-66)
            }
            Li = Li * Tr;
            if (!Li.is_black()) {
                if (delta) Ld += f * Li / lightpdf;
                else { Float weight = power_heuristic(1, lightpdf, 1, scattpdf); Ld += f * Li * weight / lightpdf; }
            }
        }
    }
    if (!delta) {
        RGB f; bool sampled_specular = false;
        if (bsdf) {
            int sampled_type = 0;
            f = bsdf->sample_f(si->wo, wi, uscatt, scattpdf, flags, sampled_type);
            f = f * abs_dot(wi, si->sh_n);
            sampled_specular = (sampled_type & BSDF_SPECULAR) != 0;
        } else { Float p = hg_sample_p(g, it.wo, wi, uscatt); f = RGB(p); scattpdf = p; }
        if (!f.is_black() && scattpdf > 0.0f) {
            Float weight = 1.0f;
            if (!sampled_specular) {
                lightpdf = ctx.lights->pdf_li(li, it, wi);
                if (lightpdf == 0.0f) return Ld;
                weight = power_heuristic(1, scattpdf, 1, lightpdf);
            }
            SurfaceInteraction lisect;
            Ray ray = spawn_ray(it, wi);
            ray.medium = medium_toward(mif, it.n, wi);
            // Scene::intersect_tr (scene.rs:68-87): on through every surface that has no material
            bool found = false;
            RGB Tr(1.0f);
            for (;;) {
                const bool hits = S.intersect(ray, lisect, *ctx.c);
                if (ray.medium != PT_NONE) Tr = Tr * medium_tr_any(S, ray.medium, ray, sampler);
                if (!hits) { found = false; break; }
                if (S.prim_material[lisect.prim] != PT_NONE) { found = true; break; }
                IData a; a.p = lisect.p; a.p_error = lisect.p_error; a.n = lisect.n;
                const uint32_t seg_medium = ray.medium;
                ray = spawn_ray(a, ray.d);                                                     // isect.spawn_ray(&ray.d)
                ray.medium = medium_toward(surface_iface(S, lisect.prim, seg_medium), lisect.n, ray.d);
            }
            RGB li_(0.0f);
            if (found) { if (S.prim_light[lisect.prim] == li) li_ = isect_le(ctx, lisect, -wi); }
            else li_ = ctx.lights->light_le(li, ray);
            if (!li_.is_black()) Ld += f * li_ * Tr * weight / scattpdf;
        }
    }
    return Ld;
}
static RGB vol_uniform_sample_onelight(const RenderCtx &ctx, const IData &it, const MedIface &mif, const SurfaceInteraction *si, const BSDF *bsdf, Float g,
                                       SobolSampler &sampler, const Distribution1D *distrib) {
    if (ctx.scene->lights.empty()) return RGB(0.0f);
    Float lightpdf = 0.0f;
    size_t lightnum = distrib->sample_discrete(sampler.get_1d(), &lightpdf);
    if (lightpdf == 0.0f) return RGB(0.0f);
    P2 ulight = sampler.get_2d();
    P2 uscatt = sampler.get_2d();
    return vol_estimate_direct(ctx, it, mif, si, bsdf, g, uscatt, (uint32_t)lightnum, ulight, sampler) / lightpdf;
}
static RGB volpath_li(const RenderCtx &ctx, const PathParams &pp, Ray ray, SobolSampler &sampler, RayDiff rdiff = RayDiff()) {
    const Scene &S = *ctx.scene;
    RGB L(0.0f), beta(1.0f);
    bool specular_bounce = false;
    uint32_t bounces = 0;
    Float etascale = 1.0f;
    for (;;) {
        if (sampler.dim_overflow) break;   // the reference panics at the first dimension it does not have (sobol.rs:69-73); a path that material-less shells keep
                                           // alive (`bounces -= 1` below) gets there: the render then reports PT_ERR_SOBOL_DIMENSIONS
        SurfaceInteraction isect;
        bool found = S.intersect(ray, isect, *ctx.c);
        MediumVertex mi;
        if (ray.medium != PT_NONE) beta *= medium_sample_any(S, ray.medium, ray, sampler, mi);
        if (beta.is_black()) break;
        if (mi.valid) {
            if (bounces >= pp.max_depth) break;
            const Distribution1D *distrib = ctx.lights->lookup(mi.p);
            IData it; it.p = mi.p; it.p_error = V3(0, 0, 0); it.n = V3(0, 0, 0); it.wo = mi.wo;
            MedIface mif; mif.inside = mif.outside = mi.medium;
            L += beta * vol_uniform_sample_onelight(ctx, it, mif, nullptr, nullptr, mi.g, sampler, distrib);
            V3 wi;
            hg_sample_p(mi.g, mi.wo, wi, sampler.get_2d());
            ray = spawn_ray(it, wi); ray.medium = mi.medium;
            rdiff.has = false;   // mi.spawn_ray creates a ray without differentials (interaction.rs:32-36)
            specular_bounce = false;
        } else {
            if (bounces == 0 || specular_bounce) {
                if (found) L += isect_le(ctx, isect, -ray.d) * beta;
                else for (uint32_t li : S.infinite_lights) L += ctx.lights->light_le(li, ray) * beta;
            }
            if (!found || bounces >= pp.max_depth) break;
            const MedIface mif = surface_iface(S, isect.prim, ray.medium);
            BSDF bsdf;
            TabulatedBSSRDF bssrdf; bool has_bssrdf = false;
            TexCtx tctx;
            const bool textured = (bool)S.textures;
            if (textured) tctx = compute_differentials(isect, rdiff);
            rdiff.has = false;
            if (!compute_scattering_functions(S, isect, bsdf, &bssrdf, &has_bssrdf, textured ? &tctx : nullptr)) {
                // a material that leaves no BSDF (App. A #14): volpath.rs:127-131 then does `bounces -= 1; continue`, which skips the
                // increment at the end of the loop -- the count drops by one (and wraps below zero, ending the path at its next vertex)
                IData it; it.p = isect.p; it.p_error = isect.p_error; it.n = isect.n;
                uint32_t med = medium_toward(mif, isect.n, ray.d);
                ray = spawn_ray(it, ray.d); ray.medium = med;
                bounces -= 1;
                continue;
            }
            const Distribution1D *distrib = ctx.lights->lookup(isect.p);
            IData it; it.p = isect.p; it.p_error = isect.p_error; it.n = isect.n; it.wo = isect.wo;
            L += beta * vol_uniform_sample_onelight(ctx, it, mif, &isect, &bsdf, 0.0f, sampler, distrib);
            V3 wo = -ray.d, wi;
            Float pdf = 0.0f; int flags = 0;
            RGB f = bsdf.sample_f(wo, wi, sampler.get_2d(), pdf, BSDF_ALL, flags);
            if (f.is_black() || pdf == 0.0f) break;
            beta *= f * abs_dot(wi, isect.sh_n) / pdf;
            if (std::isinf(beta.y())) ctx.c->ref_asserts++;  // volpath.rs:176
            specular_bounce = (flags & BSDF_SPECULAR) != 0;
            if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
                Float eta = bsdf.eta;
                etascale *= (dot(wo, isect.n) > 0.0f) ? eta * eta : 1.0f / (eta * eta);
            }
            ray = spawn_ray(it, wi); ray.medium = medium_toward(mif, isect.n, wi);
            if (has_bssrdf && (flags & BSDF_TRANSMISSION)) {   // volpath.rs:186-214: as path.rs:177-204, with the probe's samples drawn in the OTHER order
                const Float s1 = sampler.get_1d();                 // (`sample_s(scene, sampler.get_1d(), &sampler.get_2d(), ..)`: arguments left to right)
                const P2 s2 = sampler.get_2d();
                if (std::isinf(beta.y())) ctx.c->ref_asserts++;    // volpath.rs:194
                V3 start, target; Float u1n = 0.0f;
                if (!bssrdf.probe_segment(s1, s2, start, target, u1n)) break;
                // TabulatedBSSRDF::sample_sp's chain (bssrdf.rs:367-395). Every hit's MediumInterface is the primitive's own when it is a transition and
                // the probe RAY's medium on both sides otherwise (primitive.rs:139-145); the next probe ray takes its medium from that interface
                // (`base = si.get_data()`, interaction.rs:38-43,54-66); the first one starts from an interaction without any (InteractionData::default)
                IData base; base.p = start; base.p_error = V3(0, 0, 0); base.n = V3(0, 0, 0);
                MedIface base_if; bool base_has_if = false;
                std::vector<SurfaceInteraction> chain; std::vector<MedIface> chain_if;
                for (;;) {
                    V3 d = target - base.p;
                    Ray r(offset_ray_origin(base.p, base.p_error, base.n, d), d, 1.0f - SHADOW_EPSILON, 0.0f);
                    r.medium = base_has_if ? medium_toward(base_if, base.n, d) : PT_NONE;
                    SurfaceInteraction si2;
                    if ((d.x == 0.0f && d.y == 0.0f && d.z == 0.0f) || !S.intersect(r, si2, *ctx.c)) break;
                    base.p = si2.p; base.p_error = si2.p_error; base.n = si2.n;
                    base_if = surface_iface(S, si2.prim, r.medium); base_has_if = true;
                    if (S.prim_material[si2.prim] == bssrdf.material) { chain.push_back(si2); chain_if.push_back(base_if); }
                }
                const size_t nfound = chain.size();
                if (nfound == 0) break;
                const size_t selected = (size_t)clampv((int64_t)(u1n * (Float)nfound), (int64_t)0, (int64_t)nfound - 1);
                SurfaceInteraction pi = chain[selected];
                const MedIface pif = chain_if[selected];
                pdf = bssrdf.pdf_sp(pi.p, pi.n) / (Float)nfound;
                const RGB Sp = bssrdf.sr(length(bssrdf.po_p - pi.p));
                if (Sp.is_black() || pdf == 0.0f) break;
                BSDF pibsdf; pibsdf.init(pi, 1.0f);
                { Bxdf b; b.kind = BX_BSSRDF; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.etab = bssrdf.eta; pibsdf.add(b); }
                pi.wo = pi.sh_n;
                beta *= Sp / pdf;
                const Distribution1D *d2 = ctx.lights->lookup(pi.p);
                IData pit; pit.p = pi.p; pit.p_error = pi.p_error; pit.n = pi.n; pit.wo = pi.wo;
                L += beta * vol_uniform_sample_onelight(ctx, pit, pif, &pi, &pibsdf, 0.0f, sampler, d2);
                const RGB ff = pibsdf.sample_f(pi.wo, wi, sampler.get_2d(), pdf, BSDF_ALL, flags);
                if (ff.is_black() || pdf == 0.0f) break;
                beta *= ff * abs_dot(wi, pi.sh_n) / pdf;
                if (std::isinf(beta.y())) ctx.c->ref_asserts++;    // volpath.rs:210
                specular_bounce = (flags & BSDF_SPECULAR) != 0;
                ray = spawn_ray(pit, wi); ray.medium = medium_toward(pif, pi.n, wi);
            }
        }
        RGB rrbeta = beta * etascale;
        if (rrbeta.max_component_value() < pp.rr_threshold && bounces > 3) {
            Float q = fmax_(1.0f - rrbeta.max_component_value(), 0.05f);
            if (sampler.get_1d() < q) break;
            beta = beta / (1.0f - q);
            if (std::isinf(beta.y())) ctx.c->ref_asserts++;  // path.rs:213 / volpath.rs:223
        }
        bounces += 1;
    }
    ctx.c->path_len[std::min<uint32_t>(bounces, 15)]++;
    return L;
}

// ---- camera (cameras/perspective.rs:120-179, main ray only) ----------------------------------------
struct Camera { M4 raster_to_camera, camera_to_world; Float lens_radius, focal_distance, shutter_open, shutter_close; };
static Ray generate_ray(const Camera &cam, const CameraSample &cs) {
    V3 pcamera = xf_point(cam.raster_to_camera, V3(cs.pfilm.x, cs.pfilm.y, 0.0f));
    Ray r(V3(0, 0, 0), normalize(pcamera), INF, 0.0f);
    if (cam.lens_radius > 0.0f) {
        P2 d = concentric_sample_disk(cs.plens);
        P2 plens(d.x * cam.lens_radius, d.y * cam.lens_radius);
        Float ft = cam.focal_distance / r.d.z;
        V3 pfocus = r.o + r.d * ft;
        r.o = V3(plens.x, plens.y, 0.0f);
        r.d = normalize(pfocus - r.o);
    }
    r.time = lerp(cs.time, cam.shutter_open, cam.shutter_close);
    return xf_ray(cam.camera_to_world, r);
}
// generate_ray_differential's auxiliary rays (perspective.rs:143-176) after Transform::transform_ray (:565-575) and
// Ray::scale_differential(1 / sqrt(spp)) (geometry/ray.rs:34-41, integrator.rs:340). `ray` is the transformed main ray.
static RayDiff generate_ray_differentials(const Camera &cam, const CameraSample &cs, const Ray &ray, uint32_t spp) {
    V3 pcamera = xf_point(cam.raster_to_camera, V3(cs.pfilm.x, cs.pfilm.y, 0.0f));
    // PerspectiveCamera::new (perspective.rs:64-70)
    V3 p2t = xf_point(cam.raster_to_camera, V3(0.0f, 0.0f, 0.0f));
    V3 dx_camera = xf_point(cam.raster_to_camera, V3(1.0f, 0.0f, 0.0f)) - p2t;
    V3 dy_camera = xf_point(cam.raster_to_camera, V3(0.0f, 1.0f, 0.0f)) - p2t;
    RayDiff d; d.has = true;
    if (cam.lens_radius > 0.0f) {
        P2 dk = concentric_sample_disk(cs.plens);
        P2 plens(dk.x * cam.lens_radius, dk.y * cam.lens_radius);
        V3 dx = normalize(pcamera + dx_camera);
        Float ft = cam.focal_distance / dx.z;
        V3 pfocus = V3(0.0f, 0.0f, 0.0f) + (dx * ft);
        d.rx_o = V3(plens.x, plens.y, 0.0f); d.rx_d = normalize(pfocus - d.rx_o);
        V3 dy = normalize(pcamera + dy_camera);
        ft = cam.focal_distance / dy.z;
        pfocus = V3(0.0f, 0.0f, 0.0f) + (dy * ft);
        d.ry_o = V3(plens.x, plens.y, 0.0f); d.ry_d = normalize(pfocus - d.ry_o);
    } else {
        // rd.o before the lens branch is (0,0,0); the main ray's camera-space origin
        d.rx_o = V3(0.0f, 0.0f, 0.0f); d.ry_o = V3(0.0f, 0.0f, 0.0f);
        d.rx_d = normalize(pcamera + dx_camera); d.ry_d = normalize(pcamera + dy_camera);
    }
    d.rx_o = xf_point(cam.camera_to_world, d.rx_o); d.ry_o = xf_point(cam.camera_to_world, d.ry_o);
    d.rx_d = xf_vector(cam.camera_to_world, d.rx_d); d.ry_d = xf_vector(cam.camera_to_world, d.ry_d);
    const Float sc = 1.0f / std::sqrt((Float)spp);
    d.rx_o = ray.o + (d.rx_o - ray.o) * sc; d.ry_o = ray.o + (d.ry_o - ray.o) * sc;
    d.rx_d = ray.d + (d.rx_d - ray.d) * sc; d.ry_d = ray.d + (d.ry_d - ray.d) * sc;
    return d;
}

// ---- film (core/film.rs) -------------------------------------------------------------------------
struct FilmParams {
    int32_t crop[4]; Float radius[2]; Float table[256]; Float max_lum; Float scale;
};
struct FilmTile {
    int64_t b[4];  // pixel bounds xmin ymin xmax ymax
    std::vector<Float> rgbw;  // 4 per pixel: contrib_sum rgb + filter_weight_sum
    const FilmParams *fp;
    FilmTile(const FilmParams &f, const int64_t sb[4]) : fp(&f) {  // get_film_tile, film.rs:125-140
        Float p0x = std::ceil((Float)sb[0] - 0.5f - f.radius[0]), p0y = std::ceil((Float)sb[1] - 0.5f - f.radius[1]);
        Float p1x = std::floor((Float)sb[2] - 0.5f + f.radius[0]), p1y = std::floor((Float)sb[3] - 0.5f + f.radius[1]);
        b[0] = std::max<int64_t>(f2i_sat(p0x), f.crop[0]); b[1] = std::max<int64_t>(f2i_sat(p0y), f.crop[1]);
        b[2] = std::min<int64_t>(f2i_sat(p1x) + 1, f.crop[2]); b[3] = std::min<int64_t>(f2i_sat(p1y) + 1, f.crop[3]);
        int64_t w = std::max<int64_t>(0, b[2] - b[0]), h = std::max<int64_t>(0, b[3] - b[1]);
        rgbw.assign((size_t)(w * h * 4), 0.0f);
    }
    void add_sample(P2 pfilm, RGB L, Float sample_weight, Counters &c) {  // film.rs:292-331
        if (L.y() > fp->max_lum) L *= RGB(fp->max_lum / L.y());
        Float dx = pfilm.x - 0.5f, dy = pfilm.y - 0.5f;
        int64_t p0x = f2i_sat(std::ceil(dx - fp->radius[0])), p0y = f2i_sat(std::ceil(dy - fp->radius[1]));
        int64_t p1x = f2i_sat(std::floor(dx + fp->radius[0])) + 1, p1y = f2i_sat(std::floor(dy + fp->radius[1])) + 1;
        p0x = std::max(p0x, b[0]); p0y = std::max(p0y, b[1]); p1x = std::min(p1x, b[2]); p1y = std::min(p1y, b[3]);
        Float invrx = 1.0f / fp->radius[0], invry = 1.0f / fp->radius[1];
        int64_t w = b[2] - b[0];
        for (int64_t y = p0y; y < p1y; ++y) {
            Float fy = std::fabs(((Float)y - dy) * invry * 16.0f);
            uint64_t iy = std::min<uint64_t>(f2u_sat(std::floor(fy)), 15);
            for (int64_t x = p0x; x < p1x; ++x) {
                Float fx = std::fabs(((Float)x - dx) * invrx * 16.0f);
                uint64_t ix = std::min<uint64_t>(f2u_sat(std::floor(fx)), 15);
                Float fw = fp->table[iy * 16 + ix];
                Float *px = &rgbw[(size_t)(((y - b[1]) * w + (x - b[0])) * 4)];
                RGB contrib = L * RGB(sample_weight) * RGB(fw);
                px[0] += contrib.c[0]; px[1] += contrib.c[1]; px[2] += contrib.c[2]; px[3] += fw;
                c.splats++;
            }
        }
    }
};

struct RenderJob {
    const Scene *scene; LightSampler lights; PtRenderParams rp; Camera cam; FilmParams fp;
};

static void render_tiles(const RenderJob &job, float *film_xyzw, int nthreads, Counters &total, std::atomic<bool> &dim_overflow) {
    const PtRenderParams &rp = job.rp;
    const int32_t *sb = rp.sample_bounds;
    int64_t ntx = (sb[2] - sb[0] + 15) / 16, nty = (sb[3] - sb[1] + 15) / 16;
    int64_t ntiles = ntx * nty;
    std::atomic<int64_t> next(0);
    std::mutex film_mu;
    int64_t fw = rp.cropped_pixel_bounds[2] - rp.cropped_pixel_bounds[0];
    PathParams pp{rp.max_depth, rp.rr_threshold};
    auto worker = [&]() {
        Counters c;
        RenderCtx ctx{job.scene, &job.lights, &c};
        for (;;) {
            int64_t t = next.fetch_add(1);
            if (t >= ntiles) break;
            if (rp.tile_world > 1 && (uint64_t)t % rp.tile_world != rp.tile_rank) continue;
            int64_t tx = t % ntx, ty = t / ntx;
            int64_t tb[4] = {sb[0] + tx * 16, sb[1] + ty * 16, 0, 0};
            tb[2] = std::min<int64_t>(tb[0] + 16, sb[2]); tb[3] = std::min<int64_t>(tb[1] + 16, sb[3]);
            SobolSampler sampler(rp.spp, rp.sample_bounds, rp.sampler_type, rp.sample_at_pixel_center != 0);
            FilmTile tile(job.fp, tb);
            for (int64_t y = tb[1]; y < tb[3]; ++y)
                for (int64_t x = tb[0]; x < tb[2]; ++x) {
                    sampler.start_pixel(x, y);
                    if (!(x >= rp.pixel_bounds[0] && x < rp.pixel_bounds[2] && y >= rp.pixel_bounds[1] && y < rp.pixel_bounds[3])) continue;
                    for (;;) {
                        CameraSample cs = sampler.get_camera_sample(x, y);
                        Ray ray = generate_ray(job.cam, cs);
                        c.camera_rays++;
                        RayDiff rdiff;
                        if (job.scene->textures) rdiff = generate_ray_differentials(job.cam, cs, ray, rp.spp);
                        if (rp.integrator == PT_INTEGRATOR_VOLPATH) ray.medium = rp.camera_medium;   // perspective.rs:114
                        RGB L = rp.integrator == PT_INTEGRATOR_VOLPATH ? volpath_li(ctx, pp, ray, sampler, rdiff) : path_li(ctx, pp, ray, sampler, rdiff);
                        if (L.has_nans()) { L = RGB(0.0f); c.san_nan++; }
                        else if (L.y() < -1.0e-5f) { L = RGB(0.0f); c.san_neg++; }
                        else if (std::isinf(L.y())) { L = RGB(0.0f); c.san_inf++; }
                        tile.add_sample(cs.pfilm, L, 1.0f, c);
                        if (!sampler.start_next_sample()) break;
                    }
                }
            if (sampler.dim_overflow) dim_overflow = true;
            // merge_film_tile, film.rs:142-161
            std::lock_guard<std::mutex> g(film_mu);
            int64_t w = tile.b[2] - tile.b[0];
            for (int64_t y = tile.b[1]; y < tile.b[3]; ++y)
                for (int64_t x = tile.b[0]; x < tile.b[2]; ++x) {
                    const Float *px = &tile.rgbw[(size_t)(((y - tile.b[1]) * w + (x - tile.b[0])) * 4)];
                    Float xyz[3]; rgb_to_xyz(px, xyz);
                    float *out = film_xyzw + ((y - rp.cropped_pixel_bounds[1]) * fw + (x - rp.cropped_pixel_bounds[0])) * 4;
                    out[0] += xyz[0]; out[1] += xyz[1]; out[2] += xyz[2]; out[3] += px[3];
                }
        }
        std::lock_guard<std::mutex> g(film_mu);
        total.add(c);
    };
    if (nthreads <= 1) worker();
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < nthreads; ++i) th.emplace_back(worker);
        for (auto &t : th) t.join();
    }
}

}  // namespace ref

// ---- oracle C ABI (ctypes) -----------------------------------------------------------------------
using namespace ref;

struct orc_scene { Scene scene; Counters counters; double last_render_seconds = 0; };

extern "C" {

int orc_load_tables(const char *path) { return sobol_tables().load(path) ? 0 : 1; }

int orc_scene_create(const PtSceneDesc *d, orc_scene **out) {
    if (!d || !out) return PT_ERR_INVALID_ARG;
    orc_scene *h = new orc_scene();
    Scene &s = h->scene;
    s.P.resize(d->n_vertices);
    for (uint32_t i = 0; i < d->n_vertices; ++i) s.P[i] = V3(d->P[3 * i], d->P[3 * i + 1], d->P[3 * i + 2]);
    if (d->N) { s.N.resize(d->n_vertices); for (uint32_t i = 0; i < d->n_vertices; ++i) s.N[i] = V3(d->N[3 * i], d->N[3 * i + 1], d->N[3 * i + 2]); }
    if (d->S) { s.S.resize(d->n_vertices); for (uint32_t i = 0; i < d->n_vertices; ++i) s.S[i] = V3(d->S[3 * i], d->S[3 * i + 1], d->S[3 * i + 2]); }
    if (d->UV) { s.UV.resize(d->n_vertices); for (uint32_t i = 0; i < d->n_vertices; ++i) s.UV[i] = P2(d->UV[2 * i], d->UV[2 * i + 1]); }
    s.idx.assign(d->indices, d->indices + 3 * (size_t)d->n_triangles);
    if (d->tri_flags) s.tri_flags.assign(d->tri_flags, d->tri_flags + d->n_triangles); else s.tri_flags.assign(d->n_triangles, 0);
    if (d->n_spheres) s.spheres.assign(d->spheres, d->spheres + d->n_spheres);
    s.prim_shape.assign(d->prim_shape, d->prim_shape + d->n_prims);
    s.prim_material.assign(d->prim_material, d->prim_material + d->n_prims);
    s.prim_light.assign(d->prim_light, d->prim_light + d->n_prims);
    if (d->n_materials) s.materials.assign(d->materials, d->materials + d->n_materials);
    for (uint32_t i = 0; i < d->n_bssrdf_tables; ++i) {
        const PtBSSRDFTable &t = d->bssrdf_tables[i];
        BssrdfTable b; b.n_rho = (int)t.n_rho; b.n_radius = (int)t.n_radius;
        b.rho_samples.assign(t.rho_samples, t.rho_samples + t.n_rho); b.radius_samples.assign(t.radius_samples, t.radius_samples + t.n_radius);
        b.profile.assign(t.profile, t.profile + (size_t)t.n_rho * t.n_radius); b.rhoeff.assign(t.rhoeff, t.rhoeff + t.n_rho);
        b.profile_cdf.assign(t.profile_cdf, t.profile_cdf + (size_t)t.n_rho * t.n_radius);
        s.bssrdf_tables.push_back(std::move(b));
    }
    for (const PtMaterial &m : s.materials)
        if (m.type == PT_MAT_SUBSURFACE && m.bssrdf_table >= s.bssrdf_tables.size()) return PT_ERR_INVALID_ARG;
    if (d->n_textures) {
        auto ts = std::make_shared<TextureSet>();
        ts->tex.assign(d->textures, d->textures + d->n_textures);
        for (uint32_t i = 0; i < d->n_images; ++i) {
            const PtImage &im = d->images[i];
            ImagePyramid p; p.width = (int)im.width; p.height = (int)im.height; p.n_levels = (int)im.n_levels; p.channels = (int)im.channels;
            size_t off = 0;
            for (int l = 0; l < p.n_levels; ++l) { p.offset.push_back(off); off += (size_t)p.ures(l) * p.vres(l) * p.channels; }
            p.texels.assign(im.texels, im.texels + off);
            ts->images.push_back(std::move(p));
        }
        if (d->ewa_weight_lut) ts->ewa_lut.assign(d->ewa_weight_lut, d->ewa_weight_lut + 128);
        else ts->ewa_lut.assign(128, 0.0f);
        for (const PtTexture &t : ts->tex) {
            for (int k = 0; k < 3; ++k) if (t.child[k] >= (int32_t)d->n_textures) return PT_ERR_INVALID_ARG;
            if (t.type == PT_TEX_IMAGEMAP && t.image >= d->n_images) return PT_ERR_INVALID_ARG;
        }
        for (const PtMaterial &m : s.materials) for (int k = 0; k < 16; ++k) if (m.tex[k] >= (int32_t)d->n_textures) return PT_ERR_INVALID_ARG;
        s.textures = ts;
        if (d->tri_alpha) s.tri_alpha.assign(d->tri_alpha, d->tri_alpha + d->n_triangles);
        if (d->tri_shadow_alpha) s.tri_shadow_alpha.assign(d->tri_shadow_alpha, d->tri_shadow_alpha + d->n_triangles);
        for (int32_t a : s.tri_alpha) if (a >= (int32_t)d->n_textures) return PT_ERR_INVALID_ARG;
        for (int32_t a : s.tri_shadow_alpha) if (a >= (int32_t)d->n_textures) return PT_ERR_INVALID_ARG;
    }
    if (d->n_lights) s.lights.assign(d->lights, d->lights + d->n_lights);
    for (uint32_t i = 0; i < d->n_lights; ++i) if (s.lights[i].type == PT_LIGHT_INFINITE) s.infinite_lights.push_back(i);
    if (d->env_texels) {
        s.env_power_lookup = RGB(d->env_power_lookup[0], d->env_power_lookup[1], d->env_power_lookup[2]);
        s.env_w = d->env_width; s.env_h = d->env_height;
        s.env_texels.resize((size_t)s.env_w * s.env_h);
        for (size_t i = 0; i < s.env_texels.size(); ++i) s.env_texels[i] = RGB(d->env_texels[3 * i], d->env_texels[3 * i + 1], d->env_texels[3 * i + 2]);
        s.env_importance.assign(d->env_importance, d->env_importance + (size_t)4 * s.env_w * s.env_h);
    }
    s.max_node_prims = d->max_node_prims ? d->max_node_prims : 4;
    s.split_method = d->split_method;
    if (d->n_media && d->media) {
        s.media.assign(d->media, d->media + d->n_media);
        s.grid_aux.resize(d->n_media);
        for (uint32_t i = 0; i < d->n_media; ++i) {
            if (s.media[i].type != PT_MEDIUM_GRID) continue;
            const PtMedium &m = s.media[i];
            Scene::GridAux &g = s.grid_aux[i];
            g.density.assign(m.density, m.density + (size_t)m.nx * m.ny * m.nz);
            g.sigma_t = m.sigma_a[0] + m.sigma_s[0];
            Float maxd = 0.0f;
            for (Float v : g.density) maxd = fmax_(maxd, v);
            g.inv_max_density = 1.0f / maxd;
        }
    }
    if (d->prim_medium_inside && d->prim_medium_outside) { s.prim_med_in.assign(d->prim_medium_inside, d->prim_medium_inside + d->n_prims); s.prim_med_out.assign(d->prim_medium_outside, d->prim_medium_outside + d->n_prims); }
    if (d->n_instances && d->top_refs) {
        s.objects.assign(d->objects, d->objects + d->n_objects);
        s.instances.assign(d->instances, d->instances + d->n_instances);
        s.top_refs.assign(d->top_refs, d->top_refs + d->n_top);
    }
    s.build_object_accels();
    if (d->nodes && d->n_nodes) {
        s.nodes.assign(d->nodes, d->nodes + d->n_nodes);
        s.ordered.assign(d->ordered_prims, d->ordered_prims + s.n_top());
    } else s.build_bvh();
    if (!s.nodes.empty()) {
        s.wb.pmin = V3(s.nodes[0].bmin[0], s.nodes[0].bmin[1], s.nodes[0].bmin[2]);
        s.wb.pmax = V3(s.nodes[0].bmax[0], s.nodes[0].bmax[1], s.nodes[0].bmax[2]);
    }
    *out = h;
    return PT_OK;
}
void orc_scene_destroy(orc_scene *h) { delete h; }

int orc_scene_bvh_info(const orc_scene *h, uint32_t *n_nodes, uint32_t *n_prims) {
    *n_nodes = (uint32_t)h->scene.nodes.size(); *n_prims = (uint32_t)h->scene.ordered.size(); return PT_OK;
}
int orc_scene_bvh_read(const orc_scene *h, PtBVHNode *nodes, uint32_t *ordered) {
    std::memcpy(nodes, h->scene.nodes.data(), h->scene.nodes.size() * sizeof(PtBVHNode));
    std::memcpy(ordered, h->scene.ordered.data(), h->scene.ordered.size() * 4);
    return PT_OK;
}

int orc_render(orc_scene *h, const PtRenderParams *rp, float *film_xyzw, int nthreads) {
    RenderJob job;
    job.scene = &h->scene; job.rp = *rp;
    job.lights.init(h->scene, (int)rp->light_strategy);
    job.cam.raster_to_camera = m4_from(rp->raster_to_camera); job.cam.camera_to_world = m4_from(rp->camera_to_world);
    job.cam.lens_radius = rp->lens_radius; job.cam.focal_distance = rp->focal_distance;
    job.cam.shutter_open = rp->shutter_open; job.cam.shutter_close = rp->shutter_close;
    std::memcpy(job.fp.crop, rp->cropped_pixel_bounds, 16);
    job.fp.radius[0] = rp->filter_radius[0]; job.fp.radius[1] = rp->filter_radius[1];
    std::memcpy(job.fp.table, rp->filter_table, sizeof job.fp.table);
    job.fp.max_lum = rp->max_sample_luminance; job.fp.scale = rp->scale;
    h->counters = Counters();
    std::atomic<bool> overflow(false);
    auto t0 = std::chrono::steady_clock::now();
    render_tiles(job, film_xyzw, nthreads, h->counters, overflow);
    h->last_render_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return overflow ? PT_ERR_SOBOL_DIMENSIONS : PT_OK;
}
double orc_last_render_seconds(const orc_scene *h) { return h->last_render_seconds; }

int orc_get_counters(const orc_scene *h, PtCounters *o) {
    std::memset(o, 0, sizeof *o);
    const Counters &c = h->counters;
    o->camera_rays = c.camera_rays; o->intersect_tests = c.intersect_tests; o->shadow_tests = c.shadow_tests;
    o->bvh_nodes_visited = c.nodes; o->triangle_tests = c.tri_tests; o->sphere_tests = c.sphere_tests;
    o->zero_radiance_paths_num = c.zero_num; o->zero_radiance_paths_den = c.zero_den;
    for (int i = 0; i < 16; ++i) o->path_length_hist[i] = c.path_len[i];
    o->sanitized_nan = c.san_nan; o->sanitized_negative = c.san_neg; o->sanitized_infinite = c.san_inf;
    o->film_splats = c.splats; o->reference_asserts = c.ref_asserts;
    return PT_OK;
}

// Film::write_image normalisation (film.rs:217-258)
int orc_film_resolve(const float *xyzw, uint32_t npix, float scale, float *rgb) {
    for (uint32_t i = 0; i < npix; ++i) {
        Float c[3]; xyz_to_rgb(xyzw + 4 * i, c);
        Float w = xyzw[4 * i + 3];
        if (w != 0.0f) { Float inv = 1.0f / w; for (int k = 0; k < 3; ++k) c[k] = fmax_(c[k] * inv, 0.0f); }
        for (int k = 0; k < 3; ++k) rgb[3 * i + k] = c[k] * scale;
    }
    return PT_OK;
}

int orc_trace_closest(orc_scene *h, uint32_t n, const float *o, const float *d, const float *tmax, uint32_t *prim, float *t, float *b) {
    Counters c;
    for (uint32_t i = 0; i < n; ++i) {
        Ray r(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]);
        SurfaceInteraction si;
        if (h->scene.intersect(r, si, c)) { prim[i] = si.prim; t[i] = si.t; b[3 * i] = si.b[0]; b[3 * i + 1] = si.b[1]; b[3 * i + 2] = si.b[2]; }
        else { prim[i] = PT_NONE; t[i] = 0; b[3 * i] = b[3 * i + 1] = b[3 * i + 2] = 0; }
    }
    h->counters = c;
    return PT_OK;
}
int orc_trace_any(orc_scene *h, uint32_t n, const float *o, const float *d, const float *tmax, uint8_t *hit) {
    Counters c;
    for (uint32_t i = 0; i < n; ++i) {
        Ray r(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]);
        hit[i] = h->scene.intersect_p(r, c) ? 1 : 0;
    }
    h->counters = c;
    return PT_OK;
}
int orc_sobol_samples(const int32_t sb[4], uint32_t n, const int32_t *pixel_xy, const uint32_t *sample_num, uint32_t n_dims, float *out, uint64_t *out_index) {
    SobolSampler s(1u << 30, sb);
    for (uint32_t i = 0; i < n; ++i) {
        s.start_pixel(pixel_xy[2 * i], pixel_xy[2 * i + 1]);
        s.set_sample_number(sample_num[i]);
        if (out_index) out_index[i] = s.interval_sample_index;
        for (uint32_t k = 0; k < n_dims; ++k) out[(size_t)i * n_dims + k] = s.sample_dimension(s.interval_sample_index, (int)k);
    }
    return PT_OK;
}
int orc_halton_samples(const int32_t sb[4], uint32_t at_center, uint32_t n, const int32_t *pixel_xy, const uint32_t *sample_num, uint32_t n_dims, float *out, uint64_t *out_index) {
    SobolSampler s(1u << 30, sb, PT_SAMPLER_HALTON, at_center != 0);
    for (uint32_t i = 0; i < n; ++i) {
        s.start_pixel(pixel_xy[2 * i], pixel_xy[2 * i + 1]);
        s.set_sample_number(sample_num[i]);
        if (out_index) out_index[i] = s.interval_sample_index;
        for (uint32_t k = 0; k < n_dims; ++k) out[(size_t)i * n_dims + k] = s.sample_dimension(s.interval_sample_index, (int)k);
    }
    return PT_OK;
}
// KAT hooks: radical_inverse(base_index, n) (pbrt_macros:92-111) and the Halton digit permutation of a dimension
float orc_radical_inverse_any(uint32_t base_index, uint64_t n) {
    if (base_index == 0) return (float)reverse_bits64_h(n) * 0x1.0p-64f;
    return radical_inverse_base(halton_tables().primes[base_index], n);
}
uint32_t orc_halton_permutation(uint32_t dim, uint16_t *out) {
    const HaltonTables &T = halton_tables();
    if (dim >= 1000) return 0;
    for (uint32_t j = 0; j < T.primes[dim]; ++j) out[j] = T.perm[T.sums[dim] + j];
    return T.primes[dim];
}
int orc_camera_rays(const PtRenderParams *rp, uint32_t n, const float *cs, float *out_o, float *out_d) {
    Camera cam;
    cam.raster_to_camera = m4_from(rp->raster_to_camera); cam.camera_to_world = m4_from(rp->camera_to_world);
    cam.lens_radius = rp->lens_radius; cam.focal_distance = rp->focal_distance; cam.shutter_open = rp->shutter_open; cam.shutter_close = rp->shutter_close;
    for (uint32_t i = 0; i < n; ++i) {
        CameraSample c; c.pfilm = P2(cs[5 * i], cs[5 * i + 1]); c.time = cs[5 * i + 2]; c.plens = P2(cs[5 * i + 3], cs[5 * i + 4]);
        Ray r = generate_ray(cam, c);
        out_o[3 * i] = r.o.x; out_o[3 * i + 1] = r.o.y; out_o[3 * i + 2] = r.o.z;
        out_d[3 * i] = r.d.x; out_d[3 * i + 1] = r.d.y; out_d[3 * i + 2] = r.d.z;
    }
    return PT_OK;
}

// ---- known-answer-test hooks (reference tests/*.rs) ------------------------------------------------
float orc_sobol_sample_float(uint64_t index, int dim, uint32_t scramble) { return sobol_sample_float(index, dim, scramble); }
float orc_radical_inverse(int base_index, uint64_t n) { return radical_inverse(base_index, n); }
float orc_next_float_up(float v) { return next_float_up(v); }
float orc_next_float_down(float v) { return next_float_down(v); }
int orc_find_interval(int size, const float *a, float x) { return find_interval(size, [&](int i) { return a[i] <= x; }); }
uint32_t orc_rng_u32_stream(uint64_t seq, int use_default, uint32_t n, uint32_t *out, float *outf) {
    RNG r = use_default ? RNG() : RNG(seq);
    for (uint32_t i = 0; i < n; ++i) { if (out) out[i] = r.uniform_u32(); else outf[i] = r.uniform_float(); }
    return n;
}
// Distribution1D (tests/sampling.rs:202-257)
int orc_dist1d_sample_discrete(const float *func, int n, float u, float *pdf, float *uremapped) {
    Distribution1D d(std::vector<Float>(func, func + n));
    return (int)d.sample_discrete(u, pdf, uremapped);
}
float orc_dist1d_discrete_pdf(const float *func, int n, int index) { return Distribution1D(std::vector<Float>(func, func + n)).discrete_pdf((size_t)index); }
float orc_dist1d_sample_continuous(const float *func, int n, float u, float *pdf, int *offset) {   // Distribution1D::sample_continous (sampling.rs:38-64); pdf / offset may be null like the reference's Options
    size_t off = 0;
    const float x = Distribution1D(std::vector<Float>(func, func + n)).sample_continuous(u, pdf, &off);
    if (offset) *offset = (int)off;
    return x;
}
// deterministic math
float orc_dm_sin(float x) { return dm_sinf(x); }
float orc_dm_cos(float x) { return dm_cosf(x); }
float orc_dm_acos(float x) { return dm_acosf(x); }
float orc_dm_atan2(float y, float x) { return dm_atan2f(y, x); }
float orc_dm_log(float x) { return dm_logf(x); }
// single-triangle tests (tests/shapes.rs): a 1-triangle scene is created by the caller.
int orc_tri_intersect(orc_scene *h, uint32_t tri, const float *o, const float *d, float tmax, float *t, float *b, float *p, float *perr, float *n) {
    Ray r(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), tmax);
    Float tt, bb[3];
    if (!h->scene.tri_intersect(tri, r, tt, bb)) return 0;
    SurfaceInteraction si; h->scene.tri_fill_interaction(tri, r, tt, bb, true, si);
    *t = tt; b[0] = bb[0]; b[1] = bb[1]; b[2] = bb[2];
    p[0] = si.p.x; p[1] = si.p.y; p[2] = si.p.z; perr[0] = si.p_error.x; perr[1] = si.p_error.y; perr[2] = si.p_error.z;
    n[0] = si.n.x; n[1] = si.n.y; n[2] = si.n.z;
    return 1;
}
int orc_tri_intersect_p(orc_scene *h, uint32_t tri, const float *o, const float *d, float tmax) {
    Ray r(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), tmax);
    Float tt, bb[3];
    return h->scene.tri_hit_params(tri, r, tt, bb) ? 1 : 0;
}
void orc_offset_ray_origin(const float *p, const float *perr, const float *n, const float *w, float *out) {
    V3 r = offset_ray_origin(V3(p[0], p[1], p[2]), V3(perr[0], perr[1], perr[2]), V3(n[0], n[1], n[2]), V3(w[0], w[1], w[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

}  // extern "C"

// ---- property tests restated from the reference's tests/shapes.rs (run inside the oracle for speed) ----
extern "C" {
// tests/shapes.rs:173-224 triangle_reintersect: same PCG32 seeds (RNG::new(i)), pexp(rng, 8), 10 000 spawned rays
// per triangle must not re-hit it (intersect_p and intersect). Returns the number of violations; *n_tested counts
// the triangles that were actually hit by the first ray.
static float pexp_(RNG &rng, float e) { float logu = lerp(rng.uniform_float(), -e, e); return std::pow(10.0f, logu); }
int orc_test_triangle_reintersect(int n_seeds, int rays_per_tri, int *n_tested) {
    int failures = 0, tested = 0;
    for (int i = 0; i < n_seeds; ++i) {
        RNG rng((uint64_t)i);
        V3 v[3];
        for (int j = 0; j < 3; ++j) { v[j].x = pexp_(rng, 8.0f); v[j].y = pexp_(rng, 8.0f); v[j].z = pexp_(rng, 8.0f); }
        if (length_squared(cross(v[1] - v[0], v[2] - v[0])) < 1.0e-20f) continue;
        Scene s;
        s.P = {v[0], v[1], v[2]}; s.idx = {0, 1, 2}; s.tri_flags = {0};
        P2 u; u.x = rng.uniform_float(); u.y = rng.uniform_float();
        P2 b = uniform_sample_triangle(u);
        V3 ptri = v[0] * b.x + v[1] * b.y + v[2] * (1.0f - b.x - b.y);
        V3 o; o.x = pexp_(rng, 8.0f); o.y = pexp_(rng, 8.0f); o.z = pexp_(rng, 8.0f);
        Ray r(o, ptri - o, INF, 0.0f);
        Float t, bb[3];
        if (!s.tri_intersect(0, r, t, bb)) continue;
        SurfaceInteraction isect; s.tri_fill_interaction(0, r, t, bb, true, isect);
        tested++;
        for (int j = 0; j < rays_per_tri; ++j) {
            P2 uu; uu.x = rng.uniform_float(); uu.y = rng.uniform_float();
            V3 w = uniform_sample_sphere(uu);
            IData it; it.p = isect.p; it.p_error = isect.p_error; it.n = isect.n;
            Ray rout = spawn_ray(it, w);
            Float t2, b2[3];
            if (s.tri_hit_params(0, rout, t2, b2)) failures++;
            if (s.tri_intersect(0, rout, t2, b2)) failures++;
            V3 p2; p2.x = pexp_(rng, 8.0f); p2.y = pexp_(rng, 8.0f); p2.z = pexp_(rng, 8.0f);
            // spawn_rayto_point, interaction.rs:38-43
            V3 d = p2 - it.p;
            Ray r2(offset_ray_origin(it.p, it.p_error, it.n, d), d, 1.0f - SHADOW_EPSILON, 0.0f);
            if (s.tri_hit_params(0, r2, t2, b2)) failures++;
            if (s.tri_intersect(0, r2, t2, b2)) failures++;
        }
    }
    if (n_tested) *n_tested = tested;
    return failures;
}
}

extern "C" {
// tests/shapes.rs:421-487,538-565: full / partial sphere re-intersection with the reference's seeds RNG::new(0..n).
int orc_test_sphere_reintersect(int n_seeds, int rays_per_shape, int partial, int *n_tested) {
    int failures = 0, tested = 0;
    for (int i = 0; i < n_seeds; ++i) {
        RNG rng((uint64_t)i);
        float radius = pexp_(rng, 4.0f);
        float zmin = -radius, zmax = radius, phimax = 360.0f;
        if (partial) {
            zmin = (rng.uniform_float() < 0.5f) ? -radius : lerp(rng.uniform_float(), -radius, radius);
            zmax = (rng.uniform_float() < 0.5f) ? radius : lerp(rng.uniform_float(), -radius, radius);
            phimax = (rng.uniform_float() < 0.5f) ? 360.0f : rng.uniform_float() * 360.0f;
        }
        PtSphere S; std::memset(&S, 0, sizeof S);   // Sphere::new (sphere.rs:31-50)
        for (int k = 0; k < 4; ++k) S.object_to_world[5 * k] = S.world_to_object[5 * k] = 1.0f;
        S.radius = radius;
        S.z_min = clampv(std::fmin(zmin, zmax), -radius, radius); S.z_max = clampv(std::fmax(zmin, zmax), -radius, radius);
        S.theta_min = std::acos(clampv(std::fmin(zmin, zmax) / radius, -1.0f, 1.0f));
        S.theta_max = std::acos(clampv(std::fmax(zmin, zmax) / radius, -1.0f, 1.0f));
        S.phi_max = (PI / 180.0f) * clampv(phimax, 0.0f, 360.0f);
        Scene s; s.spheres.push_back(S);
        // test_reintersect_convex
        V3 o; o.x = pexp_(rng, 8.0f); o.y = pexp_(rng, 8.0f); o.z = pexp_(rng, 8.0f);
        Bounds3 bbox = s.sphere_world_bound(0);
        V3 t; t.x = rng.uniform_float(); t.y = rng.uniform_float(); t.z = rng.uniform_float();
        V3 p2 = bbox.lerp3(t);
        Ray r(o, p2 - o, INF, 0.0f);
        if (rng.uniform_float() < 0.5f) r.d = normalize(r.d);
        SurfaceInteraction isect; Float thit;
        if (!s.sphere_intersect(0, r, thit, isect, true)) continue;
        tested++;
        for (int j = 0; j < rays_per_shape; ++j) {
            P2 u; u.x = rng.uniform_float(); u.y = rng.uniform_float();
            V3 w = face_forward(uniform_sample_sphere(u), isect.n);
            IData it; it.p = isect.p; it.p_error = isect.p_error; it.n = isect.n;
            Ray rout = spawn_ray(it, w);
            SurfaceInteraction tmp; Float th2;
            if (s.sphere_intersect_p(0, rout)) failures++;
            if (s.sphere_intersect(0, rout, th2, tmp, true)) failures++;
            V3 p3; p3.x = pexp_(rng, 8.0f); p3.y = pexp_(rng, 8.0f); p3.z = pexp_(rng, 8.0f);
            w = face_forward(p3 - isect.p, isect.n);
            p3 = isect.p + w;
            V3 d = p3 - it.p;
            Ray r2(offset_ray_origin(it.p, it.p_error, it.n, d), d, 1.0f - SHADOW_EPSILON, 0.0f);
            if (s.sphere_intersect_p(0, r2)) failures++;
            // the reference overwrites `isect` here when the (unexpected) hit happens; it must not happen
            if (s.sphere_intersect(0, r2, th2, tmp, true)) failures++;
        }
    }
    if (n_tested) *n_tested = tested;
    return failures;
}
}

// ---- KAT hooks for the BSSRDF restatement (tests/test_oracle_kats.py) -------------------------------------------
extern "C" {
using namespace ref;
static void kat_bssrdf(const PtBSSRDFTable *t, const float *sigma_a, const float *sigma_s, float eta, BssrdfTable &tb, TabulatedBSSRDF &b) {
    tb.n_rho = (int)t->n_rho; tb.n_radius = (int)t->n_radius;
    tb.rho_samples.assign(t->rho_samples, t->rho_samples + t->n_rho); tb.radius_samples.assign(t->radius_samples, t->radius_samples + t->n_radius);
    tb.profile.assign(t->profile, t->profile + (size_t)t->n_rho * t->n_radius); tb.rhoeff.assign(t->rhoeff, t->rhoeff + t->n_rho);
    tb.profile_cdf.assign(t->profile_cdf, t->profile_cdf + (size_t)t->n_rho * t->n_radius);
    SurfaceInteraction si; si.p = V3(0, 0, 0); si.n = V3(0, 0, 1); si.sh_n = V3(0, 0, 1); si.sh_dpdu = V3(1, 0, 0);
    b.init(si, 0, eta, RGB(sigma_a[0], sigma_a[1], sigma_a[2]), RGB(sigma_s[0], sigma_s[1], sigma_s[2]), &tb);
}
// Sr(r) and pdf_sr(ch, r) for n radii
int orc_bssrdf_sr(const PtBSSRDFTable *t, const float *sigma_a, const float *sigma_s, float eta, uint32_t n, const float *r, float *sr3, float *pdf3) {
    BssrdfTable tb; TabulatedBSSRDF b; kat_bssrdf(t, sigma_a, sigma_s, eta, tb, b);
    for (uint32_t i = 0; i < n; ++i) {
        RGB s = b.sr(r[i]);
        for (int c = 0; c < 3; ++c) { sr3[3 * i + c] = s.c[c]; pdf3[3 * i + c] = b.pdf_sr(c, r[i]); }
    }
    return 0;
}
int orc_bssrdf_sample_sr(const PtBSSRDFTable *t, const float *sigma_a, const float *sigma_s, float eta, int ch, uint32_t n, const float *u, float *r) {
    BssrdfTable tb; TabulatedBSSRDF b; kat_bssrdf(t, sigma_a, sigma_s, eta, tb, b);
    for (uint32_t i = 0; i < n; ++i) r[i] = b.sample_sr(ch, u[i]);
    return 0;
}
int orc_catmull_rom_weights(int size, const float *nodes, float x, int *offset, float *w4) {
    return catmull_rom_weights(size, nodes, x, *offset, w4) ? 1 : 0;
}
float orc_bssrdf_sw(float eta, float cos_theta_) { return bssrdf_sw(eta, V3(std::sqrt(fmax_(0.0f, 1.0f - cos_theta_ * cos_theta_)), 0.0f, cos_theta_)); }
}

// ---- KAT hooks for light sampling (tests/test_oracle_kats.py) ----------------------------------------------------
extern "C" {
using namespace ref;
// sample_li for n sample points u (2n floats) from the reference point (p, p_error, n); outputs wi (3n), pdf (n), L (3n)
int orc_light_sample_li(orc_scene *h, uint32_t li, const float *p, const float *perr, const float *nrm, uint32_t n, const float *u,
                        float *wi_out, float *pdf_out, float *L_out) {
    LightSampler ls; ls.init(h->scene, PT_LS_UNIFORM);
    IData ref; ref.p = V3(p[0], p[1], p[2]); ref.p_error = V3(perr[0], perr[1], perr[2]); ref.n = V3(nrm[0], nrm[1], nrm[2]);
    for (uint32_t i = 0; i < n; ++i) {
        V3 wi(0, 0, 0); Float pdf = 0.0f; IData p1;
        RGB L = ls.sample_li(li, ref, P2(u[2 * i], u[2 * i + 1]), wi, pdf, p1);
        wi_out[3 * i] = wi.x; wi_out[3 * i + 1] = wi.y; wi_out[3 * i + 2] = wi.z; pdf_out[i] = pdf;
        for (int c = 0; c < 3; ++c) L_out[3 * i + c] = L.c[c];
    }
    return 0;
}
int orc_light_pdf_li(orc_scene *h, uint32_t li, const float *p, const float *perr, const float *nrm, uint32_t n, const float *wi, float *pdf_out) {
    LightSampler ls; ls.init(h->scene, PT_LS_UNIFORM);
    IData ref; ref.p = V3(p[0], p[1], p[2]); ref.p_error = V3(perr[0], perr[1], perr[2]); ref.n = V3(nrm[0], nrm[1], nrm[2]);
    for (uint32_t i = 0; i < n; ++i) pdf_out[i] = ls.pdf_li(li, ref, V3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]));
    return 0;
}
}

// ---- BSDF of a material at a canonical interaction, for the analytic lobe checks of tests/test_oracle_bsdf.py ---------------------------------
// The material `mi` is evaluated as Material::compute_scattering_functions would at a point with geometric and shading normal +z, dpdu = +x,
// dpdv = +y, no textures; then for each of n queries: f(wo, wi) and pdf(wo, wi) over all lobes (reflection.rs:1541-1600), and sample_f(wo, u)
// (reflection.rs:1602-1689) -> sampled wi, its f, pdf and lobe type. Directions are world = local here.
extern "C" {
using namespace ref;
int orc_bsdf_eval(orc_scene *h, uint32_t mi, uint32_t n, const float *wo, const float *wi, const float *u,
                  float *f_out, float *pdf_out, float *s_wi_out, float *s_f_out, float *s_pdf_out, int32_t *s_type_out, int32_t *n_lobes_out) {
    if (mi >= h->scene.materials.size()) return 1;
    SurfaceInteraction si{};
    si.p = V3(0, 0, 0); si.p_error = V3(0, 0, 0); si.n = V3(0, 0, 1); si.sh_n = V3(0, 0, 1); si.wo = V3(0, 0, 1);
    si.dpdu = V3(1, 0, 0); si.dpdv = V3(0, 1, 0); si.sh_dpdu = V3(1, 0, 0); si.sh_dpdv = V3(0, 1, 0);
    si.uv = P2(0.5f, 0.5f); si.has_shape = false; si.shape_flip = false; si.prim = 0;
    BSDF bsdf;
    if (!material_scattering_functions(h->scene, mi, si, bsdf, nullptr, nullptr, nullptr)) return 2;
    if (n_lobes_out) *n_lobes_out = bsdf.n;
    for (uint32_t i = 0; i < n; ++i) {
        const V3 o(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]);
        if (wi && f_out && pdf_out) {
            const V3 w(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
            const RGB f = bsdf.f(o, w, BSDF_ALL);
            for (int c = 0; c < 3; ++c) f_out[3 * i + c] = f.c[c];
            pdf_out[i] = bsdf.pdf(o, w, BSDF_ALL);
        }
        if (u && s_wi_out && s_f_out && s_pdf_out) {
            V3 w(0, 0, 0); Float pdf = 0.0f; int sampled = 0;
            const RGB f = bsdf.sample_f(o, w, P2(u[2 * i], u[2 * i + 1]), pdf, BSDF_ALL, sampled);
            s_wi_out[3 * i] = w.x; s_wi_out[3 * i + 1] = w.y; s_wi_out[3 * i + 2] = w.z; s_pdf_out[i] = pdf;
            for (int c = 0; c < 3; ++c) s_f_out[3 * i + c] = f.c[c];
            if (s_type_out) s_type_out[i] = sampled;
        }
    }
    return 0;
}
}

// ---- tests/hg.rs restated (the reference's assertions on HenyeyGreenstein::p / sample_p, medium.rs:149-193), run inside the oracle ----
extern "C" {
using namespace ref;
// tests/hg.rs:12-32 sampling_match: RNG::default(), g = -0.75 .. 0.75 step 0.25, 100 samples each; returns max |p0 - p(wo, wi)| / p
double orc_test_hg_sampling_match(void) {
    HaltonTables::Pcg32 rng;   // RNG::default()
    auto uf = [&]() { return fmin_(ONE_MINUS_EPSILON, (Float)rng.uniform_int32() * 0x1.0p-32f); };
    double worst = 0.0;
    for (Float g = -0.75f; g <= 0.75f; g += 0.25f)
        for (int i = 0; i < 100; ++i) {
            const Float a = uf(), b = uf();
            const V3 wo = uniform_sample_sphere(P2(a, b));
            V3 wi;
            const Float u0 = uf(), u1 = uf();
            const Float p0 = hg_sample_p(g, wo, wi, P2(u0, u1));
            const Float p1 = phase_hg(dot(wo, wi), g);
            worst = std::max(worst, (double)std::fabs(p0 - p1) / (double)std::fabs(p1));
        }
    return worst;
}
// tests/hg.rs:34-79 sampling_orientation_forward / sample_orientation_backward: wo = (-1, 0, 0), 100 samples, counts wi.x > 0
void orc_test_hg_orientation(float g, int *nforward, int *nbackward) {
    HaltonTables::Pcg32 rng;
    auto uf = [&]() { return fmin_(ONE_MINUS_EPSILON, (Float)rng.uniform_int32() * 0x1.0p-32f); };
    *nforward = *nbackward = 0;
    for (int i = 0; i < 100; ++i) {
        const Float u0 = uf(), u1 = uf();
        V3 wi;
        hg_sample_p(g, V3(-1.0f, 0.0f, 0.0f), wi, P2(u0, u1));
        if (wi.x > 0.0f) ++*nforward; else ++*nbackward;
    }
}
// tests/hg.rs:81-103 normalized: per g, the mean of p(wo, wi) over 100 000 uniform directions (expected 1 / 4 pi)
void orc_test_hg_normalized(double *means7) {
    HaltonTables::Pcg32 rng;
    auto uf = [&]() { return fmin_(ONE_MINUS_EPSILON, (Float)rng.uniform_int32() * 0x1.0p-32f); };
    int k = 0;
    for (Float g = -0.75f; g <= 0.75f; g += 0.25f, ++k) {
        const Float a = uf(), b = uf();
        const V3 wo = uniform_sample_sphere(P2(a, b));
        Float sum = 0.0f;
        const int n = 100000;
        for (int i = 0; i < n; ++i) { const Float c = uf(), d = uf(); sum += phase_hg(dot(wo, uniform_sample_sphere(P2(c, d))), g); }
        means7[k] = (double)(sum / (Float)n);
    }
}
}  // extern "C"
