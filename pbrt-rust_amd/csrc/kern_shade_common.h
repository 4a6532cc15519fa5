// kern_shade_common.h -- split out of the former single-file kernels.hip so that the translation units compile in parallel.
#pragma once
#include "kern_common.h"
// ---- shading ---------------------------------------------------------------------------------------------

// ---- optional region timers (build with -DPT_REGION_PROFILE): wave time between markers is charged to the region of the
// previous marker; one lane per wave updates three LDS words. Printed by the host when the scene is destroyed.
#ifdef PT_REGION_PROFILE
struct Prof { long long *t; int *r; unsigned long long *acc; };
#define PT_T(k) do { if ((int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) { const long long _n = clock64(); const int _w = threadIdx.x >> 6; \
    prof.acc[_w * 16 + prof.r[_w]] += (unsigned long long)(_n - prof.t[_w]); prof.t[_w] = _n; prof.r[_w] = (k); } } while (0)
#define PT_PROF_ARG , Prof prof
#define PT_PROF_PASS , prof
#else
#define PT_T(k) do {} while (0)
#define PT_PROF_ARG
#define PT_PROF_PASS
#endif

// Resolve the pending next-event estimation of the previous vertex once its shadow / MIS rays are traced
// (integrator.rs:150-171,199-233): L += beta_at_nee * Ld / choice_pdf.
template <bool SPH, bool VOL = false> PT_DEV void resolve_pending(const DeviceScene &s, const PathSoA &ps, uint32_t pid, uint32_t &flags, RGB &L,
                                                unsigned long long &zero_num, unsigned long long &n_bytes PT_PROF_ARG) {
    if (!(flags & (PF_PEND_SHADOW | PF_PEND_MIS))) return;
    PT_T(1);
    n_bytes += 4 + 4 + 12 + ((flags & PF_PEND_SHADOW) ? 1 + 12 : 0) + ((flags & PF_PEND_MIS) ? 12 + 4 + 12 + 12 + 8 : 0);  // nee_light, choice pdf, nb, occluded+A, MIS record
    RGB Ld(0.0f);
    const uint32_t li = ps.nee_light(pid);
    // volpath: VisibilityTester::tr intersects (closest hit) and every surface is opaque; the segment's transmittance is already in A
    if ((flags & PF_PEND_SHADOW) && (VOL ? ps.sh_prim(pid) == PT_NONE : !ps.occluded(pid))) Ld = Ld + RGB(ps.A_r(pid), ps.A_g(pid), ps.A_b(pid));
    if (flags & PF_PEND_MIS) {
        const PtLight &Lt = s.lights[li];
        V3 wi(ps.mis_dx(pid), ps.mis_dy(pid), ps.mis_dz(pid));
        RGB lrad(0.0f);
        const uint32_t mp = ps.mis_prim(pid);
        if (mp != PT_NONE) {
            if (s.prim_light[mp] == li) {  // Arc::ptr_eq(light), integrator.rs:222-228
                SurfaceInteraction lsi;
                fill_hit<SPH>(s, mp, PT_NONE, V3(ps.mis_ox(pid), ps.mis_oy(pid), ps.mis_oz(pid)), wi, ps.mis_b0(pid), ps.mis_b1(pid), ps.mis_b2(pid), lsi);  // lights are never inside instances (api.rs:1605-1608)
                lrad = area_l(Lt, lsi.n, -wi);
            }
        } else { PT_T(2); lrad = light_le(s, Lt, wi); PT_T(1); }
        if (!lrad.is_black()) {
            RGB f(ps.mis_f_r(pid), ps.mis_f_g(pid), ps.mis_f_b(pid));
            RGB Tr(1.0f);
            if (VOL) {   // Scene::intersect_tr (scene.rs:68-87): transmittance of the MIS ray's medium up to its hit (or to infinity)
                const uint32_t mm = ps.mis_medium(pid);
                if (mm != PT_NONE) Tr = Tr * medium_tr(s.media[mm], mp != PT_NONE ? ps.mis_t(pid) : PT_INF, wi);
            }
            Ld = Ld + f * lrad * Tr * ps.mis_w(pid) / ps.mis_spdf(pid);
        }
    }
    RGB nb(ps.nb_r(pid), ps.nb_g(pid), ps.nb_b(pid));
    RGB Ldb = nb * (Ld / ps.nee_choice_pdf(pid));
    if (!VOL && Ldb.is_black() && !(flags & PF_NEE_UNCOUNTED)) zero_num++;   // path.rs:142 counts only the regular vertices' NEE
    L = L + Ldb;
    flags &= ~(PF_PEND_SHADOW | PF_PEND_MIS | PF_NEE_UNCOUNTED);
}

// uniform_sample_onelight + estimate_direct (integrator.rs:81-237) at one vertex: samples the light and the BSDF,
// records the shadow / MIS rays and their weights in the path state; the estimate is summed by resolve_pending once
// both rays are traced. Returns whether anything is pending (false: Ld is black).
// VOL: estimate_direct with handle_media (integrator.rs:150-156,207-216) -- `mif` is the vertex's MediumInterface; MEDIUM: the
// vertex is a MediumInteraction (f = phase value, no cosine; si carries only p and wo).
template <bool SPH, class B, bool VOL = false, bool MEDIUM = false>
PT_DEV bool nee_vertex(const DeviceScene &s, const LightGrid &grid, const PathSoA &ps, uint32_t pid, Sampler &smp,
                                          const SurfaceInteraction &si, const IData &it, const B &bsdf, RGB beta, uint32_t &flags,
                                          bool &push_shadow, bool &push_mis, unsigned long long &n_bytes PT_PROF_ARG, MedIface mif = MedIface{PT_NONE, PT_NONE}) {
    bool nee_pending = false;
    if (s.n_lights > 0) {
        PT_T(5);
        Dist1D distrib = light_distribution_lookup(grid, s, si.p);
        float choice_pdf = 0.0f;
        const uint32_t li = (uint32_t)dist_sample_discrete(distrib, smp.get_1d(), choice_pdf);
        if (choice_pdf != 0.0f) {
            const P2 ulight = smp.get_2d();
            const P2 uscatt = smp.get_2d();
            // estimate_direct (integrator.rs:109-237), flags = All & !Specular
            const int bf = BSDF_ALL & ~BSDF_SPECULAR;
            V3 wi; float lightpdf = 0.0f, scattpdf = 0.0f; IData p1;
            PT_T(6);
            RGB Li = light_sample_li<SPH>(s, li, it, ulight, wi, lightpdf, p1);
            PT_T(7);
            const bool delta = light_is_delta(s.lights[li]);
            if (lightpdf > 0.0f && !Li.is_black()) {
                RGB f = MEDIUM ? bsdf.f(si.wo, wi, bf) : bsdf.f(si.wo, wi, bf) * abs_dot(wi, si.sh_n);
                scattpdf = bsdf.pdf(si.wo, wi, bf);
                if (!f.is_black()) {
                    V3 so, sd; spawn_ray_to(it, p1, so, sd);
                    if (VOL) {   // Li *= visibility.tr(): the unoccluded segment's transmittance (light.rs:125-150)
                        const uint32_t sm = medium_toward(mif, it.n, sd);
                        if (sm != PT_NONE) Li = Li * medium_tr(s.media[sm], 1.0f - kShadowEps, sd);
                    }
                    RGB A = delta ? f * Li / lightpdf : f * Li * power_heuristic(lightpdf, scattpdf) / lightpdf;
                    ps.sh_ox(pid) = so.x; ps.sh_oy(pid) = so.y; ps.sh_oz(pid) = so.z;
                    ps.sh_dx(pid) = sd.x; ps.sh_dy(pid) = sd.y; ps.sh_dz(pid) = sd.z;
                    ps.A_r(pid) = A.r; ps.A_g(pid) = A.g; ps.A_b(pid) = A.b;
                    flags |= PF_PEND_SHADOW; push_shadow = true; nee_pending = true; n_bytes += 24 + 12 + 4;
                }
            }
            if (!delta) {
                PT_T(8);
                int sampled_type = 0;
                RGB f = bsdf.sample_f(si.wo, wi, uscatt, scattpdf, bf, sampled_type);
                if (!MEDIUM) f = f * abs_dot(wi, si.sh_n);
                const bool sampled_specular = (sampled_type & BSDF_SPECULAR) != 0;
                if (!f.is_black() && scattpdf > 0.0f) {
                    float weight = 1.0f;
                    bool skip = false;
                    if (!sampled_specular) {
                        PT_T(9);
                        lightpdf = light_pdf_li<SPH>(s, li, it, wi);
                        PT_T(8);
                        if (lightpdf == 0.0f) skip = true;  // `return Ld` (integrator.rs:204)
                        else weight = power_heuristic(scattpdf, lightpdf);
                    }
                    if (!skip) {
                        V3 mo; spawn_ray(it, wi, mo);
                        ps.mis_ox(pid) = mo.x; ps.mis_oy(pid) = mo.y; ps.mis_oz(pid) = mo.z;
                        ps.mis_dx(pid) = wi.x; ps.mis_dy(pid) = wi.y; ps.mis_dz(pid) = wi.z;
                        ps.mis_f_r(pid) = f.r; ps.mis_f_g(pid) = f.g; ps.mis_f_b(pid) = f.b;
                        ps.mis_w(pid) = weight; ps.mis_spdf(pid) = scattpdf;
                        if (VOL) ps.mis_medium(pid) = medium_toward(mif, it.n, wi);
                        flags |= PF_PEND_MIS; push_mis = true; nee_pending = true; n_bytes += 24 + 12 + 8 + 4;
                    }
                }
            }
            if (nee_pending) {
                ps.nee_light(pid) = li; ps.nee_choice_pdf(pid) = choice_pdf;
                ps.nb_r(pid) = beta.r; ps.nb_g(pid) = beta.g; ps.nb_b(pid) = beta.b; n_bytes += 8 + 12;
            }
        }
    }
    return nee_pending;
}

// ---- textured material parameters (8f-1) ---------------------------------------------------------------------
struct TexMatEval {
    const DeviceScene &s; const TexCtx &c;
    // MixMaterial evaluates its second material on a fresh SurfaceInteraction (mix.rs:31-36): same p / uv, no differentials
    struct Plain {
        const DeviceScene &s; TexCtx c;
        PT_DEV bool bound(const PtMaterial &m, int slot) const { return m.tex[slot] >= 0; }
        PT_DEV RGB spec(const PtMaterial &m, int slot, const float *field) const { return m.tex[slot] >= 0 ? tex_eval(s, m.tex[slot], c) : RGB(field[0], field[1], field[2]); }
        PT_DEV float flt(const PtMaterial &m, int slot, float field) const { return m.tex[slot] >= 0 ? tex_eval(s, m.tex[slot], c).r : field; }
    };
    PT_DEV Plain plain() const { Plain q{s, c}; q.c.dpdx = V3(0.0f, 0.0f, 0.0f); q.c.dpdy = V3(0.0f, 0.0f, 0.0f); q.c.dudx = q.c.dvdx = q.c.dudy = q.c.dvdy = 0.0f; return q; }
    PT_DEV bool bound(const PtMaterial &m, int slot) const { return m.tex[slot] >= 0; }
    PT_DEV RGB spec(const PtMaterial &m, int slot, const float *field) const { return m.tex[slot] >= 0 ? tex_eval(s, m.tex[slot], c) : RGB(field[0], field[1], field[2]); }
    PT_DEV float flt(const PtMaterial &m, int slot, float field) const { return m.tex[slot] >= 0 ? tex_eval(s, m.tex[slot], c).r : field; }
};
// The auxiliary rays of PerspectiveCamera::generate_ray_differential (perspective.rs:143-176) after transform_ray
// (transform.rs:565-575) and Ray::scale_differential(1 / sqrt(spp)) (ray.rs:34-41, integrator.rs:340), recomputed from the
// film position (and the lens sample, Sobol' dimensions 3 and 4) instead of being carried in the path state.
PT_DEV RayDiff camera_ray_differentials(const RenderConst &rc, float pfx, float pfy, P2 plens_u, V3 ray_o, V3 ray_d) {
    const V3 pcamera = xf_point(rc.raster_to_camera, V3(pfx, pfy, 0.0f));
    const V3 dxc(rc.dx_camera[0], rc.dx_camera[1], rc.dx_camera[2]), dyc(rc.dy_camera[0], rc.dy_camera[1], rc.dy_camera[2]);
    RayDiff d; d.has = true;
    if (rc.lens_radius > 0.0f) {
        const P2 dk = concentric_sample_disk(plens_u);
        const float lx = dk.x * rc.lens_radius, ly = dk.y * rc.lens_radius;
        const V3 dx = normalize(pcamera + dxc);
        float ft = rc.focal_distance / dx.z;
        V3 pfocus = V3(0.0f, 0.0f, 0.0f) + (dx * ft);
        d.rx_o = V3(lx, ly, 0.0f); d.rx_d = normalize(pfocus - d.rx_o);
        const V3 dy = normalize(pcamera + dyc);
        ft = rc.focal_distance / dy.z;
        pfocus = V3(0.0f, 0.0f, 0.0f) + (dy * ft);
        d.ry_o = V3(lx, ly, 0.0f); d.ry_d = normalize(pfocus - d.ry_o);
    } else {
        d.rx_o = V3(0.0f, 0.0f, 0.0f); d.ry_o = V3(0.0f, 0.0f, 0.0f);
        d.rx_d = normalize(pcamera + dxc); d.ry_d = normalize(pcamera + dyc);
    }
    d.rx_o = xf_point(rc.camera_to_world, d.rx_o); d.ry_o = xf_point(rc.camera_to_world, d.ry_o);
    d.rx_d = xf_vector(rc.camera_to_world, d.rx_d); d.ry_d = xf_vector(rc.camera_to_world, d.ry_d);
    const float sc = rc.inv_sqrt_spp;
    d.rx_o = ray_o + (d.rx_o - ray_o) * sc; d.ry_o = ray_o + (d.ry_o - ray_o) * sc;
    d.rx_d = ray_d + (d.rx_d - ray_d) * sc; d.ry_d = ray_d + (d.ry_d - ray_d) * sc;
    return d;
}
