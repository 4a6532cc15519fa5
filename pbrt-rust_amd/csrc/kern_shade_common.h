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
// Grid media (VOL, DeviceScene::has_grid): the transmittance of the shadow ray and of the MIS ray is a ratio-tracking estimate that
// draws sampler dimensions (grid.rs:113-147), in the reference right after the vertex's light / scattering samples and before its
// continuation sample. Such vertices are therefore resolved in a second stage of the SAME vertex (PF_STAGE_B, kern_shade.h) and pass
// their sampler here; `smp` is NULL everywhere else.
// n_assert: the reference's `assert!(Ld.y() >= 0.0)` (path.rs:143) on the regular vertices' estimate, counted instead of panicking.
// EARLY (k_film_final): the MIS record is requested together with the NEE record, before either is used -- one round trip instead of two in a kernel whose threads walk
// their samples one after the other (the shade kernels keep the later request: batching their front cost registers and time, profiles/r4/NOTES.md).
template <bool SPH, bool VOL = false, bool EARLY = false> PT_DEV void resolve_pending(const DeviceScene &s, const PathSoA &ps, uint32_t pid, uint32_t &flags, RGB &L,
                                                unsigned long long &zero_num, uint32_t &n_assert, unsigned long long &n_bytes PT_PROF_ARG, Sampler *smp = nullptr) {
    if (!(flags & (PF_PEND_SHADOW | PF_PEND_MIS))) return;
    PT_T(1);
    // the pending records as whole quads: nee {sh_d.yz, occluded | sh_prim, nee_light} {A, choice_pdf} {nb, shadow grid medium}; mis {o, d.x} {d.yz, w, spdf} {prim, b} {f, t}
    const float4 *nq = reinterpret_cast<const float4 *>(ps.nee) + 4 * (size_t)pid;
    const float4 n1 = nq[1], n2 = nq[2], n3 = nq[3];
    float4 e0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), e1 = e0, e2 = e0, e3 = e0;
    if constexpr (EARLY) {
        if (flags & PF_PEND_MIS) { const float4 *mq = reinterpret_cast<const float4 *>(ps.mis) + 4 * (size_t)pid; e0 = mq[0]; e1 = mq[1]; e2 = mq[2]; e3 = mq[3]; }
        asm volatile("" :: "v"(n1.x), "v"(n2.x), "v"(n3.x), "v"(e0.x), "v"(e1.x), "v"(e2.x), "v"(e3.x));
    }
    n_bytes += 48 + ((flags & PF_PEND_MIS) ? 64 : 0);
    RGB Ld(0.0f);
    const uint32_t li = __float_as_uint(n1.w);
    // volpath: VisibilityTester::tr intersects (closest hit) and every surface is opaque; the segment's transmittance is already in A
    if ((flags & PF_PEND_SHADOW) && (VOL ? __float_as_uint(n1.z) == PT_NONE : __float_as_uint(n1.z) == 0u)) {
        RGB A(n2.x, n2.y, n2.z);
        if (VOL) {   // VisibilityTester::tr (light.rs:125-150) through a grid medium: estimated now (homogeneous media were folded into A at the vertex)
            const uint32_t sm = __float_as_uint(n3.w);
            if (sm != PT_NONE && s.media[sm].type == PT_MEDIUM_GRID) {
                const float4 n0 = nq[0];
                if (smp) A = A * grid_tr(s.media[sm], s.grid_aux[sm], V3(n0.x, n0.y, n0.z), V3(n0.w, n1.x, n1.y), 1.0f - kShadowEps, *smp);
                else A = RGB(0.0f);
            }
        }
        Ld = Ld + A;
    }
    if (flags & PF_PEND_MIS) {
        const float4 *mq = reinterpret_cast<const float4 *>(ps.mis) + 4 * (size_t)pid;
        const float4 m0 = EARLY ? e0 : mq[0], m1 = EARLY ? e1 : mq[1], m2 = EARLY ? e2 : mq[2], m3 = EARLY ? e3 : mq[3];
        const PtLight &Lt = s.lights[li];
        V3 wi(m0.w, m1.x, m1.y);
        RGB lrad(0.0f);
        const uint32_t mp = __float_as_uint(m2.x);
        if (mp != PT_NONE) {
            if (s.prim_light[mp] == li) {  // Arc::ptr_eq(light), integrator.rs:222-228
                SurfaceInteraction lsi;
                fill_hit<SPH>(s, mp, PT_NONE, V3(m0.x, m0.y, m0.z), wi, m2.y, m2.z, m2.w, lsi);  // lights are never inside instances (api.rs:1605-1608)
                lrad = area_l(Lt, lsi.n, -wi);
            }
        } else { PT_T(2); lrad = light_le(s, Lt, wi); PT_T(1); }
        RGB Tr(1.0f);
        bool tr_done = false;
        if (VOL) {   // Scene::intersect_tr (scene.rs:68-87) always evaluates the medium's tr -- a grid medium draws its dimensions whether or not the light is seen
            const uint32_t mm = ps.mis_medium(pid);
            if (mm != PT_NONE && s.media[mm].type == PT_MEDIUM_GRID) {
                Tr = smp ? RGB(grid_tr(s.media[mm], s.grid_aux[mm], V3(m0.x, m0.y, m0.z), wi, mp != PT_NONE ? m3.w : PT_INF, *smp)) : RGB(0.0f);
                tr_done = true;
            }
        }
        if (!lrad.is_black()) {
            RGB f(m3.x, m3.y, m3.z);
            if (VOL && !tr_done) {   // homogeneous: analytic transmittance of the MIS ray's medium up to its hit (or to infinity)
                const uint32_t mm = ps.mis_medium(pid);
                if (mm != PT_NONE) Tr = Tr * medium_tr(s.media[mm], mp != PT_NONE ? m3.w : PT_INF, wi);
            }
            Ld = Ld + f * lrad * Tr * m1.z / m1.w;
        }
    }
    RGB nb(n3.x, n3.y, n3.z);
    RGB Ldb = nb * (Ld / n2.w);
    if (!VOL && Ldb.is_black() && !(flags & PF_NEE_UNCOUNTED)) zero_num++;   // path.rs:142 counts only the regular vertices' NEE
    if (!VOL && !(flags & PF_NEE_UNCOUNTED) && !(Ldb.y() >= 0.0f)) n_assert++;   // path.rs:143 (a NaN estimate fails it too)
    L = L + Ldb;
    flags &= ~(PF_PEND_SHADOW | PF_PEND_MIS | PF_NEE_UNCOUNTED);
}

// Volpath in a scene with material-less shells (`Material "none"` + MediumInterface: how a .pbrt file bounds a medium, api.rs:597). The
// transmittance of a shadow ray is VisibilityTester::tr (light.rs:125-150) and that of a MIS ray Scene::intersect_tr (scene.rs:68-87): LOOPS
// that intersect, multiply the segment's medium transmittance in, and go on behind every surface that has no material. A wavefront
// vertex walks them one traced segment per iteration while it waits in stage B: first the whole shadow chain, then the MIS chain -- the
// order in which the reference draws the ratio-tracking dimensions of grid media (integrator.rs:150-156 before :207-216). The traced
// segment's full hit record is in the path's ext record; a shell hit spawns the next segment (push_shadow / push_mis: the caller keeps
// the path in stage B). Returns true while a chain goes on. When both have ended the nee / mis records are left as a plain, fully
// attenuated estimate (media NONE, the final MIS hit in the mis record) for resolve_pending to sum.
template <bool SPH> PT_DEV bool vol_chain_step(const DeviceScene &s, const PathSoA &ps, uint32_t pid, uint32_t flags, Sampler &smp, bool &push_shadow, bool &push_mis,
                                              unsigned long long &n_bytes) {
    float4 *nq = reinterpret_cast<float4 *>(ps.nee) + 4 * (size_t)pid;
    float4 *mq = reinterpret_cast<float4 *>(ps.mis) + 4 * (size_t)pid;
    float4 *xq = reinterpret_cast<float4 *>(ps.ext) + 8 * (size_t)pid;
    if ((flags & PF_PEND_SHADOW) && __float_as_uint(nq[1].z) != 1u) {   // nee word 6: 1 = the shadow chain has ended (A is final); the traced prim lives in ext
        const float4 n0 = nq[0], n1 = nq[1], n2 = nq[2], n3 = nq[3], h0 = xq[4], h1 = xq[5];
        n_bytes += 64 + 32;
        const uint32_t prim = __float_as_uint(h0.x), sm = __float_as_uint(n3.w);
        const bool hit = prim != PT_NONE;
        const V3 so(n0.x, n0.y, n0.z), sd(n0.w, n1.x, n1.y);
        RGB A(n2.x, n2.y, n2.z);
        bool ended = true;
        if (hit && s.prim_material[prim] != PT_NONE) A = RGB(0.0f);   // light.rs:136-138: an opaque surface, `return Spectrum::new(0.0)` -- before the segment's medium is asked
        else {
            if (sm != PT_NONE) {   // light.rs:141-143: Tr *= ray.medium.tr(ray, sampler), the segment ends at its hit (ray.t_max)
                const float t_seg = hit ? h1.y : 1.0f - kShadowEps;
                A = A * (s.media[sm].type == PT_MEDIUM_GRID ? RGB(grid_tr(s.media[sm], s.grid_aux[sm], so, sd, t_seg, smp)) : medium_tr(s.media[sm], t_seg, sd));
            }
            if (hit) {   // light.rs:146-147: ray = isect.spawn_rayto_interaction(p1)
                SurfaceInteraction hs;
                fill_hit_pkt<SPH>(s, __float_as_uint(h1.z), SPH ? __float_as_uint(h1.x) : PT_NONE, so, sd, h0.y, h0.z, h0.w, hs);
                IData a; a.p = hs.p; a.p_error = hs.p_error; a.n = hs.n;
                IData b; { const float4 x0 = xq[0], x1 = xq[1], x2 = xq[2]; b.p = V3(x0.x, x0.y, x0.z); b.p_error = V3(x1.x, x1.y, x1.z); b.n = V3(x2.x, x2.y, x2.z); }
                V3 o2, d2; spawn_ray_to(a, b, o2, d2);
                const uint32_t sm2 = medium_toward(surface_iface(s, prim, sm), hs.n, d2);   // interaction.rs:54-66 get_medium(d) of the shell's MediumInterface (primitive.rs:139-145)
                nq[0] = make_float4(o2.x, o2.y, o2.z, d2.x); nq[1] = make_float4(d2.y, d2.z, 0.0f, n1.w); nq[3] = make_float4(n3.x, n3.y, n3.z, __uint_as_float(sm2));
                push_shadow = true; ended = false; n_bytes += 48 + 4;
            }
        }
        nq[2] = make_float4(A.r, A.g, A.b, n2.w);
        if (!ended) return true;
        nq[1] = make_float4(n1.x, n1.y, __uint_as_float(1u), n1.w);
    }
    if (flags & PF_PEND_MIS) {
        const float4 m0 = mq[0], m1 = mq[1], m3 = mq[3], h0 = xq[6], h1 = xq[7];
        n_bytes += 64 + 32;
        const uint32_t prim = __float_as_uint(h0.x), mm = ps.mis_medium(pid);
        const bool hit = prim != PT_NONE;
        const V3 mo(m0.x, m0.y, m0.z), wi(m0.w, m1.x, m1.y);
        RGB f(m3.x, m3.y, m3.z);
        // scene.rs:76-79: *tr *= ray.medium.tr(ray, sampler) -- always, whatever the segment ends at
        if (mm != PT_NONE) f = f * (s.media[mm].type == PT_MEDIUM_GRID ? RGB(grid_tr(s.media[mm], s.grid_aux[mm], mo, wi, hit ? h1.y : PT_INF, smp)) : medium_tr(s.media[mm], hit ? h1.y : PT_INF, wi));
        if (hit && s.prim_material[prim] == PT_NONE) {   // scene.rs:84: ray = isect.spawn_ray(ray.d)
            SurfaceInteraction hs;
            fill_hit_pkt<SPH>(s, __float_as_uint(h1.z), SPH ? __float_as_uint(h1.x) : PT_NONE, mo, wi, h0.y, h0.z, h0.w, hs);
            IData a; a.p = hs.p; a.p_error = hs.p_error; a.n = hs.n;
            V3 o2; spawn_ray(a, wi, o2);
            ps.mis_medium(pid) = medium_toward(surface_iface(s, prim, mm), hs.n, wi);
            mq[0] = make_float4(o2.x, o2.y, o2.z, wi.x); mq[3] = make_float4(f.r, f.g, f.b, m3.w);
            push_mis = true; n_bytes += 32 + 4;
            return true;
        }
        // the chain has ended at an opaque surface or escaped: leave the record as resolve_pending reads it
        mq[2] = h0; mq[3] = make_float4(f.r, f.g, f.b, hit ? h1.y : 0.0f);
        ps.mis_medium(pid) = PT_NONE;
    }
    if (flags & PF_PEND_SHADOW) {   // the summed form: "unoccluded" with the fully attenuated (or zeroed) term, no medium left to ask
        const float4 n1 = nq[1], n3 = nq[3];
        nq[1] = make_float4(n1.x, n1.y, __uint_as_float(PT_NONE), n1.w); nq[3] = make_float4(n3.x, n3.y, n3.z, __uint_as_float(PT_NONE));
    }
    return false;
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
            RGB A(0.0f); V3 sh_o(0.0f, 0.0f, 0.0f), sh_d(0.0f, 0.0f, 0.0f);   // the shadow ray and its term, stored with the rest of the nee record below
            uint32_t sh_medium = PT_NONE;                                       // VOL: a grid medium the shadow ray travels in
            PT_T(6);
            RGB Li = light_sample_li<SPH>(s, li, it, ulight, wi, lightpdf, p1);
            PT_T(7);
            const bool delta = light_is_delta(s.lights[li]);
            if (lightpdf > 0.0f && !Li.is_black()) {
                RGB f = bsdf.f_pdf(si.wo, wi, bf, scattpdf);
                if (!MEDIUM) f = f * abs_dot(wi, si.sh_n);
                if (!f.is_black()) {
                    V3 so, sd; spawn_ray_to(it, p1, so, sd);
                    if (VOL) {   // Li *= visibility.tr(): the unoccluded segment's transmittance (light.rs:125-150)
                        const uint32_t sm = medium_toward(mif, it.n, sd);
                        if (s.has_shells) {   // the segment may end at a shell: every segment's transmittance is taken when it has been traced (vol_chain_step)
                            sh_medium = sm;
                            float4 *xq = reinterpret_cast<float4 *>(ps.ext) + 8 * (size_t)pid;
                            xq[0] = make_float4(p1.p.x, p1.p.y, p1.p.z, 0.0f); xq[1] = make_float4(p1.p_error.x, p1.p_error.y, p1.p_error.z, 0.0f); xq[2] = make_float4(p1.n.x, p1.n.y, p1.n.z, 0.0f);
                            n_bytes += 48;
                        }
                        else if (sm != PT_NONE && s.media[sm].type == PT_MEDIUM_GRID) sh_medium = sm;   // estimated when the shadow ray has been traced (resolve_pending)
                        else if (sm != PT_NONE) Li = Li * medium_tr(s.media[sm], 1.0f - kShadowEps, sd);
                    }
                    A = delta ? f * Li / lightpdf : f * Li * power_heuristic(lightpdf, scattpdf) / lightpdf;
                    sh_o = so; sh_d = sd;
                    flags |= PF_PEND_SHADOW; push_shadow = true; nee_pending = true; n_bytes += 16 + 4;   // first quad of the nee record + the shadow queue entry
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
                        float4 *mq = reinterpret_cast<float4 *>(ps.mis) + 4 * (size_t)pid;   // {o, d.x} {d.yz, w, spdf} . {f, -}; the third quad is the MIS ray's hit (k_trace)
                        mq[0] = make_float4(mo.x, mo.y, mo.z, wi.x); mq[1] = make_float4(wi.y, wi.z, weight, scattpdf); mq[3] = make_float4(f.r, f.g, f.b, 0.0f);
                        if (VOL) ps.mis_medium(pid) = medium_toward(mif, it.n, wi);
                        flags |= PF_PEND_MIS; push_mis = true; nee_pending = true; n_bytes += 48 + 4;
                    }
                }
            }
            if (nee_pending) {   // the nee record {sh_o, sh_d.x} {sh_d.yz, occluded (k_trace), nee_light} {A, choice_pdf} {nb, -}
                float4 *nq = reinterpret_cast<float4 *>(ps.nee) + 4 * (size_t)pid;
                if (flags & PF_PEND_SHADOW) nq[0] = make_float4(sh_o.x, sh_o.y, sh_o.z, sh_d.x);
                nq[1] = make_float4(sh_d.y, sh_d.z, 0.0f, __uint_as_float(li)); nq[2] = make_float4(A.r, A.g, A.b, choice_pdf); nq[3] = make_float4(beta.r, beta.g, beta.b, __uint_as_float(sh_medium));
                n_bytes += 48;
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
