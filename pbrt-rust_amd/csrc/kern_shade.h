// kern_shade.h -- split out of the former single-file kernels.hip so that the translation units compile in parallel.
#pragma once
#include "kern_shade_common.h"
#ifndef PT_SHADE_ATTR
#define PT_SHADE_ATTR   // experiment hook: e.g. __attribute__((amdgpu_waves_per_eu(4,4)))
#endif
// MODE: 0 = triangle-only scenes, 1 = general geometry (spheres and/or instances), 2 = general geometry + textures
// DIFF: 1 = the launch serves class 0 (matte materials: Lambertian / Oren-Nayar lobes only); 2 = class 6 (mirror / smooth glass: perfectly
//       specular lobes only, no next-event estimation); 0 = any material of the lobe budget MAXL
template <int MAXL, int MODE, int DIFF>
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 1   // experiment hook (tools/build_variant.sh -DPT_SHADE_WAVES=N): minimum waves per SIMD the one-lobe kernels are compiled for
#endif
// Waves per SIMD the kernel is compiled for. The matte kernel of triangle-only scenes (MAXL 1, MODE 0, DIFF 1: the headline's) takes three:
// 168 VGPRs + 32 bytes of scratch, VALU-bound at two waves (57 % busy, 9 % of the wave cycles waiting on memory: SQ counters in profiles/r3).
// The others lose more to spills than they gain (one-lobe general kernel at three waves: 160 bytes of scratch, 112.8 -> 114.8 ms on C3).
#ifndef PT_P2_WAVES
#define PT_P2_WAVES 3     // the plastic-like two-lobe kernel (DIFF 4) of triangle-only scenes: three waves per SIMD (168 VGPRs + 80 B of scratch) with 28 Sobol' dimensions staged
#define PT_P2_DIMS 28u    // in LDS and 512-entry queues (51 KB per workgroup: three fit a CU) -- C3 shade_2lobe 101.0 -> 86.9 ms against two waves with 56 dimensions / 1024 entries
#define PT_P2_QCAP 512
#endif
#ifndef PT_METAL_WAVES
#define PT_METAL_WAVES 3   // experiment hook: waves per SIMD of the metal-only one-lobe kernel (DIFF 3) of triangle-only scenes
#endif
__global__ __launch_bounds__(256, (MAXL == 2 && MODE == 0 && DIFF == 4) ? PT_P2_WAVES : (MAXL == 1 && MODE == 0 && DIFF == 3) ? PT_METAL_WAVES : (MAXL == 1 && MODE == 0 && DIFF == 1) ? (PT_SHADE_WAVES > 3 ? PT_SHADE_WAVES : 3) : (MAXL == 1 && MODE == 0 && DIFF == 2) ? 4 : (MAXL == 1 && MODE == 1) ? (PT_SHADE_WAVES > 2 ? PT_SHADE_WAVES : 2) : (MAXL == 1 ? PT_SHADE_WAVES : (MAXL >= 2 && MODE < 2) ? 2 : 1)) PT_SHADE_ATTR void k_shade(DeviceScene s, RenderConst rc, SobolTables tabs, LightGrid grid, PathSoA ps, ShadeJob job) {
    constexpr bool SPH = MODE >= 1, TEX = MODE >= 2, VOL = MODE == 3;   // MODE 3: general + textures + participating media (volpath.rs)
    // Sobol' nibble tables of the first dimensions (dev_sampler.h): 56 cover the vertices of bounces 0..5; the five-lobe class, whose lobe store
    // fills the LDS, keeps 20 and reads the rest from HBM (one 64-byte line per look-up, the same for every lane of a bounce)
    constexpr uint32_t LDS_DIMS = MAXL == 5 ? 20u : DIFF == 2 ? 32u : (MAXL == 2 && MODE == 0 && DIFF == 4) ? PT_P2_DIMS : 56u;   // (the specular-only kernel runs four workgroups per CU: 37 KB each)
    __shared__ uint32_t s_sobol[LDS_DIMS * kSobolNibWords];
    // Block-level queues on purpose: their barriers keep the four waves of a block in lockstep through this very large
    // kernel, which measured 10 % faster than barrier-free per-wave queues (WaveQueue) at the same occupancy.
    // (the five-lobe class flushes its queues every round -- 256-entry buffers: with the 60 KB lobe store, two of its workgroups fit a CU's
    //  LDS; its vertices cost ~0.8 ns each, so the extra global atomics, one per queue and 256 vertices, do not show)
    constexpr int QCAP = MAXL == 5 ? 256 : (MAXL == 2 && MODE == 0 && DIFF == 4) ? PT_P2_QCAP : 1024;
    __shared__ LdsQueue<QCAP> s_qext, s_qres, s_qsh, s_qmis;
    constexpr bool SSS = MAXL == 5 || DIFF == 6;   // the launch can hold subsurface materials: the five-lobe class, or its smooth-dielectric-only form (DIFF 6: one FresnelSpecular lobe + the BSSRDF)
    __shared__ LdsQueue<SSS ? QCAP : 1> s_qprobe;
    __shared__ LdsQueue<(MODE == 3) ? QCAP : 1> s_qself;   // volpath with grid media: vertices waiting for stage B, back into this class's next queue
    __shared__ uint32_t s_hist[16];
    __shared__ uint32_t s_bins[16];
    __shared__ float s_lobes[lobe_store_words<MAXL>()];   // the two- and five-lobe classes keep their BxDFs here (dev_bsdf.h)
    lq_init(s_qext); lq_init(s_qres); lq_init(s_qsh); lq_init(s_qmis); lq_init(s_qprobe); lq_init(s_qself);
    if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
#ifdef PT_REGION_PROFILE
    __shared__ long long s_pt[4]; __shared__ int s_pr[4]; __shared__ unsigned long long s_pacc[64];
    if (threadIdx.x < 64) s_pacc[threadIdx.x] = 0;
    if (threadIdx.x < 4) { s_pt[threadIdx.x] = clock64(); s_pr[threadIdx.x] = 15; }
    Prof prof{s_pt, s_pr, s_pacc};
#endif
    sobol_stage_lds(s_sobol, tabs.nib, LDS_DIMS, threadIdx.x, blockDim.x);
    __syncthreads();
    const uint32_t count = *job.count;
    const uint32_t rounded = (count + 255u) & ~255u;   // whole blocks iterate together
    unsigned long long zero_num = 0, zero_den = 0, n_valid = 0, n_bytes = 0;  // n_bytes: path-state + queue bytes
    uint32_t n_assert = 0;   // assert!()s of the reference that would have fired at this thread's vertices (PtCounters::reference_asserts)
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < rounded; qi += gridDim.x * blockDim.x) {
    const bool valid = qi < count;
    bool push_ext = false, push_resolve = false, push_shadow = false, push_mis = false, push_probe = false, push_self = false;
    int finished_bounces = -1;
    uint32_t pid = 0, ext_oct = 0;   // direction octant of the continuation ray
    PT_T(0);
    if (valid) {
        n_valid++;
        n_bytes += 4 + 48 + /* write back */ 32;   // queue entry, three quads of the core record read, two written
        pid = job.queue[qi];
        // the core record as whole quads: {L, etascale} {beta, meta} {sobol index, pfilm}
        const float4 *cq = reinterpret_cast<const float4 *>(ps.core) + 4 * (size_t)pid;
        const float4 c0 = cq[0], c1 = cq[1], c2 = cq[2];
        uint32_t meta = __float_as_uint(c1.w);
        uint32_t flags = meta >> 24, bounces = (meta >> 16) & 0xffu;
        float etascale = c0.w;
        Sampler smp; smp.index = (uint64_t)__float_as_uint(c2.x) | ((uint64_t)__float_as_uint(c2.y) << 32); smp.dim = meta & 0xffffu; smp.m32 = tabs.m32; smp.nib = tabs.nib; smp.lds = s_sobol; smp.lds_dims = LDS_DIMS; smp.overflow = false; smp.halton = MODE >= 1 && rc.halton.enabled != 0;   /* Halton scenes run the general kernels: the triangle-only ones stay Sobol'-only */ smp.prime = tabs.prime; smp.prime_sum = tabs.prime_sum; smp.perm = tabs.perm;
        smp.base = 0xffffffffu;
        RGB L(c0.x, c0.y, c0.z);
        RGB beta(c1.x, c1.y, c1.z);

        // -- resolve the previous vertex's next-event estimation (integrator.rs:150-171,199-233). Volpath with grid media: the pending
        //    estimate is THIS vertex's own (stage B): its shadow / MIS rays have been traced, their ratio-tracking transmittances draw
        //    the sampler dimensions that follow the vertex's light and scattering samples, then the vertex goes on to sample its BSDF
        const bool stage_b = VOL && (flags & PF_STAGE_B) != 0u;
        const uint32_t camera_ray_flag = flags & PF_CAMERA_RAY;   // (a deferred vertex rebuilds its BSDF in stage B with the same differentials)
        if (stage_b) smp.load_window();
        // (scenes with material-less shells: the pending estimate's rays are walked segment by segment first, vol_chain_step)
        if (VOL && stage_b && s.has_shells) { if (!vol_chain_step<SPH>(s, ps, pid, flags, smp, push_shadow, push_mis, n_bytes)) resolve_pending<SPH, VOL>(s, ps, pid, flags, L, zero_num, n_assert, n_bytes PT_PROF_PASS, nullptr); }
        else resolve_pending<SPH, VOL>(s, ps, pid, flags, L, zero_num, n_assert, n_bytes PT_PROF_PASS, stage_b ? &smp : nullptr);
        flags &= ~PF_STAGE_B;

        PT_T(3);
        if (flags & PF_DEAD) {
            finished_bounces = (int)bounces;
        } else {
            n_bytes += 32 + 32;  // ray + hit record
            float4 *rq = reinterpret_cast<float4 *>(ps.ray) + 2 * (size_t)pid;                  // {o, d.x} {d.yz, -, -}
            const float4 *hq = reinterpret_cast<const float4 *>(ps.hit) + 2 * (size_t)pid;      // {prim, b0, b1, b2} {inst, t, packet, packet flags}
            const float4 r0 = rq[0], r1 = rq[1], h0 = hq[0], h1 = hq[1];
            V3 ro(r0.x, r0.y, r0.z), rd(r0.w, r1.x, r1.y);
            const uint32_t hp = __float_as_uint(h0.x);
            const bool found = hp != PT_NONE;
            SurfaceInteraction si;
            uint32_t pfl = 0;
            if (found) n_bytes += 48;   // the hit's TriPacket: vertices, ids, flags
            if (found) pfl = fill_hit_pkt<SPH>(s, __float_as_uint(h1.z), SPH ? __float_as_uint(h1.x) : PT_NONE, ro, rd, h0.y, h0.z, h0.w, si);
            // path.rs:106-117
            if (!stage_b && (bounces == 0 || (flags & PF_SPECULAR))) {
                if (found) {
                    const uint32_t al = s.prim_light[hp];
                    L = L + (al != PT_NONE ? area_l(s.lights[al], si.n, -rd) : RGB(0.0f)) * beta;   // isect.le() of a non-emissive primitive is 0, and `L += beta * 0` still happens (path.rs:110): NaN for a non-finite beta
                } else {
                    for (uint32_t k = 0; k < s.n_infinite; ++k) L = L + light_le(s, s.lights[s.infinite_lights[k]], rd) * beta;
                }
            }
            bool terminated = !found || bounces >= rc.max_depth;  // path.rs:120
            if (!terminated) {
                PT_T(4);
                if (DIFF == 2) smp.load_window3(); else if (!stage_b) smp.load_window();
                PT_T(10);
                Bsdf<MAXL, DIFF> bsdf; bsdf.bind(s_lobes);
                const uint32_t mi = packet_material(s, pfl, hp);
                bool has_bsdf = false;
                RGB bss_sa(0.0f), bss_ss(0.0f);   // subsurface.rs:100-101: sigma_a / sigma_s textures, evaluated with the BSDF's parameters
                const bool is_sss = SSS && mi != PT_NONE && s.materials[mi].type == PT_MAT_SUBSURFACE;
                if (TEX) {
                    // compute_scattering_functions -> compute_differentials(ray) (interaction.rs:262-342): only the camera ray
                    // carries differentials; every spawned ray has none
                    RayDiff rdiff; rdiff.has = false;
                    if (flags & PF_CAMERA_RAY) {
                        P2 plens_u(0.0f, 0.0f);
                        if (rc.lens_radius > 0.0f) plens_u = rc.halton.enabled ? P2(halton_sample_dimension(tabs, rc.halton, smp.index, 3u), halton_sample_dimension(tabs, rc.halton, smp.index, 4u))
                                                                              : P2(smp.peek_sobol(3u), smp.peek_sobol(4u));
                        rdiff = camera_ray_differentials(rc, c2.z, c2.w, plens_u, ro, rd);
                    }
                    const TexCtx tctx = compute_differentials(si, rdiff);
                    if (mi != PT_NONE && s.materials[mi].tex[PT_MP_BUMP] >= 0) {   // bump() (core/material.rs:46-87)
                        const int dtex = s.materials[mi].tex[PT_MP_BUMP];
                        TexCtx e = tctx;
                        float du = 0.5f * (fabsf(tctx.dudx) + fabsf(tctx.dudy));
                        if (du == 0.0f) du = 0.0005f;
                        e.p = si.p + si.sh_dpdu * du; e.uv = P2(si.uv.x + du, si.uv.y + 0.0f);
                        const float udisplace = tex_eval(s, dtex, e).r;
                        float dv = 0.5f * (fabsf(tctx.dvdx) + fabsf(tctx.dvdy));
                        if (dv == 0.0f) dv = 0.0005f;
                        e.p = si.p + si.sh_dpdv * dv; e.uv = P2(si.uv.x + 0.0f, si.uv.y + dv);
                        const float vdisplace = tex_eval(s, dtex, e).r;
                        const float displace = tex_eval(s, dtex, tctx).r;
                        const V3 bdpdu = si.sh_dpdu + si.sh_n * ((udisplace - displace) / du) + si.sh_dndu * displace;
                        const V3 bdpdv = si.sh_dpdv + si.sh_n * ((vdisplace - displace) / dv) + si.sh_dndv * displace;
                        si.sh_n = normalize(cross(bdpdu, bdpdv));   // set_shading_geometry(.., false), interaction.rs:228-249
                        if (si.has_shape) { if (si.shape_flip) si.sh_n = -si.sh_n; si.sh_n = face_forward(si.sh_n, si.n); }
                        si.sh_dpdu = bdpdu; si.sh_dpdv = bdpdv;
                    }
                    const TexMatEval E{s, tctx};
                    has_bsdf = (mi != PT_NONE) && build_bsdf(s.materials[mi], si, bsdf, E, s.materials);
                    if (is_sss) {
                        const PtMaterial &sm = s.materials[mi];
                        if (sm.kd_subsurface) {   // kdsubsurface.rs:96-99: mfp * scale and Kd from their textures, converted here
                            const RGB mfree = E.spec(sm, PT_MP_MFP, sm.mfp).clamps(0.0f, PT_INF) * RGB(sm.scale), kd = E.spec(sm, PT_MP_KD, sm.kd).clamps(0.0f, PT_INF);
                            subsurface_from_diffuse(s.bss_tables[sm.bssrdf_table], kd, mfree, bss_sa, bss_ss);
                        } else { bss_sa = E.spec(sm, PT_MP_SIGMA_A, sm.sigma_a); bss_ss = E.spec(sm, PT_MP_SIGMA_S, sm.sigma_s); }
                    }
                } else {
                    has_bsdf = (mi != PT_NONE) && build_bsdf(s.materials[mi], si, bsdf, ConstMatEval(), s.materials);
                    if (is_sss) {
                        const PtMaterial &sm = s.materials[mi];
                        if (sm.kd_subsurface) subsurface_from_diffuse(s.bss_tables[sm.bssrdf_table], rgb3(sm.kd).clamps(0.0f, PT_INF), rgb3(sm.mfp).clamps(0.0f, PT_INF) * RGB(sm.scale), bss_sa, bss_ss);
                        else { bss_sa = rgb3(sm.sigma_a); bss_ss = rgb3(sm.sigma_s); }
                    }
                }
                flags &= ~PF_CAMERA_RAY;
                IData it; it.p = si.p; it.p_error = si.p_error; it.n = si.n;
                MedIface mif{PT_NONE, PT_NONE};
                if (VOL) mif = surface_iface(s, hp, ps.medium(pid));   // primitive.rs:139-145
                if (!has_bsdf) {  // path.rs:124-129: skip the surface, bounces unchanged
                    V3 o; spawn_ray(it, rd, o);
                    rq[0] = make_float4(o.x, o.y, o.z, rd.x);
                    if (VOL) {   // volpath.rs:127-131 `bounces -= 1; continue`: the count drops by one and wraps below zero
                        ps.medium(pid) = medium_toward(mif, si.n, rd);
                        bounces = (bounces - 1u) & 0xffu;
                    }
                    push_ext = true; ext_oct = (rd.x < 0.0f ? 1u : 0u) | (rd.y < 0.0f ? 2u : 0u) | (rd.z < 0.0f ? 4u : 0u);
                } else {
                    const V3 wo = -rd;  // path.rs:148; estimate_direct uses isect.wo (== -rd for triangles, triangle.rs:296)
                    // uniform_sample_onelight (integrator.rs:81-106)
                    bool defer = false;
                    if (VOL) {
                        if (!stage_b) {
                            // volpath.rs:136-138: unconditional. `L += beta * Ld` happens even when Ld is black: a throughput that has gone infinite or
                            // NaN poisons the sample there (0 x inf), which integrator.rs:350-368 then zeroes and counts
                            if (!nee_vertex<SPH, Bsdf<MAXL, DIFF>, true, false>(s, grid, ps, pid, smp, si, it, bsdf, beta, flags, push_shadow, push_mis, n_bytes PT_PROF_PASS, mif)) L = L + beta * RGB(0.0f);
                        }
                        // wait for the traced rays before drawing any further dimension (stage A: the vertex's shadow / MIS rays; stage B in a
                        // scene with shells: the next segment of one of them)
                        defer = (s.has_grid != 0u || s.has_shells != 0u) && (push_shadow || push_mis);
                    }
                    else if (DIFF != 2 && bsdf.num_components(BSDF_ALL & ~BSDF_SPECULAR) > 0) {   // (a specular-only BSDF has no such component: path.rs:131)
                        zero_den++;
                        const bool nee_pending = nee_vertex<SPH>(s, grid, ps, pid, smp, si, it, bsdf, beta, flags, push_shadow, push_mis, n_bytes PT_PROF_PASS);
                        if (!nee_pending) { zero_num++; const RGB Ld0 = beta * RGB(0.0f); if (!(Ld0.y() >= 0.0f)) n_assert++; /* path.rs:143 */ L = L + Ld0; }  // Ld is black (path.rs:142); `L += beta * Ld` all the same (path.rs:140-145): NaN for a non-finite beta (fuzz seed 13269)
                    }
                    // path.rs:148-174: sample the BSDF for the next direction
                    PT_T(11);
                    V3 wi; float pdf = 0.0f; int sflags = 0;
                    RGB f(0.0f);
                    if (defer) { flags |= PF_STAGE_B | camera_ray_flag; push_self = true; }
                    else f = bsdf.sample_f(wo, wi, smp.get_2d(), pdf, BSDF_ALL, sflags);
                    if (defer) { /* stage B samples on */ }
                    else if (f.is_black() || pdf == 0.0f) terminated = true;
                    else {
                        beta = beta * (f * abs_dot(wi, si.sh_n) / pdf);
                        { const float by = beta.y(); if (!VOL && !(by >= 0.0f)) n_assert++; if (__builtin_isinf(by)) n_assert++; }   // path.rs:162-163, volpath.rs:176
                        if (sflags & BSDF_SPECULAR) flags |= PF_SPECULAR; else flags &= ~PF_SPECULAR;
                        if ((sflags & BSDF_SPECULAR) && (sflags & BSDF_TRANSMISSION)) {
                            const float eta = bsdf.eta;
                            etascale *= (dot(wo, si.n) > 0.0f) ? eta * eta : 1.0f / (eta * eta);
                        }
                        V3 o; spawn_ray(it, wi, o);
                        bool rr_kill = false, to_probe = false;
                        if constexpr (SSS) {
                            // path.rs:177-183: importance sample the BSSRDF; the probe chain of sample_sp (bssrdf.rs:367-395)
                            // is walked by k_bssrdf over the following wavefront iterations
                            if ((s.materials[mi].type == PT_MAT_SUBSURFACE || disney_has_bssrdf(s.materials[mi])) && (sflags & BSDF_TRANSMISSION)) {
                                // path.rs:181-183 draws s2 then s1; volpath.rs:191 `sample_s(scene, sampler.get_1d(), &sampler.get_2d(), ..)` the other way round
                                P2 s2; float s1;
                                if (VOL) { s1 = smp.get_1d(); s2 = smp.get_2d(); } else { s2 = smp.get_2d(); s1 = smp.get_1d(); }
                                if (__builtin_isinf(beta.y())) n_assert++;   // path.rs:184 / volpath.rs:194: evaluated after sample_s whatever it returned
                                DevBssrdf bss;
                                if (is_sss) bss.init_medium(s.materials[mi], s.bss_tables, bss_sa, bss_ss); else bss.init_disney(s.materials[mi]);
                                bss.init_frame(si);
                                V3 start, target; float u1n = 0.0f;
                                const BssSoA &bs = job.bs;
                                if (!bss.probe_segment(s1, s2, start, target, u1n)) rr_kill = true;   // S is black: `break`
                                else {
                                    const V3 pd = target - start;
                                    if (pd.x == 0.0f && pd.y == 0.0f && pd.z == 0.0f) rr_kill = true;  // empty chain: nfound == 0
                                    else {
                                        // the probe state as whole quads (kernels.h: BssSoA): probe 32 B, frame 48 B, coef 32 B only where sigma_a / sigma_s are not the material's constants
                                        float4 *const pq = bs.probe + (size_t)pid * BssSoA::kProbeQuads, *const fq = bs.frame + (size_t)pid * BssSoA::kFrameQuads;
                                        pq[0] = make_float4(start.x, start.y, start.z, u1n); pq[1] = make_float4(target.x, target.y, target.z, __uint_as_float(mi));
                                        fq[0] = make_float4(si.p.x, si.p.y, si.p.z, __uint_as_float(0u)); fq[1] = make_float4(bss.ns.x, bss.ns.y, bss.ns.z, __uint_as_float(0xffffffffu));
                                        fq[2] = make_float4(bss.ss.x, bss.ss.y, bss.ss.z, __uint_as_float(mi));
                                        n_bytes += 32 + 48;
                                        if (is_sss && bss_coef_stored(s.n_textures, s.materials[mi])) {
                                            float4 *const cq = bs.coef + (size_t)pid * BssSoA::kCoefQuads;
                                            cq[0] = make_float4(bss_sa.r, bss_sa.g, bss_sa.b, 0.0f); cq[1] = make_float4(bss_ss.r, bss_ss.g, bss_ss.b, 0.0f);
                                            n_bytes += 32;
                                        }
                                        // base = {p: start, p_error: 0, n: 0}: spawn_rayto_point leaves the origin at `start`
                                        rq[0] = make_float4(start.x, start.y, start.z, pd.x); rq[1] = make_float4(pd.y, pd.z, 0.0f, 0.0f);
                                        to_probe = true; push_probe = true; n_bytes += 24 + 4;
                                    }
                                }
                            }
                        }
                        // path.rs:206-214 Russian roulette
                        RGB rrbeta = beta * etascale;
                        if (!to_probe && !rr_kill && rrbeta.max_component_value() < rc.rr_threshold && bounces > 3) {
                            const float q = maxf(1.0f - rrbeta.max_component_value(), 0.05f);
                            if (smp.get_1d() < q) rr_kill = true;
                            else { beta = beta / (1.0f - q); if (__builtin_isinf(beta.y())) n_assert++; }   // path.rs:213, volpath.rs:223
                        }
                        if (rr_kill) terminated = true;
                        else if (!to_probe) {
                            bounces += 1;
                            rq[0] = make_float4(o.x, o.y, o.z, wi.x); rq[1] = make_float4(wi.y, wi.z, 0.0f, 0.0f);
                            if (VOL) ps.medium(pid) = medium_toward(mif, si.n, wi);   // isect.spawn_ray(wi) (interaction.rs:32-36,54-66)
                            push_ext = true; n_bytes += 32 + 4;  // new ray record, ext queue entry
                            ext_oct = (wi.x < 0.0f ? 1u : 0u) | (wi.y < 0.0f ? 2u : 0u) | (wi.z < 0.0f ? 4u : 0u);
                        }
                    }
                }
            }
            if (terminated) {
                if (flags & (PF_PEND_SHADOW | PF_PEND_MIS)) { flags |= PF_DEAD; push_resolve = true; }
                else { finished_bounces = (int)bounces; flags |= PF_FINISHED; }
            }
        }
        PT_T(12);
        if (smp.overflow) atomicMax(job.error, (uint32_t)PT_ERR_SOBOL_DIMENSIONS);
        float4 *cw = reinterpret_cast<float4 *>(ps.core) + 4 * (size_t)pid;
        cw[0] = make_float4(L.r, L.g, L.b, etascale);
        cw[1] = make_float4(beta.r, beta.g, beta.b, __uint_as_float((smp.dim & 0xffffu) | ((bounces & 0xffu) << 16) | (flags << 24)));
    }
    PT_T(13);
#ifdef PT_BIN_EXT
    lq_push_binned(s_qext, s_bins, pid, push_ext, ext_oct);
#else
    lq_push(s_qext, pid, push_ext);
#endif
    lq_push(s_qres, pid, push_resolve && job.shade_next0 != nullptr);   // (no miss pass: the film kernel ends the dead paths, k_film_final)
    lq_push(s_qsh, pid, push_shadow);
    lq_push(s_qmis, pid, push_mis);
    if (finished_bounces >= 0) atomicAdd(&s_hist[finished_bounces > 15 ? 15 : finished_bounces], 1u);  // path.rs:219 (LDS)
    if constexpr (SSS) if (job.probe_next) lq_push(s_qprobe, pid, push_probe);
    if constexpr (MODE == 3) lq_push(s_qself, pid, push_self);
    __syncthreads();
    lq_flush_nosync(s_qext, job.ext_next_count, job.ext_next, 256u, false);
    lq_flush_nosync(s_qres, job.shade_next0_count, job.shade_next0, 256u, false);
    lq_flush_nosync(s_qsh, job.shadow_count, job.shadow, 256u, false);
    lq_flush_nosync(s_qmis, job.mis_count, job.mis, 256u, false);
    if constexpr (SSS) if (job.probe_next) lq_flush_nosync(s_qprobe, job.probe_next_count, job.probe_next, 256u, false);
    if constexpr (MODE == 3) lq_flush_nosync(s_qself, job.self_next_count, job.self_next, 256u, false);
    __syncthreads();
    }  // persistent loop over the queue
    lq_flush_nosync(s_qext, job.ext_next_count, job.ext_next, 0u, true);
    lq_flush_nosync(s_qres, job.shade_next0_count, job.shade_next0, 0u, true);
    lq_flush_nosync(s_qsh, job.shadow_count, job.shadow, 0u, true);
    lq_flush_nosync(s_qmis, job.mis_count, job.mis, 0u, true);
    if constexpr (SSS) if (job.probe_next) lq_flush_nosync(s_qprobe, job.probe_next_count, job.probe_next, 0u, true);
    if constexpr (MODE == 3) lq_flush_nosync(s_qself, job.self_next_count, job.self_next, 0u, true);
    __syncthreads();
    __syncthreads();   // s_hist complete
#ifdef PT_REGION_PROFILE
    PT_T(14);
    __syncthreads();
    if (threadIdx.x < 16) atomicAdd(&job.counters->regions[threadIdx.x], s_pacc[threadIdx.x] + s_pacc[16 + threadIdx.x] + s_pacc[32 + threadIdx.x] + s_pacc[48 + threadIdx.x]);
#endif
    if (threadIdx.x < 16 && s_hist[threadIdx.x]) atomicAdd(&job.counters->path_len[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
    counter_add(&job.counters->zero_num, zero_num);
    counter_add(&job.counters->zero_den, zero_den);
    counter_add(&job.counters->ref_asserts, (unsigned long long)n_assert);
    counter_add(&job.counters->stages, n_valid);
    counter_add(&job.counters->shade_items[job.cls], n_valid);
    counter_add(&job.counters->shade_bytes[job.cls], n_bytes);
}
