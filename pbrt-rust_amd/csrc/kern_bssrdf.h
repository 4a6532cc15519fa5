// kern_bssrdf.h -- split out of the former single-file kernels.hip so that the translation units compile in parallel.
#pragma once
#include "kern_shade_common.h"
// ---- subsurface scattering: probe chains + exit-point vertex (path.rs:177-204, bssrdf.rs:334-410,559-574) --------------
// One launch per wavefront iteration in which probe chains ran. Each queue entry is a path whose whole chain was walked by
// k_trace<.., PROBE> in the launch before (kern_trace.h); this kernel finishes the vertex at the selected exit point.
// When the exit point pi is reached the lane finishes the vertex: resolve po's next-event estimation, beta *= S / pdf,
// NEE at pi through the adapter BSDF, sample the adapter BSDF, Russian roulette, bounces += 1.
// VOL: under the volumetric integrator (volpath.rs:186-214) -- the estimate at pi handles media (shadow / MIS rays start in the medium pi's
// MediumInterface names for their side; the interface is the one the probe chain handed on: BssSoA::iface), and the new ray carries its medium.
template <bool SPH, bool VOL>
#ifndef PT_BSSRDF_WAVES
#define PT_BSSRDF_WAVES ((SPH || VOL) ? 1 : 3)   // waves per SIMD the kernel is compiled for. Triangle-only scenes: three (168 VGPRs + 64 bytes of scratch instead of 197: C5 41.1 -> 35.7 ms
                                        // per 216-sample pass; like the matte shade kernel it is VALU-bound at two waves); with spheres / instances three waves spill 192 bytes: left alone
#endif
__global__ __launch_bounds__(256, PT_BSSRDF_WAVES) void k_bssrdf(DeviceScene s, RenderConst rc, SobolTables tabs, LightGrid grid, PathSoA ps, BssrdfJob job) {
    constexpr uint32_t LDS_DIMS = 56u;
    __shared__ uint32_t s_sobol[LDS_DIMS * kSobolNibWords];
    __shared__ LdsQueue<1024> s_qext, s_qres, s_qsh, s_qmis;
    __shared__ LdsQueue<VOL ? 1024 : 1> s_qself;   // volpath, grid media / shells: exit-point vertices waiting for stage B
    __shared__ uint32_t s_hist[16];
    lq_init(s_qext); lq_init(s_qres); lq_init(s_qsh); lq_init(s_qmis); lq_init(s_qself);
    if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
    sobol_stage_lds(s_sobol, tabs.nib, LDS_DIMS, threadIdx.x, blockDim.x);
    __syncthreads();
#ifdef PT_REGION_PROFILE
    __shared__ long long s_pt[4]; __shared__ int s_pr[4]; __shared__ unsigned long long s_pacc[64];
    if (threadIdx.x < 4) { s_pt[threadIdx.x] = clock64(); s_pr[threadIdx.x] = 15; }
    Prof prof{s_pt, s_pr, s_pacc};
#endif
    const BssSoA &bs = job.bs;
    const uint32_t count = *job.count;
    const uint32_t rounded = (count + 255u) & ~255u;
    unsigned long long zero_num = 0, n_valid = 0, n_bytes = 0;
    uint32_t n_assert = 0;   // PtCounters::reference_asserts
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < rounded; qi += gridDim.x * blockDim.x) {
    const bool valid = qi < count;
    bool push_ext = false, push_resolve = false, push_shadow = false, push_mis = false, push_self = false;
    int finished_bounces = -1;
    uint32_t pid = 0;
    const bool stage_b = VOL && job.stage_b != 0u;   // the queue holds exit-point vertices whose shadow / MIS rays (or their next segments) have just been traced
    if (valid) {
        n_valid++;
        pid = job.queue[qi];
        // k_trace<.., PROBE> walked the whole chain: the hit record is the selected intersection (PT_NONE: no intersection with
        // the BSSRDF's material, S = 0), ps.ox.. / ps.dx.. the segment that found it, bs.cnt the chain's nfound
        const V3 ro(ps.ox(pid), ps.oy(pid), ps.oz(pid)), rd(ps.dx(pid), ps.dy(pid), ps.dz(pid));
        const uint32_t hp = ps.hit_prim(pid);
        // the frame record (kernels.h: BssSoA): {po, nfound} {ns, iface} {ss, material} -- three quads of one 48-byte record, requested together
        const float4 *const fq = bs.frame + (size_t)pid * BssSoA::kFrameQuads;
        const float4 fr0 = fq[0], fr1 = fq[1], fr2 = fq[2];
        const uint32_t mat = __float_as_uint(fr2.w);
        const uint32_t nfound = __float_as_uint(fr0.w);
        const bool at_exit = hp != PT_NONE, dead = !at_exit;
        SurfaceInteraction si;
        if (at_exit) fill_hit_pkt<SPH>(s, ps.hit_pkt(pid), SPH ? ps.hit_inst(pid) : PT_NONE, ro, rd, ps.hit_b0(pid), ps.hit_b1(pid), ps.hit_b2(pid), si);
        n_bytes += 4 + 24 + 16 + 8;

        if (at_exit || dead) {
            uint32_t meta = ps.meta(pid);
            uint32_t flags = meta >> 24, bounces = (meta >> 16) & 0xffu;
            Sampler smp; smp.index = ps.sobol_index(pid); smp.dim = meta & 0xffffu; smp.m32 = tabs.m32; smp.nib = tabs.nib; smp.lds = s_sobol; smp.lds_dims = LDS_DIMS; smp.overflow = false; smp.halton = rc.halton.enabled != 0; smp.prime = tabs.prime; smp.prime_sum = tabs.prime_sum; smp.perm = tabs.perm;
            smp.base = 0xffffffffu;
            RGB L(ps.L_r(pid), ps.L_g(pid), ps.L_b(pid));
            RGB beta(ps.beta_r(pid), ps.beta_g(pid), ps.beta_b(pid));
            n_bytes += 4 + 8 + 12 + 12 + 12 + 12 + 4;
            if (stage_b) {   // this vertex's own estimate: ratio-tracking transmittances draw the dimensions that follow its light and scattering samples (kern_shade.h, stage B)
                smp.load_window();
                if (!(s.has_shells && vol_chain_step<SPH>(s, ps, pid, flags, smp, push_shadow, push_mis, n_bytes)))
                    resolve_pending<SPH, VOL>(s, ps, pid, flags, L, zero_num, n_assert, n_bytes PT_PROF_PASS, s.has_shells ? nullptr : &smp);
                flags &= ~PF_STAGE_B;
            }
            // the outgoing vertex's NEE rays were traced at the start of the iteration after its shade
            else resolve_pending<SPH, VOL>(s, ps, pid, flags, L, zero_num, n_assert, n_bytes PT_PROF_PASS);
            bool terminated = dead;
            if (at_exit) {
                const PtMaterial &m = s.materials[mat];
                DevBssrdf bss;   // the BSSRDF of the entry point: tabulated, with the sigma_a / sigma_s its textures gave there, or DisneyBSSRDF
                if (m.type == PT_MAT_DISNEY) bss.init_disney(m);
                else if (bss_coef_stored(s.n_textures, m)) {   // sigma_a / sigma_s came from textures (or a kdsubsurface conversion) at the entry point
                    const float4 *const cq = bs.coef + (size_t)pid * BssSoA::kCoefQuads; const float4 c0 = cq[0], c1 = cq[1];
                    bss.init_medium(m, s.bss_tables, RGB(c0.x, c0.y, c0.z), RGB(c1.x, c1.y, c1.z));
                    n_bytes += 32;
                } else bss.init_medium(m, s.bss_tables, rgb3(m.sigma_a), rgb3(m.sigma_s));   // the material's constants, as kern_shade.h took them (subsurface.rs:100-101 with constant textures)
                bss.ns = V3(fr1.x, fr1.y, fr1.z); bss.ss = V3(fr2.x, fr2.y, fr2.z);
                bss.ts = cross(bss.ns, bss.ss); bss.po_p = V3(fr0.x, fr0.y, fr0.z);
                n_bytes += 48 - 8;   // (the frame record; its two id words are in the 4 + 24 + 16 + 8 above)
                // bssrdf.rs:403-405: pdf = pdf_sp(pi) / nfound ; Sp = sr(|po - pi|)
                float pdf = 0.0f;
                bool go = true;
                if (!stage_b) {
                    pdf = bss.pdf_sp(si.p, si.n) / (float)nfound;
                    const RGB S = bss.sr(length(bss.po_p - si.p));
                    if (S.is_black() || pdf == 0.0f) { terminated = true; go = false; }   // path.rs:185
                    else { smp.load_window(); beta = beta * (S / pdf); }
                }
                if (go) {
                    // sample_s (bssrdf.rs:563-571): BSDF::new(pi, 1.0) + adapter lobe; pi.wo = shading.n
                    BssrdfAdapterBsdf bsdf; bsdf.init(si, bss.eta);
                    si.wo = si.sh_n;
                    IData it; it.p = si.p; it.p_error = si.p_error; it.n = si.n;
                    // path.rs:188-192: direct lighting at pi (not part of the zero-radiance statistic)
                    MedIface mif{PT_NONE, PT_NONE};
                    if (VOL) { const uint32_t pk = __float_as_uint(fr1.w); mif.inside = (pk & 0xffffu) == 0xffffu ? PT_NONE : (pk & 0xffffu); mif.outside = (pk >> 16) == 0xffffu ? PT_NONE : (pk >> 16); }
                    if (!stage_b) {
                        if (nee_vertex<SPH, BssrdfAdapterBsdf, VOL, false>(s, grid, ps, pid, smp, si, it, bsdf, beta, flags, push_shadow, push_mis, n_bytes PT_PROF_PASS, mif)) flags |= PF_NEE_UNCOUNTED;
                        else L = L + beta * RGB(0.0f);   // path.rs:190-192 `L += beta * uniform_sample_one_light(..)` with a black estimate: 0, or NaN when pdf_sp was (an exit point a few ulps from po: inf x 0)
                    }
                    // wait for the traced rays before drawing any further dimension (the vertex's shadow / MIS rays; with shells: the next segment of one of them)
                    const bool defer = VOL && (s.has_grid != 0u || s.has_shells != 0u) && (push_shadow || push_mis);
                    // path.rs:194-201: indirect component
                    V3 wi; int sflags = 0;
                    RGB ff(0.0f);
                    if (defer) { flags |= PF_STAGE_B; push_self = true; }
                    else ff = bsdf.sample_f(si.wo, wi, smp.get_2d(), pdf, BSDF_ALL, sflags);
                    if (defer) { /* stage B samples on */ }
                    else if (ff.is_black() || pdf == 0.0f) terminated = true;
                    else {
                        beta = beta * (ff * abs_dot(wi, si.sh_n) / pdf);
                        if (__builtin_isinf(beta.y())) n_assert++;   // path.rs:201 / volpath.rs:210
                        if (sflags & BSDF_SPECULAR) flags |= PF_SPECULAR; else flags &= ~PF_SPECULAR;
                        V3 o; spawn_ray(it, wi, o);
                        // path.rs:206-214 Russian roulette
                        const RGB rrbeta = beta * ps.etascale(pid);
                        bool rr_kill = false;
                        if (rrbeta.max_component_value() < rc.rr_threshold && bounces > 3) {
                            const float q = maxf(1.0f - rrbeta.max_component_value(), 0.05f);
                            if (smp.get_1d() < q) rr_kill = true;
                            else { beta = beta / (1.0f - q); if (__builtin_isinf(beta.y())) n_assert++; }   // path.rs:213
                        }
                        if (rr_kill) terminated = true;
                        else {
                            bounces += 1;
                            ps.ox(pid) = o.x; ps.oy(pid) = o.y; ps.oz(pid) = o.z;
                            ps.dx(pid) = wi.x; ps.dy(pid) = wi.y; ps.dz(pid) = wi.z;
                            if (VOL) ps.medium(pid) = medium_toward(mif, si.n, wi);   // pi.spawn_ray(wi) (interaction.rs:32-36,54-66)
                            push_ext = true; n_bytes += 24 + 4 + 4;
                        }
                    }
                }
            }
            if (terminated) {
                if (flags & (PF_PEND_SHADOW | PF_PEND_MIS)) { flags |= PF_DEAD; push_resolve = true; }
                else { finished_bounces = (int)bounces; flags |= PF_FINISHED; }
            }
            if (smp.overflow) atomicMax(job.error, (uint32_t)PT_ERR_SOBOL_DIMENSIONS);
            ps.L_r(pid) = L.r; ps.L_g(pid) = L.g; ps.L_b(pid) = L.b;
            ps.beta_r(pid) = beta.r; ps.beta_g(pid) = beta.g; ps.beta_b(pid) = beta.b;
            ps.meta(pid) = (smp.dim & 0xffffu) | ((bounces & 0xffu) << 16) | (flags << 24);
        }
    }
    lq_push(s_qext, pid, push_ext);
    lq_push(s_qres, pid, push_resolve && job.shade_next0 != nullptr);   // (no miss pass: k_film_final ends the dead paths)
    lq_push(s_qsh, pid, push_shadow);
    lq_push(s_qmis, pid, push_mis);
    if (VOL) lq_push(s_qself, pid, push_self);
    if (finished_bounces >= 0) atomicAdd(&s_hist[finished_bounces > 15 ? 15 : finished_bounces], 1u);
    __syncthreads();
    lq_flush_nosync(s_qext, job.ext_next_count, job.ext_next, 256u, false);
    lq_flush_nosync(s_qres, job.shade_next0_count, job.shade_next0, 256u, false);
    lq_flush_nosync(s_qsh, job.shadow_count, job.shadow, 256u, false);
    lq_flush_nosync(s_qmis, job.mis_count, job.mis, 256u, false);
    if (VOL) lq_flush_nosync(s_qself, job.self_next_count, job.self_next, 256u, false);
    __syncthreads();
    }
    lq_flush_nosync(s_qext, job.ext_next_count, job.ext_next, 0u, true);
    lq_flush_nosync(s_qres, job.shade_next0_count, job.shade_next0, 0u, true);
    lq_flush_nosync(s_qsh, job.shadow_count, job.shadow, 0u, true);
    lq_flush_nosync(s_qmis, job.mis_count, job.mis, 0u, true);
    if (VOL) lq_flush_nosync(s_qself, job.self_next_count, job.self_next, 0u, true);
    __syncthreads();
    if (threadIdx.x < 16 && s_hist[threadIdx.x]) atomicAdd(&job.counters->path_len[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
    counter_add(&job.counters->zero_num, zero_num);
    counter_add(&job.counters->ref_asserts, (unsigned long long)n_assert);
    counter_add(&job.counters->stages, n_valid);
    counter_add(&job.counters->bss_items, n_valid);
    counter_add(&job.counters->bss_bytes, n_bytes);
}
