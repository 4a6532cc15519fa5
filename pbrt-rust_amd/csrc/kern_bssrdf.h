// kern_bssrdf.h -- split out of the former single-file kernels.hip so that the translation units compile in parallel.
#pragma once
#include "kern_shade_common.h"
// ---- subsurface scattering: probe chains + exit-point vertex (path.rs:177-204, bssrdf.rs:334-410,559-574) --------------
// One launch per wavefront iteration while any path walks a probe chain. Each queue entry is a path whose probe ray
// (ps.ox.. / ps.dx.., t_max = 1 - eps) was just traced into ps.hit_*. The chain is walked twice: a counting walk
// (nfound) and, once the miss ends it, a re-walk from the segment start up to match number `selected` -- the reference
// keeps the chain in a Vec and indexes it; re-walking is deterministic and keeps the per-path state fixed-size.
// When the exit point pi is reached the lane finishes the vertex: resolve po's next-event estimation, beta *= S / pdf,
// NEE at pi through the adapter BSDF, sample the adapter BSDF, Russian roulette, bounces += 1.
template <bool SPH>
__global__ __launch_bounds__(256) void k_bssrdf(DeviceScene s, RenderConst rc, SobolTables tabs, LightGrid grid, PathSoA ps, BssrdfJob job) {
    __shared__ uint32_t s_sobol[kSobolLdsWords];
    __shared__ LdsQueue<1024> s_qext, s_qres, s_qsh, s_qmis, s_qprobe;
    __shared__ uint32_t s_hist[16];
    lq_init(s_qext); lq_init(s_qres); lq_init(s_qsh); lq_init(s_qmis); lq_init(s_qprobe);
    if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
    sobol_stage_lds(s_sobol, tabs.m32, threadIdx.x, blockDim.x);
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
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < rounded; qi += gridDim.x * blockDim.x) {
    const bool valid = qi < count;
    bool push_ext = false, push_resolve = false, push_shadow = false, push_mis = false, push_probe = false;
    int finished_bounces = -1;
    uint32_t pid = 0;
    if (valid) {
        n_valid++;
        pid = job.queue[qi];
        const V3 ro(ps.ox[pid], ps.oy[pid], ps.oz[pid]), rd(ps.dx[pid], ps.dy[pid], ps.dz[pid]);
        const V3 target(bs.target_x[pid], bs.target_y[pid], bs.target_z[pid]);
        const uint32_t hp = ps.hit_prim[pid];
        const uint32_t mat = bs.mat[pid];
        uint32_t cnt = bs.cnt[pid];
        uint32_t nfound = cnt & 0xffffu, seen = (cnt >> 16) & 0x7fffu; const bool rewalk = (cnt >> 31) != 0u;
        const float u1n = bs.u1n[pid];
        bool chain_end = false, at_exit = false, dead = false;
        SurfaceInteraction si;
        if (hp != PT_NONE) {
            fill_hit<SPH>(s, hp, SPH ? ps.hit_inst[pid] : PT_NONE, ro, rd, ps.hit_b0[pid], ps.hit_b1[pid], ps.hit_b2[pid], si);
            const bool match = s.prim_material[hp] == mat;   // bssrdf.rs:385-391
            if (!rewalk) { if (match) { if (nfound < 0x7fffu) nfound++; else atomicMax(job.error, (uint32_t)PT_ERR_PROBE_CHAIN); } }   // `seen` has 15 bits
            else if (match) {
                // bssrdf.rs:398: selected = clamp((u1n * nfound) as usize, 0, nfound - 1)
                const uint32_t selected = min(f2u32_sat(u1n * (float)nfound), nfound - 1u);
                if (seen == selected) at_exit = true;
                seen++;
            }
            if (!at_exit) {  // base = si.get_data(); next segment base -> target (interaction.rs:38-43)
                const V3 d = target - si.p;
                if (d.x == 0.0f && d.y == 0.0f && d.z == 0.0f) chain_end = true;
                else {
                    const V3 o = offset_ray_origin(si.p, si.p_error, si.n, d);
                    ps.ox[pid] = o.x; ps.oy[pid] = o.y; ps.oz[pid] = o.z;
                    ps.dx[pid] = d.x; ps.dy[pid] = d.y; ps.dz[pid] = d.z;
                    push_probe = true;
                }
            }
        } else chain_end = true;
        if (chain_end) {
            if (!rewalk && nfound > 0u) {  // chain counted: walk it again up to the selected intersection
                const V3 start(bs.start_x[pid], bs.start_y[pid], bs.start_z[pid]);
                const V3 d = target - start;
                ps.ox[pid] = start.x; ps.oy[pid] = start.y; ps.oz[pid] = start.z;
                ps.dx[pid] = d.x; ps.dy[pid] = d.y; ps.dz[pid] = d.z;
                cnt = nfound | (1u << 31); seen = 0u;
                bs.cnt[pid] = cnt;
                push_probe = true;
            } else dead = true;   // nfound == 0: S = 0 (bssrdf.rs:397); a re-walk never ends before `selected`
        } else if (push_probe) bs.cnt[pid] = nfound | (seen << 16) | (rewalk ? (1u << 31) : 0u);
        n_bytes += 4 + 24 + 16 + 12 + 8 + 4 + (push_probe ? 24 + 4 + 4 : 0);

        if (at_exit || dead) {
            uint32_t meta = ps.meta[pid];
            uint32_t flags = meta >> 24, bounces = (meta >> 16) & 0xffu;
            Sampler smp; smp.index = ps.sobol_index[pid]; smp.dim = meta & 0xffffu; smp.m32 = tabs.m32; smp.lds = s_sobol; smp.overflow = false; smp.halton = rc.halton.enabled != 0; smp.prime = tabs.prime; smp.prime_sum = tabs.prime_sum; smp.perm = tabs.perm;
            smp.base = 0xffffffffu;
            RGB L(ps.L_r[pid], ps.L_g[pid], ps.L_b[pid]);
            RGB beta(ps.beta_r[pid], ps.beta_g[pid], ps.beta_b[pid]);
            n_bytes += 4 + 8 + 12 + 12 + 12 + 12 + 4;
            // the outgoing vertex's NEE rays were traced at the start of the iteration after its shade
            resolve_pending<SPH>(s, ps, pid, flags, L, zero_num, n_bytes PT_PROF_PASS);
            bool terminated = dead;
            if (at_exit) {
                const PtMaterial &m = s.materials[mat];
                DevBssrdf bss; bss.init_material(m, s.bss_tables);   // tabulated (textured sigma_a / sigma_s are rejected at scene creation) or DisneyBSSRDF
                bss.ns = V3(bs.ns_x[pid], bs.ns_y[pid], bs.ns_z[pid]); bss.ss = V3(bs.ss_x[pid], bs.ss_y[pid], bs.ss_z[pid]);
                bss.ts = cross(bss.ns, bss.ss); bss.po_p = V3(bs.po_x[pid], bs.po_y[pid], bs.po_z[pid]);
                n_bytes += 36;
                // bssrdf.rs:403-405: pdf = pdf_sp(pi) / nfound ; Sp = sr(|po - pi|)
                float pdf = bss.pdf_sp(si.p, si.n) / (float)nfound;
                const RGB S = bss.sr(length(bss.po_p - si.p));
                if (S.is_black() || pdf == 0.0f) terminated = true;   // path.rs:185
                else {
                    smp.load_window();
                    beta = beta * (S / pdf);
                    // sample_s (bssrdf.rs:563-571): BSDF::new(pi, 1.0) + adapter lobe; pi.wo = shading.n
                    BssrdfAdapterBsdf bsdf; bsdf.init(si, bss.eta);
                    si.wo = si.sh_n;
                    IData it; it.p = si.p; it.p_error = si.p_error; it.n = si.n;
                    // path.rs:188-192: direct lighting at pi (not part of the zero-radiance statistic)
                    if (nee_vertex<SPH>(s, grid, ps, pid, smp, si, it, bsdf, beta, flags, push_shadow, push_mis, n_bytes PT_PROF_PASS)) flags |= PF_NEE_UNCOUNTED;
                    // path.rs:194-201: indirect component
                    V3 wi; int sflags = 0;
                    const RGB ff = bsdf.sample_f(si.wo, wi, smp.get_2d(), pdf, BSDF_ALL, sflags);
                    if (ff.is_black() || pdf == 0.0f) terminated = true;
                    else {
                        beta = beta * (ff * abs_dot(wi, si.sh_n) / pdf);
                        if (sflags & BSDF_SPECULAR) flags |= PF_SPECULAR; else flags &= ~PF_SPECULAR;
                        V3 o; spawn_ray(it, wi, o);
                        // path.rs:206-214 Russian roulette
                        const RGB rrbeta = beta * ps.etascale[pid];
                        bool rr_kill = false;
                        if (rrbeta.max_component_value() < rc.rr_threshold && bounces > 3) {
                            const float q = maxf(1.0f - rrbeta.max_component_value(), 0.05f);
                            if (smp.get_1d() < q) rr_kill = true;
                            else beta = beta / (1.0f - q);
                        }
                        if (rr_kill) terminated = true;
                        else {
                            bounces += 1;
                            ps.ox[pid] = o.x; ps.oy[pid] = o.y; ps.oz[pid] = o.z;
                            ps.dx[pid] = wi.x; ps.dy[pid] = wi.y; ps.dz[pid] = wi.z;
                            push_ext = true; n_bytes += 24 + 4 + 4;
                        }
                    }
                }
            }
            if (terminated) {
                if (flags & (PF_PEND_SHADOW | PF_PEND_MIS)) { flags |= PF_DEAD; push_resolve = true; }
                else finished_bounces = (int)bounces;
            }
            if (smp.overflow) atomicMax(job.error, (uint32_t)PT_ERR_SOBOL_DIMENSIONS);
            ps.L_r[pid] = L.r; ps.L_g[pid] = L.g; ps.L_b[pid] = L.b;
            ps.beta_r[pid] = beta.r; ps.beta_g[pid] = beta.g; ps.beta_b[pid] = beta.b;
            ps.meta[pid] = (smp.dim & 0xffffu) | ((bounces & 0xffu) << 16) | (flags << 24);
        }
    }
    lq_push(s_qprobe, pid, push_probe);
    lq_push(s_qext, pid, push_ext);
    lq_push(s_qres, pid, push_resolve);
    lq_push(s_qsh, pid, push_shadow);
    lq_push(s_qmis, pid, push_mis);
    if (finished_bounces >= 0) atomicAdd(&s_hist[finished_bounces > 15 ? 15 : finished_bounces], 1u);
    __syncthreads();
    lq_flush_nosync(s_qprobe, job.probe_next_count, job.probe_next, 256u, false);
    lq_flush_nosync(s_qext, job.ext_next_count, job.ext_next, 256u, false);
    lq_flush_nosync(s_qres, job.shade_next0_count, job.shade_next0, 256u, false);
    lq_flush_nosync(s_qsh, job.shadow_count, job.shadow, 256u, false);
    lq_flush_nosync(s_qmis, job.mis_count, job.mis, 256u, false);
    __syncthreads();
    }
    lq_flush_nosync(s_qprobe, job.probe_next_count, job.probe_next, 0u, true);
    lq_flush_nosync(s_qext, job.ext_next_count, job.ext_next, 0u, true);
    lq_flush_nosync(s_qres, job.shade_next0_count, job.shade_next0, 0u, true);
    lq_flush_nosync(s_qsh, job.shadow_count, job.shadow, 0u, true);
    lq_flush_nosync(s_qmis, job.mis_count, job.mis, 0u, true);
    __syncthreads();
    if (threadIdx.x < 16 && s_hist[threadIdx.x]) atomicAdd(&job.counters->path_len[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
    counter_add(&job.counters->zero_num, zero_num);
    counter_add(&job.counters->stages, n_valid);
    counter_add(&job.counters->bss_items, n_valid);
    counter_add(&job.counters->bss_bytes, n_bytes);
}
