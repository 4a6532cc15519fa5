// kern_aux.h -- split out of the former single-file kernels.hip so that the translation units compile in parallel.
#pragma once
#include "kern_shade_common.h"
#include "kern_film.h"
// ---- volumetric path integrator: medium sampling between traversal and shading ---------------------------------------
// volpath.rs:98-104: after Scene::intersect, a ray that travels in a medium samples it (two sampler dimensions) and scales beta;
// a sampled medium vertex goes to the medium class, a black beta ends the path, everything else is routed as k_route does.
__global__ __launch_bounds__(256) void k_medium_route(DeviceScene s, RenderConst rc, SobolTables tabs, PathSoA ps, const uint32_t *queue, const uint32_t *count_ptr,
                                                     uint32_t *class_count, uint32_t *c0, uint32_t *c1, uint32_t *c2, uint32_t *c3, uint32_t *c4, uint32_t *c5, uint32_t *error) {
    __shared__ LdsQueue<1024> q0, q1, q2, q3, q4, q5;
    // (the two medium-sampling dimensions, or a grid medium's delta-tracking run, come one at a time from the HBM nibble tables: no LDS copy)
    lq_init(q0); lq_init(q1); lq_init(q2); lq_init(q3); lq_init(q4); lq_init(q5);
    __syncthreads();
    const uint32_t count = *count_ptr;
    const uint32_t rounded = (count + 255u) & ~255u;
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < rounded; qi += gridDim.x * blockDim.x) {
        const bool valid = qi < count;
        uint32_t pid = 0, cls = (uint32_t)kMissClass;
        if (valid) {
            pid = queue[qi];
            const uint32_t hp = ps.hit_prim(pid);
            cls = class_general((ps.hit_pflags(pid) >> kTpClassShift) & kTpClassMask);   // the hit packet's class bits, kMissClass for a miss (written by k_trace); the lobe-set classes fold into their lobe-count class (general kernels here)
            if (cls == (uint32_t)kSpecClass) cls = 1u;            // volpath.rs:136-138 estimates direct light at every vertex: no specular-only class here
            const uint32_t med = ps.medium(pid);
            if (med != PT_NONE) {
                uint32_t meta = ps.meta(pid);
                Sampler smp; smp.index = ps.sobol_index(pid); smp.dim = meta & 0xffffu; smp.m32 = tabs.m32; smp.nib = tabs.nib; smp.lds = nullptr; smp.lds_dims = 0u; smp.overflow = false;
                smp.halton = rc.halton.enabled != 0; smp.prime = tabs.prime; smp.prime_sum = tabs.prime_sum; smp.perm = tabs.perm; smp.base = 0xffffffffu;
                const V3 rd(ps.dx(pid), ps.dy(pid), ps.dz(pid));
                bool sampled; float t;
                RGB w(1.0f);
                if (s.media[med].type == PT_MEDIUM_GRID) {   // delta tracking: a data-dependent number of dimensions (grid.rs:149-182)
                    const V3 ro(ps.ox(pid), ps.oy(pid), ps.oz(pid));
                    w = grid_sample(s.media[med], s.grid_aux[med], ro, rd, hp != PT_NONE ? ps.hit_t(pid) : PT_INF, smp, sampled, t);
                } else {
                    const float u_channel = smp.get_1d(), u_dist = smp.get_1d();
                    w = medium_sample(s.media[med], hp != PT_NONE ? ps.hit_t(pid) : PT_INF, rd, u_channel, u_dist, sampled, t);
                }
                const RGB beta = RGB(ps.beta_r(pid), ps.beta_g(pid), ps.beta_b(pid)) * w;
                ps.beta_r(pid) = beta.r; ps.beta_g(pid) = beta.g; ps.beta_b(pid) = beta.b;
                uint32_t flags = meta >> 24;
                if (beta.is_black()) { flags |= PF_DEAD; cls = (uint32_t)kMissClass; }          // volpath.rs:105 `break`
                else if (sampled) { ps.hit_t(pid) = t; cls = (uint32_t)kMediumClass; }
                if (smp.overflow) atomicMax(error, (uint32_t)PT_ERR_SOBOL_DIMENSIONS);
                ps.meta(pid) = (smp.dim & 0xffffu) | (meta & 0x00ff0000u) | (flags << 24);
            } else if (RGB(ps.beta_r(pid), ps.beta_g(pid), ps.beta_b(pid)).is_black()) {
                // volpath.rs:111 tests beta after the OPTIONAL medium sample: a throughput that two surfaces with disjoint colours
                // multiplied to zero ends the path here too, after this ray's Scene::intersect (found by the fuzz sweep, seed 2005)
                const uint32_t meta = ps.meta(pid);
                ps.meta(pid) = (meta & 0x00ffffffu) | (((meta >> 24) | PF_DEAD) << 24);
                cls = (uint32_t)kMissClass;
            }
        }
        lq_push(q0, pid, valid && cls == 0u); lq_push(q1, pid, valid && cls == 1u); lq_push(q2, pid, valid && cls == 2u);
        lq_push(q3, pid, valid && cls == 3u); lq_push(q4, pid, valid && cls == 4u); lq_push(q5, pid, valid && cls == 5u);
        __syncthreads();
        lq_flush_nosync(q0, class_count + 0, c0, 256u, false); lq_flush_nosync(q1, class_count + 1, c1, 256u, false);
        lq_flush_nosync(q2, class_count + 2, c2, 256u, false); lq_flush_nosync(q3, class_count + 3, c3, 256u, false);
        lq_flush_nosync(q4, class_count + 4, c4, 256u, false); lq_flush_nosync(q5, class_count + 5, c5, 256u, false);
        __syncthreads();
    }
    lq_flush_nosync(q0, class_count + 0, c0, 0u, true); lq_flush_nosync(q1, class_count + 1, c1, 0u, true);
    lq_flush_nosync(q2, class_count + 2, c2, 0u, true); lq_flush_nosync(q3, class_count + 3, c3, 0u, true);
    lq_flush_nosync(q4, class_count + 4, c4, 0u, true); lq_flush_nosync(q5, class_count + 5, c5, 0u, true);
}

// ---- medium vertices (class kMediumClass): volpath.rs:107-123 ---------------------------------------------------------------
// uniform_sample_onelight with the phase function in the BSDF's place (integrator.rs:142-147,186-190), then a new direction from
// the phase function; beta is unchanged (phase value / its pdf = 1) and the ray stays in the same medium.
__global__ __launch_bounds__(256) void k_shade_medium(DeviceScene s, RenderConst rc, SobolTables tabs, LightGrid grid, PathSoA ps, ShadeJob job) {
    constexpr uint32_t LDS_DIMS = 56u;
    __shared__ uint32_t s_sobol[LDS_DIMS * kSobolNibWords];
    __shared__ LdsQueue<1024> s_qext, s_qres, s_qsh, s_qmis, s_qself;
    __shared__ uint32_t s_hist[16];
    lq_init(s_qext); lq_init(s_qres); lq_init(s_qsh); lq_init(s_qmis); lq_init(s_qself);
    if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
#ifdef PT_REGION_PROFILE
    __shared__ long long s_pt[4]; __shared__ int s_pr[4]; __shared__ unsigned long long s_pacc[64];
    if (threadIdx.x < 4) { s_pt[threadIdx.x] = clock64(); s_pr[threadIdx.x] = 15; }
    Prof prof{s_pt, s_pr, s_pacc};
#endif
    sobol_stage_lds(s_sobol, tabs.nib, LDS_DIMS, threadIdx.x, blockDim.x);
    __syncthreads();
    const uint32_t count = *job.count;
    const uint32_t rounded = (count + 255u) & ~255u;
    unsigned long long zero_num = 0, n_valid = 0, n_bytes = 0;
    uint32_t n_assert = 0;   // PtCounters::reference_asserts
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < rounded; qi += gridDim.x * blockDim.x) {
        const bool valid = qi < count;
        bool push_ext = false, push_resolve = false, push_shadow = false, push_mis = false, push_self = false;
        int finished_bounces = -1;
        uint32_t pid = 0;
        if (valid) {
            n_valid++;
            pid = job.queue[qi];
            const uint32_t meta = ps.meta(pid);
            uint32_t flags = meta >> 24, bounces = (meta >> 16) & 0xffu;
            Sampler smp; smp.index = ps.sobol_index(pid); smp.dim = meta & 0xffffu; smp.m32 = tabs.m32; smp.nib = tabs.nib; smp.lds = s_sobol; smp.lds_dims = LDS_DIMS; smp.overflow = false;
            smp.halton = rc.halton.enabled != 0; smp.prime = tabs.prime; smp.prime_sum = tabs.prime_sum; smp.perm = tabs.perm; smp.base = 0xffffffffu;
            RGB L(ps.L_r(pid), ps.L_g(pid), ps.L_b(pid));
            RGB beta(ps.beta_r(pid), ps.beta_g(pid), ps.beta_b(pid));
            const bool stage_b = (flags & PF_STAGE_B) != 0u;   // grid media: this vertex's own NEE rays are back (see k_shade)
            if (stage_b) smp.load_window();
            if (stage_b && s.has_shells) { if (!vol_chain_step<true>(s, ps, pid, flags, smp, push_shadow, push_mis, n_bytes)) resolve_pending<true, true>(s, ps, pid, flags, L, zero_num, n_assert, n_bytes PT_PROF_PASS, nullptr); }
            else resolve_pending<true, true>(s, ps, pid, flags, L, zero_num, n_assert, n_bytes PT_PROF_PASS, stage_b ? &smp : nullptr);
            flags &= ~(PF_CAMERA_RAY | PF_STAGE_B);
            bool terminated = bounces >= rc.max_depth;   // volpath.rs:108
            if (!terminated) {
                if (!stage_b) smp.load_window();
                const V3 ro(ps.ox(pid), ps.oy(pid), ps.oz(pid)), rd(ps.dx(pid), ps.dy(pid), ps.dz(pid));
                const uint32_t med = ps.medium(pid);
                SurfaceInteraction si;   // only p and wo are read through the MediumInteraction
                si.p = ro + rd * ps.hit_t(pid); si.wo = -rd; si.n = V3(0.0f, 0.0f, 0.0f); si.sh_n = V3(0.0f, 0.0f, 0.0f); si.p_error = V3(0.0f, 0.0f, 0.0f);
                IData it; it.p = si.p; it.p_error = V3(0.0f, 0.0f, 0.0f); it.n = V3(0.0f, 0.0f, 0.0f);
                const PhaseBsdf phase{s.media[med].g, si.wo};
                bool defer = false;
                if (!stage_b) {
                    if (!nee_vertex<true, PhaseBsdf, true, true>(s, grid, ps, pid, smp, si, it, phase, beta, flags, push_shadow, push_mis, n_bytes PT_PROF_PASS, MedIface{med, med})) L = L + beta * RGB(0.0f);   // volpath.rs:120: `L += beta * Ld` with a black Ld
                }
                defer = (s.has_grid != 0u || s.has_shells != 0u) && (push_shadow || push_mis);
                if (defer) { flags |= PF_STAGE_B; push_self = true; }
                else {
                V3 wi;
                hg_sample_p(phase.g, si.wo, wi, smp.get_2d());
                flags &= ~PF_SPECULAR;   // specular_bounce = false
                // Russian roulette (volpath.rs:171-176)
                const RGB rrbeta = beta * ps.etascale(pid);
                bool rr_kill = false;
                if (rrbeta.max_component_value() < rc.rr_threshold && bounces > 3) {
                    const float q = maxf(1.0f - rrbeta.max_component_value(), 0.05f);
                    if (smp.get_1d() < q) rr_kill = true;
                    else { beta = beta / (1.0f - q); if (__builtin_isinf(beta.y())) n_assert++; }   // volpath.rs:223
                }
                if (rr_kill) terminated = true;
                else {
                    bounces += 1;
                    ps.ox(pid) = si.p.x; ps.oy(pid) = si.p.y; ps.oz(pid) = si.p.z;   // mi.spawn_ray(wi): no offset (n = 0, p_error = 0)
                    ps.dx(pid) = wi.x; ps.dy(pid) = wi.y; ps.dz(pid) = wi.z;
                    push_ext = true;
                }
                }
            }
            if (terminated) {
                if (flags & (PF_PEND_SHADOW | PF_PEND_MIS)) { flags |= PF_DEAD; push_resolve = true; }
                else finished_bounces = (int)bounces;
            }
            if (smp.overflow) atomicMax(job.error, (uint32_t)PT_ERR_SOBOL_DIMENSIONS);
            ps.L_r(pid) = L.r; ps.L_g(pid) = L.g; ps.L_b(pid) = L.b;
            ps.beta_r(pid) = beta.r; ps.beta_g(pid) = beta.g; ps.beta_b(pid) = beta.b;
            ps.meta(pid) = (smp.dim & 0xffffu) | ((bounces & 0xffu) << 16) | (flags << 24);
        }
        lq_push(s_qext, pid, push_ext); lq_push(s_qres, pid, push_resolve); lq_push(s_qsh, pid, push_shadow); lq_push(s_qmis, pid, push_mis); lq_push(s_qself, pid, push_self);
        if (finished_bounces >= 0) atomicAdd(&s_hist[finished_bounces > 15 ? 15 : finished_bounces], 1u);
        __syncthreads();
        lq_flush_nosync(s_qext, job.ext_next_count, job.ext_next, 256u, false);
        lq_flush_nosync(s_qres, job.shade_next0_count, job.shade_next0, 256u, false);
        lq_flush_nosync(s_qsh, job.shadow_count, job.shadow, 256u, false);
        lq_flush_nosync(s_qmis, job.mis_count, job.mis, 256u, false);
        lq_flush_nosync(s_qself, job.self_next_count, job.self_next, 256u, false);
        __syncthreads();
    }
    lq_flush_nosync(s_qext, job.ext_next_count, job.ext_next, 0u, true);
    lq_flush_nosync(s_qres, job.shade_next0_count, job.shade_next0, 0u, true);
    lq_flush_nosync(s_qsh, job.shadow_count, job.shadow, 0u, true);
    lq_flush_nosync(s_qmis, job.mis_count, job.mis, 0u, true);
    lq_flush_nosync(s_qself, job.self_next_count, job.self_next, 0u, true);
    __syncthreads();
    __syncthreads();
    if (threadIdx.x < 16 && s_hist[threadIdx.x]) atomicAdd(&job.counters->path_len[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
    counter_add(&job.counters->stages, n_valid);
    counter_add(&job.counters->shade_items[kMediumClass], n_valid);
    counter_add(&job.counters->shade_bytes[kMediumClass], n_bytes);
    counter_add(&job.counters->ref_asserts, (unsigned long long)n_assert);
    (void)zero_num;
}

// ---- escaped rays and dead paths (class kMissClass) -------------------------------------------------------------------
// More than half of the vertices of an open scene are rays that left it (S2: 54 %). They only need the previous vertex's
// NEE resolved, the environment's Le where path.rs:106-117 adds it, and the path-length histogram entry -- none of the
// BSDF / light-sampling code. Keeping them out of k_shade leaves its waves full of real surface hits.
template <bool SPH, bool VOL>
#ifndef PT_MISS_WAVES
#define PT_MISS_WAVES 1   // experiment hook
#endif
__global__ __launch_bounds__(256, PT_MISS_WAVES) void k_shade_miss(DeviceScene s, RenderConst rc, PathSoA ps, ShadeJob job) {
    __shared__ uint32_t s_hist[16];
    if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
#ifdef PT_REGION_PROFILE
    __shared__ long long s_pt[4]; __shared__ int s_pr[4]; __shared__ unsigned long long s_pacc[64];
    if (threadIdx.x < 4) { s_pt[threadIdx.x] = clock64(); s_pr[threadIdx.x] = 15; }
    Prof prof{s_pt, s_pr, s_pacc};   // not reported: the region table is k_shade's
#endif
    __syncthreads();
    const uint32_t count = *job.count;
    unsigned long long zero_num = 0, n_valid = 0, n_bytes = 0;
    uint32_t n_assert = 0;   // PtCounters::reference_asserts
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < count; qi += gridDim.x * blockDim.x) {
        n_valid++;
        n_bytes += 4 + 32 + /* write back */ 16 + 4;
        const uint32_t pid = job.queue[qi];
        float4 *cq = reinterpret_cast<float4 *>(ps.core) + 4 * (size_t)pid;   // {L, etascale} {beta, meta}
        const float4 c0 = cq[0], c1 = cq[1];
        const uint32_t meta = __float_as_uint(c1.w);
        uint32_t flags = meta >> 24; const uint32_t bounces = (meta >> 16) & 0xffu;
        RGB L(c0.x, c0.y, c0.z);
        resolve_pending<SPH, VOL>(s, ps, pid, flags, L, zero_num, n_assert, n_bytes PT_PROF_PASS);
        if (!(flags & PF_DEAD) && (bounces == 0 || (flags & PF_SPECULAR)) && s.n_infinite > 0) {   // path.rs:106-117, ray escaped
            n_bytes += 32;
            const RGB beta(c1.x, c1.y, c1.z);
            const float4 *rq = reinterpret_cast<const float4 *>(ps.ray) + 2 * (size_t)pid;
            const float4 r0 = rq[0], r1 = rq[1];
            const V3 rd(r0.w, r1.x, r1.y);
            for (uint32_t k = 0; k < s.n_infinite; ++k) L = L + light_le(s, s.lights[s.infinite_lights[k]], rd) * beta;
        }
        cq[0] = make_float4(L.r, L.g, L.b, c0.w);
        ps.meta(pid) = (meta & 0x00ffffffu) | ((flags & ~PF_CAMERA_RAY) << 24);
        atomicAdd(&s_hist[bounces > 15u ? 15u : bounces], 1u);   // path.rs:219
    }
    __syncthreads();
    if (threadIdx.x < 16 && s_hist[threadIdx.x]) atomicAdd(&job.counters->path_len[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
    counter_add(&job.counters->zero_num, zero_num);
    counter_add(&job.counters->ref_asserts, (unsigned long long)n_assert);
    counter_add(&job.counters->stages, n_valid);
    counter_add(&job.counters->shade_items[kMissClass], n_valid);
    counter_add(&job.counters->shade_bytes[kMissClass], n_bytes);
    (void)rc;
}

// ---- the film kernel that also ends the paths (plain path integrator, round 5) ------------------------------------------------
// Every path ends exactly once, by leaving the scene or by dying with its last vertex's next-event estimate still pending, and all k_shade_miss does for it is read its
// records (core, the pending NEE / MIS records, the ray for the environment's Le), add to L and write L back -- 54 % of S2's vertices, fetched by path id out of queues:
// scattered 64-byte gathers at 1.8x their useful bytes. The film kernel reads every path's core record anyway, in path-id order: it now does that last step itself,
// on fully coalesced streams, and the per-iteration miss pass (a queue append in the shade kernels and the router, a launch, a gather and a write-back) is gone.
// The arithmetic on L is k_shade_miss's, in its order (resolve_pending, then Le): the same bits reach the film.
#ifndef PT_FILM_WAVES
#define PT_FILM_WAVES 4   // waves per SIMD the kernel is compiled for: 128 registers + 32 B of scratch; C2 film 21.9 (three waves, no scratch) -> 19.2 ms, five / six / eight: 24.3 / 28.9 / 33.7
#endif
template <bool SPH>
__global__ __launch_bounds__(256, PT_FILM_WAVES) void k_film_final(DeviceScene s, RenderConst rc, PathSoA ps, const float *filter_table, float *film_rgbw, DevCounters *counters) {
    __shared__ uint32_t s_hist[16];
    if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
    __syncthreads();
#ifdef PT_REGION_PROFILE
    __shared__ long long s_pt[4]; __shared__ int s_pr[4]; __shared__ unsigned long long s_pacc[64];
    if (threadIdx.x < 4) { s_pt[threadIdx.x] = clock64(); s_pr[threadIdx.x] = 15; }
    Prof prof{s_pt, s_pr, s_pacc};   // not reported: the region table is k_shade's
#endif
    unsigned long long zero_num = 0, n_final = 0, n_bytes = 0;
    uint32_t n_assert = 0;
    film_slot(rc, ps, filter_table, film_rgbw, counters, [&](uint32_t pid, RGB &L) {
        const float4 c1 = reinterpret_cast<const float4 *>(ps.core)[4 * (size_t)pid + 1];
        const uint32_t meta = __float_as_uint(c1.w);
        uint32_t flags = meta >> 24; const uint32_t bounces = (meta >> 16) & 0xffu;
        const bool pend = (flags & (PF_PEND_SHADOW | PF_PEND_MIS)) != 0u;
        // When the pass's queues are empty every path has ended in one of three ways: its shade kernel finished it (PF_FINISHED: nothing was pending, or a subsurface probe
        // chain found no exit point), it died with an estimate pending (PF_DEAD), or -- neither flag -- its last ray left the scene. No look at the hit record needed.
        const bool escaped = !(flags & (PF_DEAD | PF_FINISHED));
        if (!pend && !escaped) return;
        n_final++; n_bytes += 32 + 4;
        resolve_pending<SPH, false, true>(s, ps, pid, flags, L, zero_num, n_assert, n_bytes PT_PROF_PASS);
        if (escaped && (bounces == 0 || (flags & PF_SPECULAR)) && s.n_infinite > 0) {   // path.rs:106-117, ray escaped
            n_bytes += 32;
            const RGB beta(c1.x, c1.y, c1.z);
            const float4 *rq = reinterpret_cast<const float4 *>(ps.ray) + 2 * (size_t)pid;
            const float4 r0 = rq[0], r1 = rq[1];
            const V3 rd(r0.w, r1.x, r1.y);
            for (uint32_t k = 0; k < s.n_infinite; ++k) L = L + light_le(s, s.lights[s.infinite_lights[k]], rd) * beta;
        }
        atomicAdd(&s_hist[bounces > 15u ? 15u : bounces], 1u);   // path.rs:219
    });
    __syncthreads();
    if (threadIdx.x < 16 && s_hist[threadIdx.x]) atomicAdd(&counters->path_len[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
    counter_add(&counters->zero_num, zero_num);
    counter_add(&counters->ref_asserts, (unsigned long long)n_assert);
    counter_add(&counters->stages, n_final);
    counter_add(&counters->shade_items[kMissClass], n_final);
    counter_add(&counters->shade_bytes[kMissClass], n_bytes);
}
