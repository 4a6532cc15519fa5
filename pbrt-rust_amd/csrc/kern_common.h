// kern_common.h -- includes and wave / LDS-queue helpers shared by every kernel translation unit.
//
//
// Pipeline per pass (one pass = s_count samples of every pixel owned by this rank):
//   k_generate   integrator.rs:331-346  sampler.rs:170-180  perspective.rs:120-179
//   repeat until no path is alive:
//     k_trace<closest>  bvh.rs:705-760 + triangle.rs:136-233   (continuation rays, then MIS rays)
//     k_trace<any>      bvh.rs:762-814 + triangle.rs:400-495   (shadow rays)
//     k_shade<class>    path.rs:97-217 + integrator.rs:81-237  (one launch per material class queue)
//   k_film       integrator.rs:350-374 + film.rs:292-331
// Queues hold path ids; path state is SoA in HBM (kernels.h). Compaction is wave64 ballot + popcount.
#pragma once
#include "kernels.h"
#include "dev_bsdf.h"
#include "dev_sphere.h"
#include "dev_texture.h"
#include "dev_medium.h"

using namespace ptd;

// ---- wave-level helpers ------------------------------------------------------------------------
PT_DEV uint32_t lane_id() { return __lane_id(); }

// Stream compaction: append `value` for every lane with pred set; one atomic per wave.
PT_DEV void queue_push(uint32_t *count, uint32_t *buf, uint32_t value, bool pred) {
    unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    uint32_t lane = lane_id();
    uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    if (pred) buf[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = value;
}
PT_DEV unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
PT_DEV void counter_add(unsigned long long *dst, unsigned long long v) {  // call wave-convergent
    v = wave_sum(v);
    if (lane_id() == 0 && v) atomicAdd(dst, v);
}

// Staging of queue appends in LDS. A single global counter sustains only ~88 returning atomics per microsecond
// (MI355X_MICROARCH.md, row "dequeue"); one atomic per wave per append made every queue-producing kernel atomic bound.
//
// LdsQueue (block level, streaming kernels): appends go to a block-wide LDS buffer with LDS atomics (one per wave) and
// the whole block flushes ~1000 entries with ONE global atomic. All threads of the block call lq_sync_flush together.
template <int CAP> struct LdsQueue { uint32_t count; uint32_t base; uint32_t buf[CAP]; };
template <int CAP> PT_DEV void lq_init(LdsQueue<CAP> &q) { if (threadIdx.x == 0) { q.count = 0; q.base = 0; } }
template <int CAP> PT_DEV void lq_push(LdsQueue<CAP> &q, uint32_t value, bool pred) {
    unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    uint32_t lane = lane_id();
    uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&q.count, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    if (pred) q.buf[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = value;
}
// Block-wide append in bins: the entries one round of the block appends land grouped by `bin` (0..7: the direction octant of a new
// ray), so the 64 consecutive entries a traversal wave takes point the same way more often. All threads of the block call it together;
// `bins` is 16 words of shared memory. Three barriers; the order inside a bin is not defined (nothing depends on it).
template <int CAP> PT_DEV void lq_push_binned(LdsQueue<CAP> &q, uint32_t *bins, uint32_t value, bool pred, uint32_t bin) {
    if (threadIdx.x < 8) bins[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t rank = pred ? atomicAdd(&bins[bin], 1u) : 0u;
    __syncthreads();
    uint32_t start = 0, total = 0;
#pragma unroll
    for (uint32_t b = 0; b < 8; ++b) { const uint32_t c = bins[b]; if (b < bin) start += c; total += c; }
    const uint32_t base = q.count;
    if (pred) q.buf[base + start + rank] = value;
    __syncthreads();
    if (threadIdx.x == 0) q.count = base + total;
}
// Flush when fewer than `reserve` free slots remain (or force). Block-uniform; contains __syncthreads().
template <int CAP> PT_DEV void lq_sync_flush(LdsQueue<CAP> &q, uint32_t *gcount, uint32_t *gbuf, uint32_t reserve, bool force) {
    __syncthreads();
    const uint32_t n = q.count;
    if (n != 0 && (force || n + reserve > (uint32_t)CAP)) {
        if (threadIdx.x == 0) q.base = atomicAdd(gcount, n);
        __syncthreads();
        const uint32_t b = q.base;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) gbuf[b + i] = q.buf[i];
        __syncthreads();
        if (threadIdx.x == 0) q.count = 0;
    }
    __syncthreads();
}

// (Measured alternative, not kept: barrier-free per-wave buffers. With 256-entry buffers the 4x more frequent returning
// atomics made every producer slower; with 768-entry buffers k_shade still lost 10 % -- the barriers keep the four waves of
// a block in lockstep through a 150 KB kernel, which evidently helps instruction fetch.)

// Several queues per kernel: one barrier makes the pushes visible, each queue that is nearly full flushes (block-uniform
// decision, rare), one barrier closes the round -- instead of two barriers per queue per iteration.
template <int CAP> PT_DEV void lq_flush_nosync(LdsQueue<CAP> &q, uint32_t *gcount, uint32_t *gbuf, uint32_t reserve, bool force) {
    const uint32_t n = q.count;   // the caller's barrier precedes this read
    if (n != 0 && (force || n + reserve > (uint32_t)CAP)) {
        if (threadIdx.x == 0) q.base = atomicAdd(gcount, n);
        __syncthreads();
        const uint32_t b = q.base;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) gbuf[b + i] = q.buf[i];
        __syncthreads();
        if (threadIdx.x == 0) q.count = 0;
    }
}

// Rebuild the SurfaceInteraction of a recorded hit (triangle: from the barycentrics; sphere: re-evaluated from the ray).
template <bool SPH> PT_DEV void fill_hit(const DeviceScene &s, uint32_t prim, uint32_t inst, V3 ro, V3 rd, float b0, float b1, float b2, SurfaceInteraction &si) {
    const uint32_t sh = s.prim_shape[prim];
    if (SPH && inst != PT_NONE) {  // TransformedPrimitive::intersect (primitive.rs:58-80): object-space interaction, then to world
        const DevInstance &I = s.instances[inst];
        const M4 w2i = ldm4g(I.world_to_instance), i2w = ldm4g(I.instance_to_world);
        V3 oerr; V3 o2 = xf_point_err(w2i, ro, oerr); const V3 d2 = xf_vector(w2i, rd);
        const float l2 = length_squared(d2);
        if (l2 > 0.0f) { const float dt = dot(vabs(d2), oerr) / l2; o2 = o2 + d2 * dt; }
        if ((sh >> 30) == PT_SHAPE_SPHERE) sphere_fill_interaction(s.spheres[sh & 0x3fffffffu], o2, d2, si);
        else tri_fill_interaction(s, sh & 0x3fffffffu, d2, b0, b1, b2, true, si);
        if (!I.identity) {  // transform_surface_interaction (transform.rs:607-636)
            V3 perr;
            si.p = xf_point_abs_err(i2w, si.p, si.p_error, perr); si.p_error = perr;
            si.n = normalize(xf_normal_inv(w2i, si.n));
            si.wo = normalize(xf_vector(i2w, si.wo));
            si.dpdu = xf_vector(i2w, si.dpdu); si.dpdv = xf_vector(i2w, si.dpdv);
            si.sh_n = face_forward(normalize(xf_normal_inv(w2i, si.sh_n)), si.n);
            si.sh_dpdu = xf_vector(i2w, si.sh_dpdu); si.sh_dpdv = xf_vector(i2w, si.sh_dpdv);
            si.sh_dndu = xf_normal_inv(w2i, si.sh_dndu); si.sh_dndv = xf_normal_inv(w2i, si.sh_dndv);
        }
        return;
    }
    if (SPH && (sh >> 30) == PT_SHAPE_SPHERE) { sphere_fill_interaction(s.spheres[sh & 0x3fffffffu], ro, rd, si); return; }
    tri_fill_interaction(s, sh & 0x3fffffffu, rd, b0, b1, b2, true, si);
}

// The same interaction rebuilt from the hit's TriPacket (hit record word `hit_pkt`): vertices, shape reference and flag byte come
// from ONE 48-byte line (three dwordx4 loads) instead of the prim_shape -> indices -> P chain of dependent gathers. Returns the
// packet's flag word (material index and class in its upper bits, dev_scene.h).
template <bool SPH> PT_DEV uint32_t fill_hit_pkt(const DeviceScene &s, uint32_t pkt, uint32_t inst, V3 ro, V3 rd, float b0, float b1, float b2, SurfaceInteraction &si) {
    const uint4 *q = reinterpret_cast<const uint4 *>(s.leaf) + 3 * (size_t)pkt;
    const uint4 q0 = q[0], q1 = q[1], q2 = q[2];
    const uint32_t sh = q1.w, fl = q2.w;   // one quad per axis (dev_scene.h: TriPacket)
    const V3 p0(__uint_as_float(q0.x), __uint_as_float(q1.x), __uint_as_float(q2.x));
    const V3 p1(__uint_as_float(q0.y), __uint_as_float(q1.y), __uint_as_float(q2.y));
    const V3 p2(__uint_as_float(q0.z), __uint_as_float(q1.z), __uint_as_float(q2.z));
    if (SPH && inst != PT_NONE) {  // TransformedPrimitive::intersect (primitive.rs:58-80): object-space interaction, then to world
        const DevInstance &I = s.instances[inst];
        const M4 w2i = ldm4g(I.world_to_instance), i2w = ldm4g(I.instance_to_world);
        V3 oerr; V3 o2 = xf_point_err(w2i, ro, oerr); const V3 d2 = xf_vector(w2i, rd);
        const float l2 = length_squared(d2);
        if (l2 > 0.0f) { const float dt = dot(vabs(d2), oerr) / l2; o2 = o2 + d2 * dt; }
        if ((sh >> 30) == PT_SHAPE_SPHERE) sphere_fill_interaction(s.spheres[sh & 0x3fffffffu], o2, d2, si);
        else tri_fill_from(s, sh & 0x3fffffffu, fl & 0xffu, p0, p1, p2, d2, b0, b1, b2, true, si);
        if (!I.identity) {  // transform_surface_interaction (transform.rs:607-636)
            V3 perr;
            si.p = xf_point_abs_err(i2w, si.p, si.p_error, perr); si.p_error = perr;
            si.n = normalize(xf_normal_inv(w2i, si.n));
            si.wo = normalize(xf_vector(i2w, si.wo));
            si.dpdu = xf_vector(i2w, si.dpdu); si.dpdv = xf_vector(i2w, si.dpdv);
            si.sh_n = face_forward(normalize(xf_normal_inv(w2i, si.sh_n)), si.n);
            si.sh_dpdu = xf_vector(i2w, si.sh_dpdu); si.sh_dpdv = xf_vector(i2w, si.sh_dpdv);
            si.sh_dndu = xf_normal_inv(w2i, si.sh_dndu); si.sh_dndv = xf_normal_inv(w2i, si.sh_dndv);
        }
        return fl;
    }
    if (SPH && (sh >> 30) == PT_SHAPE_SPHERE) { sphere_fill_interaction(s.spheres[sh & 0x3fffffffu], ro, rd, si); return fl; }
    tri_fill_from(s, sh & 0x3fffffffu, fl & 0xffu, p0, p1, p2, rd, b0, b1, b2, true, si);
    return fl;
}
PT_DEV uint32_t packet_material(const DeviceScene &s, uint32_t pflags, uint32_t prim) {
    const uint32_t m = pflags >> kTpMatShift;
    return m != kTpMatNone ? m : s.prim_material[prim];
}
