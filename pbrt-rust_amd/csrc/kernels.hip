// kernels.hip -- the gfx950 wavefront path-tracing kernels (wave64).
//
// Pipeline per pass (one pass = s_count samples of every pixel owned by this rank):
//   k_generate   integrator.rs:331-346  sampler.rs:170-180  perspective.rs:120-179
//   repeat until no path is alive:
//     k_trace<closest>  bvh.rs:705-760 + triangle.rs:136-233   (continuation rays, then MIS rays)
//     k_trace<any>      bvh.rs:762-814 + triangle.rs:400-495   (shadow rays)
//     k_shade<class>    path.rs:97-217 + integrator.rs:81-237  (one launch per material class queue)
//   k_film       integrator.rs:350-374 + film.rs:292-331
// Queues hold path ids; path state is SoA in HBM (kernels.h). Compaction is wave64 ballot + popcount.
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

// ---- scene preparation ---------------------------------------------------------------------------
// Triangle packets in leaf order + the per-triangle "degenerate -> intersect() always fails" flag
// (triangle.rs:254-261, evaluated once here instead of per accepted candidate).
__global__ void k_build_packets(DeviceScene s, const uint32_t *ordered, uint32_t n_refs, TriPacket *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_refs) return;
    uint32_t prim = ordered[i];      // primitive index, or PT_TOP_INSTANCE | instance index
    TriPacket p;
    if (prim & PT_TOP_INSTANCE) {
        p.p0[0] = p.p0[1] = p.p0[2] = p.p1x = p.p1yz[0] = p.p1yz[1] = p.p2xy[0] = p.p2xy[1] = p.p2z = 0.0f;
        p.prim = PT_NONE; p.shape = prim & ~PT_TOP_INSTANCE; p.flags = TP_INSTANCE;
        out[i] = p;
        return;
    }
    uint32_t shape = s.prim_shape[prim];
    p.prim = prim; p.shape = shape; p.flags = 0;
    if ((shape >> 30) == PT_SHAPE_TRIANGLE) {
        uint32_t tri = shape & 0x3fffffffu;
        uint32_t i0 = s.indices[3 * tri], i1 = s.indices[3 * tri + 1], i2 = s.indices[3 * tri + 2];
        V3 p0 = ld3(s.P, i0), p1 = ld3(s.P, i1), p2 = ld3(s.P, i2);
        P2 uv[3]; tri_uvs(s, tri, i0, i1, i2, uv);
        V3 dpdu, dpdv;
        bool ok = tri_partials(p0, p1, p2, uv, dpdu, dpdv);
        p.p0[0] = p0.x; p.p0[1] = p0.y; p.p0[2] = p0.z; p.p1x = p1.x;
        p.p1yz[0] = p1.y; p.p1yz[1] = p1.z; p.p2xy[0] = p2.x; p.p2xy[1] = p2.y; p.p2z = p2.z;
        p.flags = (uint32_t)s.tri_flags[tri] | (ok ? 0u : (uint32_t)TP_BOGUS);
        if ((s.tri_alpha && s.tri_alpha[tri] >= 0) || (s.tri_shadow_alpha && s.tri_shadow_alpha[tri] >= 0)) p.flags |= TP_ALPHA;
    } else {
        p.p0[0] = p.p0[1] = p.p0[2] = p.p1x = p.p1yz[0] = p.p1yz[1] = p.p2xy[0] = p.p2xy[1] = p.p2z = 0.0f;
        p.flags = TP_SPHERE;
    }
    out[i] = p;
}

// Mark the last packet of every leaf (offsets of the last primitive of each leaf, computed on the host).
__global__ void k_mark_leaf_ends(TriPacket *leaf, const uint32_t *last_index, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) leaf[last_index[i]].flags |= TP_LAST;
}

__global__ void k_light_area(DeviceScene s, float *area) {  // DiffuseAreaLight::new -> shape.area()
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n_lights) return;
    float a = 0.0f;
    const PtLight &L = s.lights[i];
    if (L.type == PT_LIGHT_DIFFUSE_AREA) {
        uint32_t shape = s.prim_shape[L.prim];
        if ((shape >> 30) == PT_SHAPE_TRIANGLE) {
            uint32_t tri = shape & 0x3fffffffu;
            a = tri_area(ld3(s.P, s.indices[3 * tri]), ld3(s.P, s.indices[3 * tri + 1]), ld3(s.P, s.indices[3 * tri + 2]));
        } else {  // Sphere::area (sphere.rs:291-293)
            const PtSphere &S = s.spheres[shape & 0x3fffffffu];
            a = S.kind == PT_QUADRIC_DISK ? S.phi_max * 0.5f * (S.radius * S.radius - S.inner_radius * S.inner_radius)   // Disk::area (disk.rs:120-122)
                                          : S.phi_max * S.radius * (S.z_max - S.z_min);
        }
    }
    area[i] = a;
}

// ---- BVH traversal ---------------------------------------------------------------------------------

// Persistent waves ("persistent threads"): every lane owns one ray at a time; lanes whose ray has finished are
// refilled from the queue with one atomicAdd per wave, so a wave stays populated until the queue drains.
//
// The reference visits nodes one at a time (test node, then push far child / descend near child, bvh.rs:728-751).
// Here one 64-byte record per interior node carries BOTH children's bounds, so a ray performs one dependent fetch
// per interior node it enters instead of one per node it tests. Results and counters stay those of the reference:
//  * the near child is tested immediately with the current t_max -- exactly when the reference tests it;
//  * the far child's slab arithmetic is evaluated now but its `tmin < ray.t_max` comparison is deferred to pop time
//    (tmin is kept on the stack), which is when the reference performs the whole test with the then-current t_max;
//  * a far child whose t_max-independent part already fails is not pushed; the reference would pop, test and
//    discard it later, so the number of such skipped entries lying directly below each pushed entry is carried along
//    (6 bits in the stack word) and added to the node-visit counter at the moment the reference would pop them.
// Node steps and leaf (triangle) work run in separate phases so that neither executes with a mostly idle wave.

// GEN = the scene has spheres and/or object instances (lean triangle-only code otherwise).
#ifndef PT_TRACE_ATTR
#define PT_TRACE_ATTR   // experiment hook: e.g. __attribute__((amdgpu_waves_per_eu(6,6))) -- measured: 6 waves/SIMD needs 64-72 B of
                        // scratch and loses 12 %; 5 waves/SIMD (the launch bound below) is free for both triangle-only kernels
#endif
// MODE: 0 = triangle-only scenes, 1 = general geometry (spheres, instances), 2 = general geometry + alpha-masked triangles
template <bool ANY, int MODE>
__global__ __launch_bounds__(kTraceBlock, (MODE == 0) ? 5 : 1) PT_TRACE_ATTR void k_trace(DeviceScene s, TraceJob job) {
    constexpr bool SPH = MODE >= 1, ALPHA = MODE == 2;
    __shared__ uint32_t lds_stack[(kTraceBlock / 64) * kLdsStack * 2 * 64];
    const uint32_t lane = lane_id();
    const uint32_t wave_in_block = threadIdx.x >> 6;
    uint32_t *stack = lds_stack + wave_in_block * (kLdsStack * 2 * 64) + lane;   // entry e: words at [2e*64], [(2e+1)*64]
    // spilled entries: [wave][word][lane], so that lanes at the same depth touch consecutive dwords
    uint32_t *spill = job.spill + (size_t)(blockIdx.x * (kTraceBlock / 64) + wave_in_block) * 64 * (2 * (kMaxStack - kLdsStack)) + lane;
    const uint32_t count = *job.count;
    const uint4 *wide4 = reinterpret_cast<const uint4 *>(s.wide);
    const uint4 *leaf4 = reinterpret_cast<const uint4 *>(s.leaf);
    uint32_t n_nodes = 0, n_tris = 0, n_rays = 0, n_sph = 0;
#ifdef PT_TRACE_UTIL   // SIMD utilisation study: wave iterations and active lanes of the node phase / the leaf phase
    uint32_t u_it1 = 0, u_act1 = 0, u_it2 = 0, u_act2 = 0;
#define PT_UTIL(it, act, pred) do { const unsigned long long m_ = __ballot(pred); if (pred) { act++; it += (lane == (uint32_t)(__ffsll((long long)m_) - 1)); } } while (0)
#else
#define PT_UTIL(it, act, pred) do { } while (0)
#endif

    // lane state: ST_IDLE (no ray), ST_ENTER (fetch record `cur`), ST_LEAF (test packets from `cur`), ST_DONE
    enum : uint32_t { ST_IDLE = 0, ST_ENTER = 1, ST_LEAF = 2, ST_DONE = 3 };
    uint32_t state = ST_IDLE;
    bool exhausted = false;
    constexpr int kChunk = 256;
    uint32_t chunk_next = 0, chunk_left = 0;   // wave-uniform
    uint32_t pid = 0, cur = 0, sp = 0, pending = 0;
    V3 ro, rd, inv_dir;
    TriRay tray; tray.kz = 2; tray.Sx = tray.Sy = tray.Sz = 0.0f;   // per-ray half of the triangle test
    bool nx = false, ny = false, nz = false, found = false;
    float t_max = 0.0f;
    uint32_t hit_prim = PT_NONE; float hit_t = 0.0f, hb0 = 0.0f, hb1 = 0.0f, hb2 = 0.0f;
    // instancing (primitive.rs:58-88): while inside an instance the lane's ray is the object-space ray
    uint32_t in_inst = PT_NONE, hit_inst = PT_NONE; float t_max_world = 0.0f; bool inst_hit = false;
    constexpr uint32_t kMarker = 0xFFC0DEADu;   // stack word 1 of an "end of instance" entry (never a real tmin)

    // Pop entries until one passes its deferred `tmin < t_max` test (or the stack is empty).
    auto pop_next = [&]() {
        for (;;) {
            n_nodes += pending; pending = 0;          // skipped far children above the top entry: popped + failed
            if (sp == 0) { state = ST_DONE; return; }
            sp--;
            uint32_t w0, w1;
            if (sp < (uint32_t)kLdsStack) { w0 = stack[(2 * sp) * 64]; w1 = stack[(2 * sp + 1) * 64]; }
            else { w0 = spill[(2 * (sp - kLdsStack)) * 64]; w1 = spill[(2 * (sp - kLdsStack) + 1) * 64]; }
            if (SPH && w1 == kMarker) {                // the object's BVH is exhausted: back to world space (primitive.rs:70-77)
                pending = (w0 >> 25) & 63u;            // the outer traversal's skipped entries
                ro = V3(job.ox[pid], job.oy[pid], job.oz[pid]); rd = V3(job.dx[pid], job.dy[pid], job.dz[pid]);
                inv_dir = V3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                nx = inv_dir.x < 0.0f; ny = inv_dir.y < 0.0f; nz = inv_dir.z < 0.0f;
                tray = tri_ray_setup(rd);
                t_max = inst_hit ? t_max : t_max_world;  // r.t_max = ray.t_max only when the instance was hit
                in_inst = PT_NONE; inst_hit = false;
                if (w0 & kLeafBit) { cur = w0 & kRefMask; state = ST_LEAF; return; }  // remaining packets of the outer leaf
                continue;
            }
            n_nodes++;                                 // the reference tests the popped node now
            pending = (w0 >> 25) & 63u;
            if (__uint_as_float(w1) < t_max) {         // deferred half of intersect_p2
                cur = w0 & kRefMask;
                state = (w0 & kLeafBit) ? ST_LEAF : ST_ENTER;
                return;
            }
        }
    };

    for (;;) {
        // ---- retire finished rays and refill their lanes, in batches: finished lanes wait (idle) until at least
        //      `refill_min` of them have accumulated or nothing else is running, so the queue atomics below are
        //      paid once per batch instead of once per ray.
        const unsigned long long donem = __ballot(state == ST_DONE || state == ST_IDLE);
        const unsigned long long busy = __ballot(state == ST_ENTER || state == ST_LEAF);
        if (donem != 0ull && ((uint32_t)__popcll(donem) >= job.refill_min || busy == 0ull)) {
            const bool retire = state == ST_DONE;
            if (retire) {
                if (ANY) job.out_occluded[pid] = found ? 1 : 0;
                else {
                    job.out_prim[pid] = hit_prim;
                    if (job.out_t) job.out_t[pid] = hit_t;
                    if (job.out_b0) { job.out_b0[pid] = hb0; job.out_b1[pid] = hb1; job.out_b2[pid] = hb2; }   // NULL: only the hit / miss matters (volpath shadow rays)
                    if (SPH && job.out_inst) job.out_inst[pid] = hit_inst;
                }
            }
            if (retire) state = ST_IDLE;
            if (!exhausted) {
                // work fetch: the wave reserves kChunk consecutive queue entries with one atomic and hands them
                // out over several refills (consecutive entries are spatially coherent rays)
                if (chunk_left == 0) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(job.head, (uint32_t)kChunk);
                    chunk_next = __shfl(base, 0);
                    chunk_left = (chunk_next < count) ? min((uint32_t)kChunk, count - chunk_next) : 0u;
                    if (chunk_left == 0) exhausted = true;
                }
                const uint32_t rank = (uint32_t)__popcll(donem & ((1ull << lane) - 1ull));
                const uint32_t take = min(chunk_left, (uint32_t)__popcll(donem));
                const uint32_t qi = chunk_next + rank;
                const bool get = state == ST_IDLE && rank < take;
                chunk_next += take; chunk_left -= take;
                if (get) {
                    pid = job.queue ? job.queue[qi] : qi;
                    ro = V3(job.ox[pid], job.oy[pid], job.oz[pid]);
                    rd = V3(job.dx[pid], job.dy[pid], job.dz[pid]);
                    t_max = job.tmax ? job.tmax[pid] : job.scalar_tmax;
                    inv_dir = V3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                    nx = inv_dir.x < 0.0f; ny = inv_dir.y < 0.0f; nz = inv_dir.z < 0.0f;
                    tray = tri_ray_setup(rd);
                    sp = 0; pending = 0; found = false;
                    hit_prim = PT_NONE; hit_t = 0.0f; hb0 = hb1 = hb2 = 0.0f;
                    in_inst = PT_NONE; hit_inst = PT_NONE; inst_hit = false;
                    n_rays++;
                    state = ST_DONE;
                    if (s.n_nodes > 0) {  // the root node's own test (bvh.rs:725-727)
                        n_nodes++;
                        if (slab_test(s.root_min, s.root_max, ro, inv_dir, nx, ny, nz, t_max)) {
                            cur = s.root_ref & kRefMask;
                            state = (s.root_ref & kLeafBit) ? ST_LEAF : ST_ENTER;
                        }
                    }
                }
            }
        }
        if (__ballot(state != ST_IDLE) == 0ull) break;   // queue drained and every lane retired

        // ---- one record per lane and iteration: a lane at an interior node fetches its 64-byte two-wide record, a lane at a leaf
        //      its next 48-byte packet; both kinds of fetch are in flight together and nobody waits for a phase change
        // lanes at a leaf join in once `leaf_quorum` of them wait (or no lane is at a node), so the triangle test is not
        // executed for a handful of lanes in every iteration
        const bool at_node = state == ST_ENTER;
        const unsigned long long leaf_m = __ballot(state == ST_LEAF);
        const bool at_leaf = state == ST_LEAF && ((uint32_t)__popcll(leaf_m) >= job.leaf_quorum || __ballot(at_node) == 0ull);
        PT_UTIL(u_it1, u_act1, at_node || at_leaf);
        PT_UTIL(u_it2, u_act2, at_leaf);
        if (at_node || at_leaf) {
            const uint4 *rec = at_leaf ? leaf4 + 3 * (size_t)cur : wide4 + 4 * (size_t)cur;
            const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];   // q3 of a packet = start of the next one (array is padded)
            // Pin the whole record in front of the node / leaf branch: left alone, the compiler sinks the fields only the node path
            // reads (q2.z, q3.z) below the branch as two more dword loads, i.e. a second dependent L1 round trip in every node step
            // (measured: extend 143 -> 126 ms per step).
            asm volatile("" :: "v"(q2.z), "v"(q3.x), "v"(q3.y), "v"(q3.z));
            bool need_pop = false;
            if (at_node) {
                const float lmin[3] = {__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z)};
                const float lmax[3] = {__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y)};
                const float rmin[3] = {__uint_as_float(q1.z), __uint_as_float(q1.w), __uint_as_float(q2.x)};
                const float rmax[3] = {__uint_as_float(q2.y), __uint_as_float(q2.z), __uint_as_float(q2.w)};
                const uint32_t axis = q3.z & 0xffu;
                const bool neg = axis == 0 ? nx : (axis == 1 ? ny : nz);   // near child = right when the ray is negative along the split axis
                float tmin_l, tmin_r;
                const bool geo_l = slab_geo(lmin, lmax, ro, inv_dir, nx, ny, nz, tmin_l);
                const bool geo_r = slab_geo(rmin, rmax, ro, inv_dir, nx, ny, nz, tmin_r);
                const bool geo_near = neg ? geo_r : geo_l, geo_far = neg ? geo_l : geo_r;
                const float tmin_near = neg ? tmin_r : tmin_l, tmin_far = neg ? tmin_l : tmin_r;
                const uint32_t near_ref = neg ? q3.y : q3.x, far_ref = neg ? q3.x : q3.y;
                // reference: push far, cur = near, test near
                if (geo_far) {
                    if (pending > 63u || sp >= (uint32_t)kMaxStack) atomicMax(job.error, (uint32_t)PT_ERR_STACK_OVERFLOW);
                    else {
                        const uint32_t w0 = far_ref | (pending << 25), w1 = __float_as_uint(tmin_far);
                        if (sp < (uint32_t)kLdsStack) { stack[(2 * sp) * 64] = w0; stack[(2 * sp + 1) * 64] = w1; }
                        else { spill[(2 * (sp - kLdsStack)) * 64] = w0; spill[(2 * (sp - kLdsStack) + 1) * 64] = w1; }
                        sp++; pending = 0;
                    }
                } else pending++;
                n_nodes++;  // the near child's test
                if (geo_near && tmin_near < t_max) {
                    cur = near_ref & kRefMask;
                    state = (near_ref & kLeafBit) ? ST_LEAF : ST_ENTER;
                } else need_pop = true;
            } else {
                // leaf packets in ordered_prims order
                const uint32_t fl = q2.w, li = cur;
                bool advance = true;   // false: the lane left the leaf (entered an instance / finished an any-hit ray)
                if (fl & TP_INSTANCE) {
                    if constexpr (SPH) {  // TransformedPrimitive::intersect / intersect_p (primitive.rs:58-88)
                        const DevInstance &I = s.instances[q2.z];
                        // ray = inverse(prim_to_world).transform_ray(r)  (transform.rs:543-577, t_max -= dt)
                        const M4 w2i = ldm4g(I.world_to_instance);
                        V3 oerr; V3 o2 = xf_point_err(w2i, ro, oerr); const V3 d2 = xf_vector(w2i, rd);
                        const float l2 = length_squared(d2);
                        float tm2 = t_max;
                        if (l2 > 0.0f) { const float dt = dot(vabs(d2), oerr) / l2; o2 = o2 + d2 * dt; tm2 -= dt; }
                        const V3 inv2(1.0f / d2.x, 1.0f / d2.y, 1.0f / d2.z);
                        const bool nx2 = inv2.x < 0.0f, ny2 = inv2.y < 0.0f, nz2 = inv2.z < 0.0f;
                        bool enter = true;
                        if (!I.single) { n_nodes++; enter = slab_test(I.root_min, I.root_max, o2, inv2, nx2, ny2, nz2, tm2); }  // object BVH root (bvh.rs:725-727)
                        if (enter) {
                            // remember where to resume: the rest of this leaf (if any) and the outer skip count
                            const bool more = !(fl & TP_LAST);
                            const uint32_t w0 = (more ? (kLeafBit | ((li + 1u) & kRefMask)) : 0u) | (pending << 25);
                            if (pending > 63u || sp >= (uint32_t)kMaxStack) atomicMax(job.error, (uint32_t)PT_ERR_STACK_OVERFLOW);
                            else {
                                if (sp < (uint32_t)kLdsStack) { stack[(2 * sp) * 64] = w0; stack[(2 * sp + 1) * 64] = kMarker; }
                                else { spill[(2 * (sp - kLdsStack)) * 64] = w0; spill[(2 * (sp - kLdsStack) + 1) * 64] = kMarker; }
                                sp++; pending = 0;
                                t_max_world = t_max; in_inst = q2.z; inst_hit = false;
                                ro = o2; rd = d2; inv_dir = inv2; nx = nx2; ny = ny2; nz = nz2; t_max = tm2;
                                tray = tri_ray_setup(rd);
                                cur = I.root_ref & kRefMask;
                                state = (I.root_ref & kLeafBit) ? ST_LEAF : ST_ENTER;
                                advance = false;
                            }
                        }
                    }
                } else if (fl & TP_SPHERE) {
                    if constexpr (SPH) {  // GeometricPrimitive -> Sphere::intersect / intersect_p (sphere.rs:59-286)
                        n_sph++;
                        float t, phi; V3 ph, dobj;
                        if (sphere_hit(s.spheres[q2.z & 0x3fffffffu], ro, rd, t_max, ANY, t, ph, phi, dobj)) {
                            found = true;
                            if (ANY) { state = ST_DONE; advance = false; }
                            else {
                                t_max = t;
                                hit_prim = q2.y; hit_t = t; hb0 = hb1 = hb2 = 0.0f;
                                hit_inst = in_inst; inst_hit = in_inst != PT_NONE;
                            }
                        }
                    }
                } else {
                    n_tris++;
                    V3 p0(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
                    V3 p1(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
                    V3 p2(__uint_as_float(q1.z), __uint_as_float(q1.w), __uint_as_float(q2.x));
                    float t, b0, b1, b2;
                    bool hit = tri_hit_params(p0, p1, p2, ro, tray, t_max, t, b0, b1, b2);
                    if constexpr (ALPHA) {
                        // Triangle::intersect (triangle.rs:275-285) / intersect_p (:497-545) with an alpha mask: the hit is
                        // discarded where the mask evaluates to 0; intersect_p then also rejects degenerate triangles
                        if (hit && (fl & TP_ALPHA) && !(fl & TP_BOGUS)) {
                            const uint32_t tri = q2.z & 0x3fffffffu;
                            P2 uv[3]; tri_uvs(s, tri, s.indices[3 * tri], s.indices[3 * tri + 1], s.indices[3 * tri + 2], uv);
                            TexCtx c; c.dpdx = V3(0.0f, 0.0f, 0.0f); c.dpdy = V3(0.0f, 0.0f, 0.0f); c.dudx = c.dvdx = c.dudy = c.dvdy = 0.0f;
                            c.p = p0 * b0 + p1 * b1 + p2 * b2;
                            c.uv = P2(uv[0].x * b0 + uv[1].x * b1 + uv[2].x * b2, uv[0].y * b0 + uv[1].y * b1 + uv[2].y * b2);
                            const int32_t a = s.tri_alpha ? s.tri_alpha[tri] : -1;
                            if (a >= 0 && tex_eval(s, a, c).r == 0.0f) hit = false;
                            if (ANY && hit) { const int32_t sa = s.tri_shadow_alpha ? s.tri_shadow_alpha[tri] : -1; if (sa >= 0 && tex_eval(s, sa, c).r == 0.0f) hit = false; }
                        } else if (ANY && hit && (fl & TP_ALPHA) && (fl & TP_BOGUS)) hit = false;
                    }
                    if (hit) {
                        if (ANY) { found = true; state = ST_DONE; advance = false; }
                        else if (!(fl & TP_BOGUS)) {  // triangle.rs:258-261
                            found = true; t_max = t;  // primitive.rs:137
                            hit_prim = q2.y; hit_t = t; hb0 = b0; hb1 = b1; hb2 = b2;
                            if (SPH) { hit_inst = in_inst; inst_hit = in_inst != PT_NONE; }
                        }
                    }
                }
                if (advance) { if (fl & TP_LAST) need_pop = true; else cur = li + 1u; }
            }
            if (need_pop) pop_next();
        }
    }
    counter_add(&job.counters->nodes, n_nodes);
    counter_add(&job.counters->tri_tests, n_tris);
    if (SPH) counter_add(&job.counters->sphere_tests, n_sph);
    counter_add(ANY ? &job.counters->shadow_tests : &job.counters->intersect_tests, n_rays);
    counter_add(&job.counters->k_nodes[job.kind], n_nodes);
    counter_add(&job.counters->k_tris[job.kind], n_tris);
    counter_add(&job.counters->k_rays[job.kind], n_rays);
#ifdef PT_TRACE_UTIL
    for (int o = 32; o > 0; o >>= 1) { u_it1 += __shfl_xor(u_it1, o); u_act1 += __shfl_xor(u_act1, o); u_it2 += __shfl_xor(u_it2, o); u_act2 += __shfl_xor(u_act2, o); }
    if (lane == 0) {
        atomicAdd(&job.counters->regions[4 * job.kind + 0], (unsigned long long)u_it1); atomicAdd(&job.counters->regions[4 * job.kind + 1], (unsigned long long)u_act1);
        atomicAdd(&job.counters->regions[4 * job.kind + 2], (unsigned long long)u_it2); atomicAdd(&job.counters->regions[4 * job.kind + 3], (unsigned long long)u_act2);
    }
#endif
}
template __global__ void k_trace<false, 0>(DeviceScene, TraceJob);
template __global__ void k_trace<true, 0>(DeviceScene, TraceJob);
template __global__ void k_trace<false, 1>(DeviceScene, TraceJob);
template __global__ void k_trace<true, 1>(DeviceScene, TraceJob);
template __global__ void k_trace<false, 2>(DeviceScene, TraceJob);
template __global__ void k_trace<true, 2>(DeviceScene, TraceJob);

// ---- material-class routing (material-sorted shade queues) ------------------------------------------------------
// Reads the hit record of every traced continuation ray and appends the path id to the shade queue of the hit
// material's class (escaped rays -> the miss class). Block-level staged appends: one global atomic per ~1000 entries per class.
__global__ __launch_bounds__(256) void k_route(DeviceScene s, const uint32_t *queue, const uint32_t *count_ptr, const uint32_t *hit_prim,
                                              uint32_t *class_count, uint32_t *c0, uint32_t *c1, uint32_t *c2, uint32_t *c3, uint32_t *c4) {
    __shared__ LdsQueue<1024> q0, q1, q2, q3, q4;
    lq_init(q0); lq_init(q1); lq_init(q2); lq_init(q3); lq_init(q4);
    __syncthreads();
    const uint32_t count = *count_ptr;
    const uint32_t rounded = (count + 255u) & ~255u;
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < rounded; qi += gridDim.x * blockDim.x) {
        const bool valid = qi < count;
        uint32_t pid = 0, cls = (uint32_t)kMissClass;   // escaped rays: their own light kernel (k_shade_miss)
        if (valid) {
            pid = queue[qi];
            const uint32_t hp = hit_prim[pid];
            if (hp != PT_NONE) { const uint32_t m = s.prim_material[hp]; cls = (m == PT_NONE) ? 0u : (uint32_t)s.mat_class[m]; }
        }
        lq_push(q0, pid, valid && cls == 0u); lq_push(q1, pid, valid && cls == 1u);
        lq_push(q2, pid, valid && cls == 2u); lq_push(q3, pid, valid && cls == 3u); lq_push(q4, pid, valid && cls == 4u);
        __syncthreads();
        lq_flush_nosync(q0, class_count + 0, c0, 256u, false); lq_flush_nosync(q1, class_count + 1, c1, 256u, false);
        lq_flush_nosync(q2, class_count + 2, c2, 256u, false); lq_flush_nosync(q3, class_count + 3, c3, 256u, false);
        lq_flush_nosync(q4, class_count + 4, c4, 256u, false);
        __syncthreads();
    }
    lq_flush_nosync(q0, class_count + 0, c0, 0u, true); lq_flush_nosync(q1, class_count + 1, c1, 0u, true);
    lq_flush_nosync(q2, class_count + 2, c2, 0u, true); lq_flush_nosync(q3, class_count + 3, c3, 0u, true);
    lq_flush_nosync(q4, class_count + 4, c4, 0u, true);
}

// ---- camera rays -------------------------------------------------------------------------------------
// pixel slot -> pixel: slot = tile_slot*256 + ty*16 + tx, tile index = tile_rank + tile_slot*tile_world
PT_DEV bool slot_to_pixel(const RenderConst &rc, uint32_t slot, int32_t &px, int32_t &py) {
    uint32_t tile_slot = slot >> 8, in_tile = slot & 255u;
    uint32_t tile = rc.tile_rank + tile_slot * rc.tile_world;
    uint32_t tx = tile % rc.ntx, ty = tile / rc.ntx;
    px = rc.sample_bounds[0] + (int32_t)(tx * 16u + (in_tile & 15u));
    py = rc.sample_bounds[1] + (int32_t)(ty * 16u + (in_tile >> 4));
    if (ty >= rc.nty || px >= rc.sample_bounds[2] || py >= rc.sample_bounds[3]) return false;
    // integrator.rs:328: pixels outside the integrator's pixel_bounds are skipped
    return px >= rc.pixel_bounds[0] && px < rc.pixel_bounds[2] && py >= rc.pixel_bounds[1] && py < rc.pixel_bounds[3];
}

PT_DEV void camera_ray(const RenderConst &rc, float pfx, float pfy, float time_u, P2 plens_u, V3 &o, V3 &d) {  // perspective.rs:120-179
    V3 pcamera = xf_point(rc.raster_to_camera, V3(pfx, pfy, 0.0f));
    V3 ro(0.0f, 0.0f, 0.0f), rd = normalize(pcamera);
    if (rc.lens_radius > 0.0f) {
        P2 dsk = concentric_sample_disk(plens_u);
        float lx = dsk.x * rc.lens_radius, ly = dsk.y * rc.lens_radius;
        float ft = rc.focal_distance / rd.z;
        V3 pfocus = ro + rd * ft;
        ro = V3(lx, ly, 0.0f);
        rd = normalize(pfocus - ro);
    }
    (void)time_u;  // ray.time = lerp(time, open, close) has no effect without animated transforms
    // Transform::transform_ray (transform.rs:543-577)
    V3 oerr;
    V3 ow = xf_point_err(rc.camera_to_world, ro, oerr);
    V3 dw = xf_vector(rc.camera_to_world, rd);
    float l2 = length_squared(dw);
    if (l2 > 0.0f) { float dt = dot(vabs(dw), oerr) / l2; ow = ow + dw * dt; }
    o = ow; d = dw;
}

__global__ __launch_bounds__(256) void k_generate(RenderConst rc, SobolTables tabs, PathSoA ps, uint32_t *q_ext, uint32_t *q_ext_count, DevCounters *counters) {
    // LDS copies of what every lane needs: Sobol' rows of dimensions 0..4 and the two van-der-Corput matrices of m
    __shared__ uint32_t s_rows[5 * 52];
    __shared__ uint64_t s_vdc[2 * 52];
    __shared__ LdsQueue<1024> s_q;
    lq_init(s_q);
    const uint32_t m = (uint32_t)rc.sobol.log2_resolution;
    for (uint32_t i = threadIdx.x; i < 5 * 52; i += blockDim.x) s_rows[i] = tabs.m32[i];
    if (m > 0) for (uint32_t i = threadIdx.x; i < 2 * 52; i += blockDim.x) s_vdc[i] = (i < 52) ? tabs.vdc[(m - 1) * 52 + i] : tabs.vdc_inv[(m - 1) * 52 + (i - 52)];
    __syncthreads();
    const uint32_t total = rc.n_pix_slots * rc.s_count;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounded = (total + 255u) & ~255u;   // whole blocks iterate together (block-level queue flushes)
    unsigned long long n_alive = 0;
    for (uint32_t pid = blockIdx.x * blockDim.x + threadIdx.x; pid < rounded; pid += stride) {
        bool alive = false;
        if (pid < total) {
            const uint32_t slot = pid % rc.n_pix_slots, sl = pid / rc.n_pix_slots;
            int32_t px, py;
            if (slot_to_pixel(rc, slot, px, py)) {
                const uint64_t sample = rc.s_begin + sl;
                if (rc.halton.enabled) {   // HaltonSampler: same GlobalSampler bookkeeping, its own index and dimensions (halton.rs:122-165)
                    const uint64_t index = halton_index_for_sample(rc.halton, px, py, sample);
                    const float fx = halton_sample_dimension(tabs, rc.halton, index, 0u), fy = halton_sample_dimension(tabs, rc.halton, index, 1u);
                    const float pfx = (float)px + fx, pfy = (float)py + fy;   // get_camera_sample: p_film = pixel + get_2d() (sampler.rs:170-180)
                    const float tm = halton_sample_dimension(tabs, rc.halton, index, 2u);
                    const P2 pl(halton_sample_dimension(tabs, rc.halton, index, 3u), halton_sample_dimension(tabs, rc.halton, index, 4u));
                    V3 o, d;
                    camera_ray(rc, pfx, pfy, tm, pl, o, d);
                    ps.pfilm_x[pid] = pfx; ps.pfilm_y[pid] = pfy;
                    ps.ox[pid] = o.x; ps.oy[pid] = o.y; ps.oz[pid] = o.z;
                    ps.dx[pid] = d.x; ps.dy[pid] = d.y; ps.dz[pid] = d.z;
                    ps.beta_r[pid] = 1.0f; ps.beta_g[pid] = 1.0f; ps.beta_b[pid] = 1.0f;
                    ps.L_r[pid] = 0.0f; ps.L_g[pid] = 0.0f; ps.L_b[pid] = 0.0f;
                    ps.etascale[pid] = 1.0f;
                    ps.sobol_index[pid] = index;
                    ps.meta[pid] = 5u | (PF_CAMERA_RAY << 24);
                    alive = true;
                } else {
                const uint64_t index = sobol_interval_to_index(s_vdc, s_vdc + 52, m, sample, (uint32_t)(px - rc.sobol.sb_min[0]), (uint32_t)(py - rc.sobol.sb_min[1]));
                // get_camera_sample (sampler.rs:170-180): pfilm = get_2d, time = get_1d, plens = get_2d; one pass over the index bits
                uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                for (uint64_t a = index; a != 0; a &= a - 1) {
                    const int i = __builtin_ctzll(a);
                    v0 ^= s_rows[i]; v1 ^= s_rows[52 + i]; v2 ^= s_rows[104 + i]; v3 ^= s_rows[156 + i]; v4 ^= s_rows[208 + i];
                }
                // sobol.rs:77-81: film dimensions are remapped to the pixel
                float fx = sobol_to_float(v0) * (float)rc.sobol.resolution + (float)rc.sobol.sb_min[0];
                fx = clampf(fx - (float)px, 0.0f, kOneMinusEps);
                float fy = sobol_to_float(v1) * (float)rc.sobol.resolution + (float)rc.sobol.sb_min[1];
                fy = clampf(fy - (float)py, 0.0f, kOneMinusEps);
                const float pfx = (float)px + fx, pfy = (float)py + fy;
                V3 o, d;
                camera_ray(rc, pfx, pfy, sobol_to_float(v2), P2(sobol_to_float(v3), sobol_to_float(v4)), o, d);
                ps.pfilm_x[pid] = pfx; ps.pfilm_y[pid] = pfy;
                ps.ox[pid] = o.x; ps.oy[pid] = o.y; ps.oz[pid] = o.z;
                ps.dx[pid] = d.x; ps.dy[pid] = d.y; ps.dz[pid] = d.z;
                ps.beta_r[pid] = 1.0f; ps.beta_g[pid] = 1.0f; ps.beta_b[pid] = 1.0f;
                ps.L_r[pid] = 0.0f; ps.L_g[pid] = 0.0f; ps.L_b[pid] = 0.0f;
                ps.etascale[pid] = 1.0f;
                ps.sobol_index[pid] = index;
                ps.meta[pid] = 5u | (PF_CAMERA_RAY << 24);  // dimension 5 after the camera sample, bounces 0, flags: camera ray
                alive = true;
                }
            }
        }
        if (alive && rc.volpath) ps.medium[pid] = rc.camera_medium;   // the camera ray starts in the camera's medium (perspective.rs:114)
        lq_push(s_q, pid, alive);
        lq_sync_flush(s_q, q_ext_count, q_ext, 256u, false);
        n_alive += alive ? 1ull : 0ull;
    }
    lq_sync_flush(s_q, q_ext_count, q_ext, 0u, true);
    counter_add(&counters->camera_rays, n_alive);
}

// ---- shading ---------------------------------------------------------------------------------------------

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
    const uint32_t li = ps.nee_light[pid];
    // volpath: VisibilityTester::tr intersects (closest hit) and every surface is opaque; the segment's transmittance is already in A
    if ((flags & PF_PEND_SHADOW) && (VOL ? ps.sh_prim[pid] == PT_NONE : !ps.occluded[pid])) Ld = Ld + RGB(ps.A_r[pid], ps.A_g[pid], ps.A_b[pid]);
    if (flags & PF_PEND_MIS) {
        const PtLight &Lt = s.lights[li];
        V3 wi(ps.mis_dx[pid], ps.mis_dy[pid], ps.mis_dz[pid]);
        RGB lrad(0.0f);
        const uint32_t mp = ps.mis_prim[pid];
        if (mp != PT_NONE) {
            if (s.prim_light[mp] == li) {  // Arc::ptr_eq(light), integrator.rs:222-228
                SurfaceInteraction lsi;
                fill_hit<SPH>(s, mp, PT_NONE, V3(ps.mis_ox[pid], ps.mis_oy[pid], ps.mis_oz[pid]), wi, ps.mis_b0[pid], ps.mis_b1[pid], ps.mis_b2[pid], lsi);  // lights are never inside instances (api.rs:1605-1608)
                lrad = area_l(Lt, lsi.n, -wi);
            }
        } else { PT_T(2); lrad = light_le(s, Lt, wi); PT_T(1); }
        if (!lrad.is_black()) {
            RGB f(ps.mis_f_r[pid], ps.mis_f_g[pid], ps.mis_f_b[pid]);
            RGB Tr(1.0f);
            if (VOL) {   // Scene::intersect_tr (scene.rs:68-87): transmittance of the MIS ray's medium up to its hit (or to infinity)
                const uint32_t mm = ps.mis_medium[pid];
                if (mm != PT_NONE) Tr = Tr * medium_tr(s.media[mm], mp != PT_NONE ? ps.mis_t[pid] : PT_INF, wi);
            }
            Ld = Ld + f * lrad * Tr * ps.mis_w[pid] / ps.mis_spdf[pid];
        }
    }
    RGB nb(ps.nb_r[pid], ps.nb_g[pid], ps.nb_b[pid]);
    RGB Ldb = nb * (Ld / ps.nee_choice_pdf[pid]);
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
                    ps.sh_ox[pid] = so.x; ps.sh_oy[pid] = so.y; ps.sh_oz[pid] = so.z;
                    ps.sh_dx[pid] = sd.x; ps.sh_dy[pid] = sd.y; ps.sh_dz[pid] = sd.z;
                    ps.A_r[pid] = A.r; ps.A_g[pid] = A.g; ps.A_b[pid] = A.b;
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
                        ps.mis_ox[pid] = mo.x; ps.mis_oy[pid] = mo.y; ps.mis_oz[pid] = mo.z;
                        ps.mis_dx[pid] = wi.x; ps.mis_dy[pid] = wi.y; ps.mis_dz[pid] = wi.z;
                        ps.mis_f_r[pid] = f.r; ps.mis_f_g[pid] = f.g; ps.mis_f_b[pid] = f.b;
                        ps.mis_w[pid] = weight; ps.mis_spdf[pid] = scattpdf;
                        if (VOL) ps.mis_medium[pid] = medium_toward(mif, it.n, wi);
                        flags |= PF_PEND_MIS; push_mis = true; nee_pending = true; n_bytes += 24 + 12 + 8 + 4;
                    }
                }
            }
            if (nee_pending) {
                ps.nee_light[pid] = li; ps.nee_choice_pdf[pid] = choice_pdf;
                ps.nb_r[pid] = beta.r; ps.nb_g[pid] = beta.g; ps.nb_b[pid] = beta.b; n_bytes += 8 + 12;
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

#ifndef PT_SHADE_ATTR
#define PT_SHADE_ATTR   // experiment hook: e.g. __attribute__((amdgpu_waves_per_eu(4,4)))
#endif
// MODE: 0 = triangle-only scenes, 1 = general geometry (spheres and/or instances), 2 = general geometry + textures
// DIFF: the launch serves class 0 (matte materials: Lambertian / Oren-Nayar lobes only)
template <int MAXL, int MODE, bool DIFF>
__global__ __launch_bounds__(256, (MAXL == 1 && MODE == 1) ? 2 : 1) PT_SHADE_ATTR void k_shade(DeviceScene s, RenderConst rc, SobolTables tabs, LightGrid grid, PathSoA ps, ShadeJob job) {
    constexpr bool SPH = MODE >= 1, TEX = MODE >= 2, VOL = MODE == 3;   // MODE 3: general + textures + participating media (volpath.rs)
    __shared__ uint32_t s_sobol[kSobolLdsWords];
    // Block-level queues on purpose: their barriers keep the four waves of a block in lockstep through this very large
    // kernel, which measured 10 % faster than barrier-free per-wave queues (WaveQueue) at the same occupancy.
    __shared__ LdsQueue<1024> s_qext, s_qres, s_qsh, s_qmis;
    __shared__ LdsQueue<(MAXL == 5) ? 1024 : 1> s_qprobe;
    __shared__ uint32_t s_hist[16];
    lq_init(s_qext); lq_init(s_qres); lq_init(s_qsh); lq_init(s_qmis); lq_init(s_qprobe);
    if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
#ifdef PT_REGION_PROFILE
    __shared__ long long s_pt[4]; __shared__ int s_pr[4]; __shared__ unsigned long long s_pacc[64];
    if (threadIdx.x < 64) s_pacc[threadIdx.x] = 0;
    if (threadIdx.x < 4) { s_pt[threadIdx.x] = clock64(); s_pr[threadIdx.x] = 15; }
    Prof prof{s_pt, s_pr, s_pacc};
#endif
    sobol_stage_lds(s_sobol, tabs.m32, threadIdx.x, blockDim.x);
    __syncthreads();
    const uint32_t count = *job.count;
    const uint32_t rounded = (count + 255u) & ~255u;   // whole blocks iterate together
    unsigned long long zero_num = 0, zero_den = 0, n_valid = 0, n_bytes = 0;  // n_bytes: path-state + queue bytes
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < rounded; qi += gridDim.x * blockDim.x) {
    const bool valid = qi < count;
    bool push_ext = false, push_resolve = false, push_shadow = false, push_mis = false, push_probe = false;
    int finished_bounces = -1;
    uint32_t pid = 0;
    PT_T(0);
    if (valid) {
        n_valid++;
        n_bytes += 4 + 4 + 8 + 12 + 12 + /* write back */ 12 + 12 + 4;   // queue, meta, sobol index, L, beta
        pid = job.queue[qi];
        uint32_t meta = ps.meta[pid];
        uint32_t flags = meta >> 24, bounces = (meta >> 16) & 0xffu;
        Sampler smp; smp.index = ps.sobol_index[pid]; smp.dim = meta & 0xffffu; smp.m32 = tabs.m32; smp.lds = s_sobol; smp.overflow = false; smp.halton = MODE >= 1 && rc.halton.enabled != 0;   /* Halton scenes run the general kernels: the triangle-only ones stay Sobol'-only */ smp.prime = tabs.prime; smp.prime_sum = tabs.prime_sum; smp.perm = tabs.perm;
        smp.base = 0xffffffffu;
        RGB L(ps.L_r[pid], ps.L_g[pid], ps.L_b[pid]);
        RGB beta(ps.beta_r[pid], ps.beta_g[pid], ps.beta_b[pid]);

        // -- resolve the previous vertex's next-event estimation (integrator.rs:150-171,199-233)
        resolve_pending<SPH, VOL>(s, ps, pid, flags, L, zero_num, n_bytes PT_PROF_PASS);

        PT_T(3);
        if (flags & PF_DEAD) {
            finished_bounces = (int)bounces;
        } else {
            n_bytes += 24 + 16;  // ray + hit record
            V3 ro(ps.ox[pid], ps.oy[pid], ps.oz[pid]), rd(ps.dx[pid], ps.dy[pid], ps.dz[pid]);
            const uint32_t hp = ps.hit_prim[pid];
            const bool found = hp != PT_NONE;
            SurfaceInteraction si;
            if (found) fill_hit<SPH>(s, hp, SPH ? ps.hit_inst[pid] : PT_NONE, ro, rd, ps.hit_b0[pid], ps.hit_b1[pid], ps.hit_b2[pid], si);
            // path.rs:106-117
            if (bounces == 0 || (flags & PF_SPECULAR)) {
                if (found) {
                    const uint32_t al = s.prim_light[hp];
                    if (al != PT_NONE) L = L + area_l(s.lights[al], si.n, -rd) * beta;
                } else {
                    for (uint32_t k = 0; k < s.n_infinite; ++k) L = L + light_le(s, s.lights[s.infinite_lights[k]], rd) * beta;
                }
            }
            bool terminated = !found || bounces >= rc.max_depth;  // path.rs:120
            if (!terminated) {
                PT_T(4);
                smp.load_window();
                PT_T(10);
                Bsdf<MAXL, DIFF> bsdf;
                const uint32_t mi = s.prim_material[hp];
                bool has_bsdf = false;
                if (TEX) {
                    // compute_scattering_functions -> compute_differentials(ray) (interaction.rs:262-342): only the camera ray
                    // carries differentials; every spawned ray has none
                    RayDiff rdiff; rdiff.has = false;
                    if (flags & PF_CAMERA_RAY) {
                        P2 plens_u(0.0f, 0.0f);
                        if (rc.lens_radius > 0.0f) plens_u = rc.halton.enabled ? P2(halton_sample_dimension(tabs, rc.halton, smp.index, 3u), halton_sample_dimension(tabs, rc.halton, smp.index, 4u))
                                                                              : P2(sobol_sample_float(s_sobol, smp.index, 3u), sobol_sample_float(s_sobol, smp.index, 4u));
                        rdiff = camera_ray_differentials(rc, ps.pfilm_x[pid], ps.pfilm_y[pid], plens_u, ro, rd);
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
                } else has_bsdf = (mi != PT_NONE) && build_bsdf(s.materials[mi], si, bsdf, ConstMatEval(), s.materials);
                flags &= ~PF_CAMERA_RAY;
                IData it; it.p = si.p; it.p_error = si.p_error; it.n = si.n;
                MedIface mif{PT_NONE, PT_NONE};
                if (VOL) mif = surface_iface(s, hp, ps.medium[pid]);   // primitive.rs:139-145
                if (!has_bsdf) {  // path.rs:124-129: skip the surface, bounces unchanged
                    V3 o; spawn_ray(it, rd, o);
                    ps.ox[pid] = o.x; ps.oy[pid] = o.y; ps.oz[pid] = o.z;
                    if (VOL) {   // volpath.rs:127-131 `bounces -= 1; continue`: the count drops by one and wraps below zero
                        ps.medium[pid] = medium_toward(mif, si.n, rd);
                        bounces = (bounces - 1u) & 0xffu;
                    }
                    push_ext = true;
                } else {
                    const V3 wo = -rd;  // path.rs:148; estimate_direct uses isect.wo (== -rd for triangles, triangle.rs:296)
                    // uniform_sample_onelight (integrator.rs:81-106)
                    if (VOL) nee_vertex<SPH, Bsdf<MAXL, DIFF>, true, false>(s, grid, ps, pid, smp, si, it, bsdf, beta, flags, push_shadow, push_mis, n_bytes PT_PROF_PASS, mif);   // volpath.rs:136-138: unconditional
                    else if (bsdf.num_components(BSDF_ALL & ~BSDF_SPECULAR) > 0) {
                        zero_den++;
                        const bool nee_pending = nee_vertex<SPH>(s, grid, ps, pid, smp, si, it, bsdf, beta, flags, push_shadow, push_mis, n_bytes PT_PROF_PASS);
                        if (!nee_pending) zero_num++;  // Ld is black (path.rs:142)
                    }
                    // path.rs:148-174: sample the BSDF for the next direction
                    PT_T(11);
                    V3 wi; float pdf = 0.0f; int sflags = 0;
                    RGB f = bsdf.sample_f(wo, wi, smp.get_2d(), pdf, BSDF_ALL, sflags);
                    if (f.is_black() || pdf == 0.0f) terminated = true;
                    else {
                        beta = beta * (f * abs_dot(wi, si.sh_n) / pdf);
                        if (sflags & BSDF_SPECULAR) flags |= PF_SPECULAR; else flags &= ~PF_SPECULAR;
                        float etascale = ps.etascale[pid];
                        if ((sflags & BSDF_SPECULAR) && (sflags & BSDF_TRANSMISSION)) {
                            const float eta = bsdf.eta;
                            etascale *= (dot(wo, si.n) > 0.0f) ? eta * eta : 1.0f / (eta * eta);
                            ps.etascale[pid] = etascale;
                        }
                        V3 o; spawn_ray(it, wi, o);
                        bool rr_kill = false, to_probe = false;
                        if constexpr (MAXL == 5) {
                            // path.rs:177-183: importance sample the BSSRDF; the probe chain of sample_sp (bssrdf.rs:367-395)
                            // is walked by k_bssrdf over the following wavefront iterations
                            if ((s.materials[mi].type == PT_MAT_SUBSURFACE || disney_has_bssrdf(s.materials[mi])) && (sflags & BSDF_TRANSMISSION)) {
                                const P2 s2 = smp.get_2d();
                                const float s1 = smp.get_1d();
                                DevBssrdf bss; bss.init_material(s.materials[mi], s.bss_tables); bss.init_frame(si);
                                V3 start, target; float u1n = 0.0f;
                                const BssSoA &bs = job.bs;
                                if (!bss.probe_segment(s1, s2, start, target, u1n)) rr_kill = true;   // S is black: `break`
                                else {
                                    const V3 pd = target - start;
                                    if (pd.x == 0.0f && pd.y == 0.0f && pd.z == 0.0f) rr_kill = true;  // empty chain: nfound == 0
                                    else {
                                        bs.start_x[pid] = start.x; bs.start_y[pid] = start.y; bs.start_z[pid] = start.z;
                                        bs.target_x[pid] = target.x; bs.target_y[pid] = target.y; bs.target_z[pid] = target.z;
                                        bs.po_x[pid] = si.p.x; bs.po_y[pid] = si.p.y; bs.po_z[pid] = si.p.z;
                                        bs.ns_x[pid] = bss.ns.x; bs.ns_y[pid] = bss.ns.y; bs.ns_z[pid] = bss.ns.z;
                                        bs.ss_x[pid] = bss.ss.x; bs.ss_y[pid] = bss.ss.y; bs.ss_z[pid] = bss.ss.z;
                                        bs.u1n[pid] = u1n; bs.mat[pid] = mi; bs.cnt[pid] = 0u;
                                        // base = {p: start, p_error: 0, n: 0}: spawn_rayto_point leaves the origin at `start`
                                        ps.ox[pid] = start.x; ps.oy[pid] = start.y; ps.oz[pid] = start.z;
                                        ps.dx[pid] = pd.x; ps.dy[pid] = pd.y; ps.dz[pid] = pd.z;
                                        to_probe = true; push_probe = true; n_bytes += 18 * 4 + 24 + 4;
                                    }
                                }
                            }
                        }
                        // path.rs:206-214 Russian roulette
                        RGB rrbeta = beta * etascale;
                        if (!to_probe && !rr_kill && rrbeta.max_component_value() < rc.rr_threshold && bounces > 3) {
                            const float q = maxf(1.0f - rrbeta.max_component_value(), 0.05f);
                            if (smp.get_1d() < q) rr_kill = true;
                            else beta = beta / (1.0f - q);
                        }
                        if (rr_kill) terminated = true;
                        else if (!to_probe) {
                            bounces += 1;
                            ps.ox[pid] = o.x; ps.oy[pid] = o.y; ps.oz[pid] = o.z;
                            ps.dx[pid] = wi.x; ps.dy[pid] = wi.y; ps.dz[pid] = wi.z;
                            if (VOL) ps.medium[pid] = medium_toward(mif, si.n, wi);   // isect.spawn_ray(wi) (interaction.rs:32-36,54-66)
                            push_ext = true; n_bytes += 24 + 4 + 4;  // new ray, etascale, ext queue entry
                        }
                    }
                }
            }
            if (terminated) {
                if (flags & (PF_PEND_SHADOW | PF_PEND_MIS)) { flags |= PF_DEAD; push_resolve = true; }
                else finished_bounces = (int)bounces;
            }
        }
        PT_T(12);
        if (smp.overflow) atomicMax(job.error, (uint32_t)PT_ERR_SOBOL_DIMENSIONS);
        ps.L_r[pid] = L.r; ps.L_g[pid] = L.g; ps.L_b[pid] = L.b;
        ps.beta_r[pid] = beta.r; ps.beta_g[pid] = beta.g; ps.beta_b[pid] = beta.b;
        ps.meta[pid] = (smp.dim & 0xffffu) | ((bounces & 0xffu) << 16) | (flags << 24);
    }
    PT_T(13);
    lq_push(s_qext, pid, push_ext);
    lq_push(s_qres, pid, push_resolve);
    lq_push(s_qsh, pid, push_shadow);
    lq_push(s_qmis, pid, push_mis);
    if (finished_bounces >= 0) atomicAdd(&s_hist[finished_bounces > 15 ? 15 : finished_bounces], 1u);  // path.rs:219 (LDS)
    if constexpr (MAXL == 5) if (job.probe_next) lq_push(s_qprobe, pid, push_probe);
    __syncthreads();
    lq_flush_nosync(s_qext, job.ext_next_count, job.ext_next, 256u, false);
    lq_flush_nosync(s_qres, job.shade_next0_count, job.shade_next0, 256u, false);
    lq_flush_nosync(s_qsh, job.shadow_count, job.shadow, 256u, false);
    lq_flush_nosync(s_qmis, job.mis_count, job.mis, 256u, false);
    if constexpr (MAXL == 5) if (job.probe_next) lq_flush_nosync(s_qprobe, job.probe_next_count, job.probe_next, 256u, false);
    __syncthreads();
    }  // persistent loop over the queue
    lq_flush_nosync(s_qext, job.ext_next_count, job.ext_next, 0u, true);
    lq_flush_nosync(s_qres, job.shade_next0_count, job.shade_next0, 0u, true);
    lq_flush_nosync(s_qsh, job.shadow_count, job.shadow, 0u, true);
    lq_flush_nosync(s_qmis, job.mis_count, job.mis, 0u, true);
    if constexpr (MAXL == 5) if (job.probe_next) lq_flush_nosync(s_qprobe, job.probe_next_count, job.probe_next, 0u, true);
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
    counter_add(&job.counters->stages, n_valid);
    counter_add(&job.counters->shade_items[job.cls], n_valid);
    counter_add(&job.counters->shade_bytes[job.cls], n_bytes);
}
#define PT_INST_SHADE(L, S, D) template __global__ void k_shade<L, S, D>(DeviceScene, RenderConst, SobolTables, LightGrid, PathSoA, ShadeJob);
PT_INST_SHADE(1, 0, true) PT_INST_SHADE(1, 0, false) PT_INST_SHADE(2, 0, false) PT_INST_SHADE(5, 0, false)
PT_INST_SHADE(1, 1, true) PT_INST_SHADE(1, 1, false) PT_INST_SHADE(2, 1, false) PT_INST_SHADE(5, 1, false)
PT_INST_SHADE(1, 2, true) PT_INST_SHADE(1, 2, false) PT_INST_SHADE(2, 2, false) PT_INST_SHADE(5, 2, false)
PT_INST_SHADE(1, 3, true) PT_INST_SHADE(1, 3, false) PT_INST_SHADE(2, 3, false) PT_INST_SHADE(5, 3, false)   // volpath


// ---- volumetric path integrator: medium sampling between traversal and shading ---------------------------------------
// volpath.rs:98-104: after Scene::intersect, a ray that travels in a medium samples it (two sampler dimensions) and scales beta;
// a sampled medium vertex goes to the medium class, a black beta ends the path, everything else is routed as k_route does.
__global__ __launch_bounds__(256) void k_medium_route(DeviceScene s, RenderConst rc, SobolTables tabs, PathSoA ps, const uint32_t *queue, const uint32_t *count_ptr,
                                                     uint32_t *class_count, uint32_t *c0, uint32_t *c1, uint32_t *c2, uint32_t *c3, uint32_t *c4, uint32_t *c5, uint32_t *error) {
    __shared__ LdsQueue<1024> q0, q1, q2, q3, q4, q5;
    __shared__ uint32_t s_sobol[kSobolLdsWords];
    lq_init(q0); lq_init(q1); lq_init(q2); lq_init(q3); lq_init(q4); lq_init(q5);
    sobol_stage_lds(s_sobol, tabs.m32, threadIdx.x, blockDim.x);
    __syncthreads();
    const uint32_t count = *count_ptr;
    const uint32_t rounded = (count + 255u) & ~255u;
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < rounded; qi += gridDim.x * blockDim.x) {
        const bool valid = qi < count;
        uint32_t pid = 0, cls = (uint32_t)kMissClass;
        if (valid) {
            pid = queue[qi];
            const uint32_t hp = ps.hit_prim[pid];
            if (hp != PT_NONE) { const uint32_t m = s.prim_material[hp]; cls = (m == PT_NONE) ? 0u : (uint32_t)s.mat_class[m]; }
            const uint32_t med = ps.medium[pid];
            if (med != PT_NONE) {
                uint32_t meta = ps.meta[pid];
                Sampler smp; smp.index = ps.sobol_index[pid]; smp.dim = meta & 0xffffu; smp.m32 = tabs.m32; smp.lds = s_sobol; smp.overflow = false;
                smp.halton = rc.halton.enabled != 0; smp.prime = tabs.prime; smp.prime_sum = tabs.prime_sum; smp.perm = tabs.perm; smp.base = 0xffffffffu;
                const float u_channel = smp.get_1d(), u_dist = smp.get_1d();
                const V3 rd(ps.dx[pid], ps.dy[pid], ps.dz[pid]);
                bool sampled; float t;
                const RGB w = medium_sample(s.media[med], hp != PT_NONE ? ps.hit_t[pid] : PT_INF, rd, u_channel, u_dist, sampled, t);
                const RGB beta = RGB(ps.beta_r[pid], ps.beta_g[pid], ps.beta_b[pid]) * w;
                ps.beta_r[pid] = beta.r; ps.beta_g[pid] = beta.g; ps.beta_b[pid] = beta.b;
                uint32_t flags = meta >> 24;
                if (beta.is_black()) { flags |= PF_DEAD; cls = (uint32_t)kMissClass; }          // volpath.rs:105 `break`
                else if (sampled) { ps.hit_t[pid] = t; cls = (uint32_t)kMediumClass; }
                if (smp.overflow) atomicMax(error, (uint32_t)PT_ERR_SOBOL_DIMENSIONS);
                ps.meta[pid] = (smp.dim & 0xffffu) | (meta & 0x00ff0000u) | (flags << 24);
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
    __shared__ uint32_t s_sobol[kSobolLdsWords];
    __shared__ LdsQueue<1024> s_qext, s_qres, s_qsh, s_qmis;
    __shared__ uint32_t s_hist[16];
    lq_init(s_qext); lq_init(s_qres); lq_init(s_qsh); lq_init(s_qmis);
    if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
#ifdef PT_REGION_PROFILE
    __shared__ long long s_pt[4]; __shared__ int s_pr[4]; __shared__ unsigned long long s_pacc[64];
    if (threadIdx.x < 4) { s_pt[threadIdx.x] = clock64(); s_pr[threadIdx.x] = 15; }
    Prof prof{s_pt, s_pr, s_pacc};
#endif
    sobol_stage_lds(s_sobol, tabs.m32, threadIdx.x, blockDim.x);
    __syncthreads();
    const uint32_t count = *job.count;
    const uint32_t rounded = (count + 255u) & ~255u;
    unsigned long long zero_num = 0, n_valid = 0, n_bytes = 0;
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < rounded; qi += gridDim.x * blockDim.x) {
        const bool valid = qi < count;
        bool push_ext = false, push_resolve = false, push_shadow = false, push_mis = false;
        int finished_bounces = -1;
        uint32_t pid = 0;
        if (valid) {
            n_valid++;
            pid = job.queue[qi];
            const uint32_t meta = ps.meta[pid];
            uint32_t flags = meta >> 24, bounces = (meta >> 16) & 0xffu;
            Sampler smp; smp.index = ps.sobol_index[pid]; smp.dim = meta & 0xffffu; smp.m32 = tabs.m32; smp.lds = s_sobol; smp.overflow = false;
            smp.halton = rc.halton.enabled != 0; smp.prime = tabs.prime; smp.prime_sum = tabs.prime_sum; smp.perm = tabs.perm; smp.base = 0xffffffffu;
            RGB L(ps.L_r[pid], ps.L_g[pid], ps.L_b[pid]);
            RGB beta(ps.beta_r[pid], ps.beta_g[pid], ps.beta_b[pid]);
            resolve_pending<true, true>(s, ps, pid, flags, L, zero_num, n_bytes PT_PROF_PASS);
            flags &= ~PF_CAMERA_RAY;
            bool terminated = bounces >= rc.max_depth;   // volpath.rs:108
            if (!terminated) {
                smp.load_window();
                const V3 ro(ps.ox[pid], ps.oy[pid], ps.oz[pid]), rd(ps.dx[pid], ps.dy[pid], ps.dz[pid]);
                const uint32_t med = ps.medium[pid];
                SurfaceInteraction si;   // only p and wo are read through the MediumInteraction
                si.p = ro + rd * ps.hit_t[pid]; si.wo = -rd; si.n = V3(0.0f, 0.0f, 0.0f); si.sh_n = V3(0.0f, 0.0f, 0.0f); si.p_error = V3(0.0f, 0.0f, 0.0f);
                IData it; it.p = si.p; it.p_error = V3(0.0f, 0.0f, 0.0f); it.n = V3(0.0f, 0.0f, 0.0f);
                const PhaseBsdf phase{s.media[med].g, si.wo};
                nee_vertex<true, PhaseBsdf, true, true>(s, grid, ps, pid, smp, si, it, phase, beta, flags, push_shadow, push_mis, n_bytes PT_PROF_PASS, MedIface{med, med});
                V3 wi;
                hg_sample_p(phase.g, si.wo, wi, smp.get_2d());
                flags &= ~PF_SPECULAR;   // specular_bounce = false
                // Russian roulette (volpath.rs:171-176)
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
                    ps.ox[pid] = si.p.x; ps.oy[pid] = si.p.y; ps.oz[pid] = si.p.z;   // mi.spawn_ray(wi): no offset (n = 0, p_error = 0)
                    ps.dx[pid] = wi.x; ps.dy[pid] = wi.y; ps.dz[pid] = wi.z;
                    push_ext = true;
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
        lq_push(s_qext, pid, push_ext); lq_push(s_qres, pid, push_resolve); lq_push(s_qsh, pid, push_shadow); lq_push(s_qmis, pid, push_mis);
        if (finished_bounces >= 0) atomicAdd(&s_hist[finished_bounces > 15 ? 15 : finished_bounces], 1u);
        __syncthreads();
        lq_flush_nosync(s_qext, job.ext_next_count, job.ext_next, 256u, false);
        lq_flush_nosync(s_qres, job.shade_next0_count, job.shade_next0, 256u, false);
        lq_flush_nosync(s_qsh, job.shadow_count, job.shadow, 256u, false);
        lq_flush_nosync(s_qmis, job.mis_count, job.mis, 256u, false);
        __syncthreads();
    }
    lq_flush_nosync(s_qext, job.ext_next_count, job.ext_next, 0u, true);
    lq_flush_nosync(s_qres, job.shade_next0_count, job.shade_next0, 0u, true);
    lq_flush_nosync(s_qsh, job.shadow_count, job.shadow, 0u, true);
    lq_flush_nosync(s_qmis, job.mis_count, job.mis, 0u, true);
    __syncthreads();
    __syncthreads();
    if (threadIdx.x < 16 && s_hist[threadIdx.x]) atomicAdd(&job.counters->path_len[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
    counter_add(&job.counters->stages, n_valid);
    counter_add(&job.counters->shade_items[kMediumClass], n_valid);
    counter_add(&job.counters->shade_bytes[kMediumClass], n_bytes);
    (void)zero_num;
}

// ---- escaped rays and dead paths (class kMissClass) -------------------------------------------------------------------
// More than half of the vertices of an open scene are rays that left it (S2: 54 %). They only need the previous vertex's
// NEE resolved, the environment's Le where path.rs:106-117 adds it, and the path-length histogram entry -- none of the
// BSDF / light-sampling code. Keeping them out of k_shade leaves its waves full of real surface hits.
template <bool SPH, bool VOL>
__global__ __launch_bounds__(256) void k_shade_miss(DeviceScene s, RenderConst rc, PathSoA ps, ShadeJob job) {
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
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < count; qi += gridDim.x * blockDim.x) {
        n_valid++;
        n_bytes += 4 + 4 + 12 + 12 + /* write back */ 12 + 4;
        const uint32_t pid = job.queue[qi];
        const uint32_t meta = ps.meta[pid];
        uint32_t flags = meta >> 24; const uint32_t bounces = (meta >> 16) & 0xffu;
        RGB L(ps.L_r[pid], ps.L_g[pid], ps.L_b[pid]);
        resolve_pending<SPH, VOL>(s, ps, pid, flags, L, zero_num, n_bytes PT_PROF_PASS);
        if (!(flags & PF_DEAD) && (bounces == 0 || (flags & PF_SPECULAR)) && s.n_infinite > 0) {   // path.rs:106-117, ray escaped
            n_bytes += 12 + 12;
            const RGB beta(ps.beta_r[pid], ps.beta_g[pid], ps.beta_b[pid]);
            const V3 rd(ps.dx[pid], ps.dy[pid], ps.dz[pid]);
            for (uint32_t k = 0; k < s.n_infinite; ++k) L = L + light_le(s, s.lights[s.infinite_lights[k]], rd) * beta;
        }
        ps.L_r[pid] = L.r; ps.L_g[pid] = L.g; ps.L_b[pid] = L.b;
        ps.meta[pid] = (meta & 0x00ffffffu) | ((flags & ~PF_CAMERA_RAY) << 24);
        atomicAdd(&s_hist[bounces > 15u ? 15u : bounces], 1u);   // path.rs:219
    }
    __syncthreads();
    if (threadIdx.x < 16 && s_hist[threadIdx.x]) atomicAdd(&job.counters->path_len[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
    counter_add(&job.counters->zero_num, zero_num);
    counter_add(&job.counters->stages, n_valid);
    counter_add(&job.counters->shade_items[kMissClass], n_valid);
    counter_add(&job.counters->shade_bytes[kMissClass], n_bytes);
    (void)rc;
}
template __global__ void k_shade_miss<false, false>(DeviceScene, RenderConst, PathSoA, ShadeJob);
template __global__ void k_shade_miss<true, false>(DeviceScene, RenderConst, PathSoA, ShadeJob);
template __global__ void k_shade_miss<true, true>(DeviceScene, RenderConst, PathSoA, ShadeJob);

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
template __global__ void k_bssrdf<false>(DeviceScene, RenderConst, SobolTables, LightGrid, PathSoA, BssrdfJob);
template __global__ void k_bssrdf<true>(DeviceScene, RenderConst, SobolTables, LightGrid, PathSoA, BssrdfJob);

// ---- film ----------------------------------------------------------------------------------------------------
// One thread per pixel slot; its s_count samples are added in sample order (integrator.rs:331-376),
// FilmTile::add_sample (film.rs:292-331) with the tile's pixel bounds == footprint clipped to the crop window.
__global__ __launch_bounds__(256) void k_film(RenderConst rc, PathSoA ps, const float *filter_table, float *film_rgbw, DevCounters *counters) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nan_c = 0, neg_c = 0, inf_c = 0, splats = 0;
    int32_t px, py;
    if (slot < rc.n_pix_slots && slot_to_pixel(rc, slot, px, py)) {
        // tile pixel bounds (Film::get_film_tile, film.rs:125-140)
        uint32_t tile_slot = slot >> 8;
        uint32_t tile = rc.tile_rank + tile_slot * rc.tile_world;
        int32_t tx0 = rc.sample_bounds[0] + (int32_t)((tile % rc.ntx) * 16u), ty0 = rc.sample_bounds[1] + (int32_t)((tile / rc.ntx) * 16u);
        int32_t tx1 = min(tx0 + 16, rc.sample_bounds[2]), ty1 = min(ty0 + 16, rc.sample_bounds[3]);
        int64_t tb0 = max(f2i_sat(ceilf((float)tx0 - 0.5f - rc.filter_radius[0])), (int64_t)rc.crop[0]);
        int64_t tb1 = max(f2i_sat(ceilf((float)ty0 - 0.5f - rc.filter_radius[1])), (int64_t)rc.crop[1]);
        int64_t tb2 = min(f2i_sat(floorf((float)tx1 - 0.5f + rc.filter_radius[0])) + 1, (int64_t)rc.crop[2]);
        int64_t tb3 = min(f2i_sat(floorf((float)ty1 - 0.5f + rc.filter_radius[1])) + 1, (int64_t)rc.crop[3]);
        const float invrx = 1.0f / rc.filter_radius[0], invry = 1.0f / rc.filter_radius[1];
        // Splats onto this thread's own pixel are accumulated in registers, seeded with the pixel's current value, and written
        // back once: the additions happen in sample order exactly as before (and as FilmTile::add_sample does), without one
        // L2 atomic per channel per sample. Splats onto other pixels (wide filters; for the box filter only the pfilm == px
        // edge case) still use atomics; should one of them land on this pixel meanwhile, the final compare-and-swap fails
        // and the delta is added atomically instead (contribution preserved, order then unspecified as for any such splat).
        const bool own_ok = px >= tb0 && px < tb2 && py >= tb1 && py < tb3;
        float *own = film_rgbw + 4 * ((size_t)(py - rc.crop[1]) * rc.film_w + (size_t)(px - rc.crop[0]));
        float seed[4] = {0.0f, 0.0f, 0.0f, 0.0f}, acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (own_ok) for (int k = 0; k < 4; ++k) { seed[k] = __hip_atomic_load(own + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc[k] = seed[k]; }
        for (uint32_t sl = 0; sl < rc.s_count; ++sl) {
            const uint32_t pid = sl * rc.n_pix_slots + slot;
            RGB L(ps.L_r[pid], ps.L_g[pid], ps.L_b[pid]);
            // integrator.rs:350-368
            if (L.has_nans()) { L = RGB(0.0f); nan_c++; }
            else if (L.y() < -1.0e-5f) { L = RGB(0.0f); neg_c++; }
            else if (__builtin_isinf(L.y())) { L = RGB(0.0f); inf_c++; }
            if (L.y() > rc.max_sample_luminance) L = L * RGB(rc.max_sample_luminance / L.y());
            const float dx = ps.pfilm_x[pid] - 0.5f, dy = ps.pfilm_y[pid] - 0.5f;
            int64_t p0x = max(f2i_sat(ceilf(dx - rc.filter_radius[0])), tb0), p0y = max(f2i_sat(ceilf(dy - rc.filter_radius[1])), tb1);
            int64_t p1x = min(f2i_sat(floorf(dx + rc.filter_radius[0])) + 1, tb2), p1y = min(f2i_sat(floorf(dy + rc.filter_radius[1])) + 1, tb3);
            for (int64_t y = p0y; y < p1y; ++y) {
                const float fy = fabsf(((float)y - dy) * invry * 16.0f);
                const uint32_t iy = min(f2u32_sat(floorf(fy)), 15u);
                for (int64_t x = p0x; x < p1x; ++x) {
                    const float fx = fabsf(((float)x - dx) * invrx * 16.0f);
                    const uint32_t ix = min(f2u32_sat(floorf(fx)), 15u);
                    const float fw = filter_table[iy * 16 + ix];
                    const RGB c = L * RGB(1.0f) * RGB(fw);
                    if (own_ok && x == (int64_t)px && y == (int64_t)py) { acc[0] += c.r; acc[1] += c.g; acc[2] += c.b; acc[3] += fw; }
                    else {
                        float *dst = film_rgbw + 4 * ((size_t)(y - rc.crop[1]) * rc.film_w + (size_t)(x - rc.crop[0]));
                        atomicAdd(dst + 0, c.r); atomicAdd(dst + 1, c.g); atomicAdd(dst + 2, c.b); atomicAdd(dst + 3, fw);
                    }
                    splats++;
                }
            }
        }
        if (own_ok) for (int k = 0; k < 4; ++k) {
            if (__float_as_uint(acc[k]) == __float_as_uint(seed[k])) continue;
            const uint32_t old = atomicCAS((uint32_t *)(own + k), __float_as_uint(seed[k]), __float_as_uint(acc[k]));
            if (old != __float_as_uint(seed[k])) atomicAdd(own + k, acc[k] - seed[k]);
        }
    }
    counter_add(&counters->san_nan, nan_c); counter_add(&counters->san_neg, neg_c);
    counter_add(&counters->san_inf, inf_c); counter_add(&counters->splats, splats);
}

// Film::merge_film_tile (film.rs:142-161): RGB sums -> XYZ, added to the caller's film.
__global__ void k_film_finish(const float *film_rgbw, float *film_xyzw, uint32_t npix) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    float rgb[3] = {film_rgbw[4 * i], film_rgbw[4 * i + 1], film_rgbw[4 * i + 2]}, xyz[3];
    rgb_to_xyz(rgb, xyz);
    film_xyzw[4 * i] += xyz[0]; film_xyzw[4 * i + 1] += xyz[1]; film_xyzw[4 * i + 2] += xyz[2]; film_xyzw[4 * i + 3] += film_rgbw[4 * i + 3];
}

// ---- spatial light distribution (lightdistrib.rs:151-228), all voxels precomputed ---------------------------
__global__ __launch_bounds__(256) void k_light_grid_contrib(DeviceScene s, uint32_t nvx, uint32_t nvy, uint32_t nvz, float *func) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t ncell = (size_t)nvx * nvy * nvz;
    if (gid >= ncell * s.n_lights) return;
    const uint32_t j = (uint32_t)(gid % s.n_lights);
    const size_t cell = gid / s.n_lights;
    const uint32_t pi0 = (uint32_t)(cell % nvx), pi1 = (uint32_t)((cell / nvx) % nvy), pi2 = (uint32_t)(cell / ((size_t)nvx * nvy));
    V3 p0((float)pi0 / (float)nvx, (float)pi1 / (float)nvy, (float)pi2 / (float)nvz);
    V3 p1((float)(pi0 + 1) / (float)nvx, (float)(pi1 + 1) / (float)nvy, (float)(pi2 + 1) / (float)nvz);
    V3 a(lerpf(p0.x, s.wb_min[0], s.wb_max[0]), lerpf(p0.y, s.wb_min[1], s.wb_max[1]), lerpf(p0.z, s.wb_min[2], s.wb_max[2]));
    V3 b(lerpf(p1.x, s.wb_min[0], s.wb_max[0]), lerpf(p1.y, s.wb_min[1], s.wb_max[1]), lerpf(p1.z, s.wb_min[2], s.wb_max[2]));
    V3 vmin(minf(a.x, b.x), minf(a.y, b.y), minf(a.z, b.z)), vmax(maxf(a.x, b.x), maxf(a.y, b.y), maxf(a.z, b.z));
    float contrib = 0.0f;
    for (uint32_t i = 0; i < 128; ++i) {
        V3 u3(radical_inverse(0, i), radical_inverse(1, i), radical_inverse(2, i));
        IData intr;
        intr.p = V3(lerpf(u3.x, vmin.x, vmax.x), lerpf(u3.y, vmin.y, vmax.y), lerpf(u3.z, vmin.z, vmax.z));
        P2 u(radical_inverse(3, i), radical_inverse(4, i));
        float pdf = 0.0f; V3 wi; IData vis;
        RGB Li = light_sample_li<true>(s, j, intr, u, wi, pdf, vis);
        if (pdf > 0.0f) contrib += Li.y() / pdf;
    }
    func[gid] = contrib;
}
// Per voxel: floor at 0.001*avg, then Distribution1D::new (sampling.rs:12-34)
__global__ __launch_bounds__(256) void k_light_grid_finish(uint32_t n_lights, size_t ncell, float *func, float *cdf, float *func_int) {
    const size_t cell = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= ncell) return;
    float *f = func + cell * n_lights, *c = cdf + cell * (n_lights + 1);
    float sum = 0.0f;
    for (uint32_t j = 0; j < n_lights; ++j) sum += f[j];
    const float avg = sum / (128.0f * (float)n_lights);
    const float min_contrib = (avg > 0.0f) ? 0.001f * avg : 1.0f;
    for (uint32_t j = 0; j < n_lights; ++j) f[j] = maxf(f[j], min_contrib);
    c[0] = 0.0f;
    for (uint32_t i = 1; i < n_lights + 1; ++i) c[i] = c[i - 1] + f[i - 1] / (float)n_lights;
    const float fi = c[n_lights];
    if (fi == 0.0f) { for (uint32_t i = 1; i < n_lights + 1; ++i) c[i] = (float)i / (float)n_lights; }
    else { for (uint32_t i = 1; i < n_lights + 1; ++i) c[i] /= fi; }
    func_int[cell] = fi;
}

// ---- parity helpers -------------------------------------------------------------------------------------------
__global__ void k_halton_samples(SobolTables tabs, HaltonParams hp, uint32_t n, const int32_t *pixel_xy, const uint32_t *sample_num,
                                 uint32_t n_dims, float *out, uint64_t *out_index) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t index = halton_index_for_sample(hp, pixel_xy[2 * i], pixel_xy[2 * i + 1], sample_num[i]);
    if (out_index) out_index[i] = index;
    for (uint32_t d = 0; d < n_dims; ++d) out[(size_t)i * n_dims + d] = halton_sample_dimension(tabs, hp, index, d);
}

__global__ void k_sobol_samples(SobolTables tabs, SobolParams sp, uint32_t n, const int32_t *pixel_xy, const uint32_t *sample_num,
                                uint32_t n_dims, float *out, uint64_t *out_index) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t px = pixel_xy[2 * i], py = pixel_xy[2 * i + 1];
    uint64_t index = sobol_interval_to_index(tabs, (uint32_t)sp.log2_resolution, sample_num[i], (uint32_t)(px - sp.sb_min[0]), (uint32_t)(py - sp.sb_min[1]));
    if (out_index) out_index[i] = index;
    for (uint32_t d = 0; d < n_dims; ++d)
        out[(size_t)i * n_dims + d] = (d < 2) ? sobol_pixel_dim(tabs.m32, sp, index, (int)d, d == 0 ? px : py) : sobol_sample_float(tabs.m32, index, d);
}
__global__ void k_camera_rays(RenderConst rc, uint32_t n, const float *cs, float *out_o, float *out_d) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 o, d;
    camera_ray(rc, cs[5 * i], cs[5 * i + 1], cs[5 * i + 2], P2(cs[5 * i + 3], cs[5 * i + 4]), o, d);
    out_o[3 * i] = o.x; out_o[3 * i + 1] = o.y; out_o[3 * i + 2] = o.z;
    out_d[3 * i] = d.x; out_d[3 * i + 1] = d.y; out_d[3 * i + 2] = d.z;
}
