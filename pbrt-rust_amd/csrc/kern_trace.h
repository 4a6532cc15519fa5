// kern_trace.h -- split out of the former single-file kernels.hip so that the translation units compile in parallel.
#pragma once
#include "kern_common.h"
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
// Node steps and leaf (triangle) work share one loop: every iteration a lane fetches ONE record, an interior node's or a leaf packet's
// ("if-if"); lanes at a leaf join once a quorum of them waits. Instance entry / exit is a step of its own (ST_INST / ST_RET).

#ifndef PT_TRACE_WAVES
#define PT_TRACE_WAVES 6        // waves per SIMD of the triangle-only kernels of the exact (two-wide) walk: 80 VGPRs, no scratch (knob history: profiles/HISTORY.md)
#endif
#ifndef PT_TRACE_WAVES_INST
#define PT_TRACE_WAVES_INST 4   // exact walk, triangles + instances (MODE 3)
#endif
#ifndef PT_TRACE_ATTR
#define PT_TRACE_ATTR   // experiment hook, e.g. __attribute__((amdgpu_waves_per_eu(6,6)))
#endif
// MODE: 0 = triangle-only scenes, 1 = general geometry (spheres, instances), 2 = general geometry + alpha-masked triangles
// PROBE (closest hit only): every queue entry is a whole BSSRDF probe chain (TabulatedBSSRDF::sample_sp, bssrdf.rs:367-402). The lane
// walks the chain of Scene::intersect calls itself -- hit, interaction, next segment towards the target -- keeps the last
// kProbeRing matching intersections in a per-lane ring in HBM, and retires with the SELECTED intersection as its hit record
// (`selected = clamp((u1 * nfound) as usize, 0, nfound - 1)`), so that one launch replaces what used to be one wavefront
// iteration (a trace launch, a k_bssrdf launch and a host round trip) per segment, twice over.
// ANY: 0 = closest hit (Scene::intersect), 1 = any hit (Scene::intersect_p), 2 = mixed: the launch walks the queues of job.sub[0..2]
// back to back and every lane carries the kind of its own ray (the continuation, MIS and shadow rays of one wavefront iteration in one
// launch: one tail of straggling rays per iteration instead of three; measured with the PT_TRACE_UTIL build on S2: the wave slots of the three
// separate launches were busy 73 / 53 / 62 % of launch span x resident waves, ~0.8 ms of tail each).
// QUAD: the production traversal -- four-wide records (dev_scene.h: QuadNode), one dependent fetch per two levels of the reference tree, children
// visited in the reference's order, so every hit (primitive, t, barycentrics) and the triangle / sphere test counters are the reference's; the node
// counter then counts RECORDS fetched (128 B each). QUAD = false walks the two-wide records and reproduces the reference's node-visit counter
// (pt_set_trace_exact / PT_TRACE_EXACT=1: the counter tests and the oracle comparisons of bvh_nodes_visited).
template <int ANY, int MODE, bool PROBE, int QUADK>
#ifndef PT_TRACE_WAVES_PROBE
#define PT_TRACE_WAVES_PROBE 1   // experiment hook: waves per SIMD of the triangle-only probe-chain kernel (125 VGPRs = four by itself)
#endif
#ifndef PT_TRACE_WAVES_QUAD
#define PT_TRACE_WAVES_QUAD 5        // waves per SIMD of the four-wide triangle-only kernels (eight quads of a record in flight per lane)
#endif
#ifndef PT_TRACE_WAVES_QUAD_INST
#define PT_TRACE_WAVES_QUAD_INST 4
#endif
__global__ __launch_bounds__(kTraceBlock, (MODE == 0 && !PROBE) ? ((QUADK != 0) ? PT_TRACE_WAVES_QUAD : PT_TRACE_WAVES) : (MODE == 3 && !PROBE) ? ((QUADK != 0) ? PT_TRACE_WAVES_QUAD_INST : PT_TRACE_WAVES_INST) : (MODE == 0 && PROBE) ? PT_TRACE_WAVES_PROBE : 1) PT_TRACE_ATTR void k_trace(DeviceScene s, TraceJob job) {
    static_assert(!(ANY != 0 && PROBE), "probe chains are closest-hit queries");
    // QUADK: 0 = the two-wide (exact) walk, 1 = the production walk, 2 = the production walk of a scene whose records + packets exceed the 4 GB that a buffer load's
    // 32-bit byte offset reaches: the same walk through global loads with 64-bit addresses (~15 more vector instructions per step for the address arithmetic)
    constexpr bool QUAD = QUADK != 0, BIG = QUADK == 2;
    constexpr bool MIX = ANY == 2;
    // MODE: 0 triangles only; 1 general geometry (spheres / disks and object instances); 2 general + alpha-masked triangles;
    //       3 triangles + object instances (no quadrics, no masks: config C4's kind of scene)
    constexpr bool SPH = MODE == 1 || MODE == 2, INST = MODE >= 1, ALPHA = MODE == 2;
    constexpr uint32_t kMaskRef = QUAD ? kRefMaskQuad : kRefMask;   // index bits of a child reference / stack word
    constexpr int kLds = QUAD ? (INST ? kLdsStackQuadInst : kLdsStackQuad) : (MODE == 0 ? kLdsStack : kLdsStackGeneral);   // LDS stack entries per lane
    constexpr int kMaxS = QUAD ? kMaxStackQuad : kMaxStack;   // deepest stack (the four-wide walk pushes up to three entries per record)
    __shared__ uint32_t lds_stack[(kTraceBlock / 64) * kLds * 2 * 64];
#ifndef PT_WRAY_HBM
#define PT_WRAY_HBM 1
#endif
    // the world-space ray of a lane that is inside an instance: six words, written when the instance is entered and read when its marker is popped. The production walk keeps
    // them in the wave's HBM slab behind the spilled stack entries -- the 1.5 KB per wave buy three more LDS stack entries
    constexpr bool kWrayHbm = INST && QUAD && PT_WRAY_HBM != 0;
    __shared__ float lds_wray[(INST && !kWrayHbm) ? (kTraceBlock / 64) * 6 * 64 : 1];
    const uint32_t lane = lane_id();
    const uint32_t wave_in_block = threadIdx.x >> 6;
    // (stack and spill slab through pointers that NAME their address space: a push or pop that picks between them can then only be a branch -- left generic, the
    //  compiler turned `sp < kLds ? LDS : HBM` into flat accesses through a selected address, ~27 instructions per push and a wait on both memory counters per pop;
    //  until round 5 a scratch entry behind the LDS ones took the unconditional write instead, which cost every lane an entry of LDS: each one is worth ~1.4 % of C4's traversal)
    typedef __attribute__((address_space(3))) uint32_t lds_u32; typedef __attribute__((address_space(1))) uint32_t glb_u32;
    lds_u32 *stack = (lds_u32 *)lds_stack + wave_in_block * (kLds * 2 * 64) + lane;   // entry e: words at [2e*64], [(2e+1)*64]
    float *wray_l = lds_wray + ((INST && !kWrayHbm) ? wave_in_block * (6 * 64) + lane : 0u);        // word k at [k*64]
    // spilled entries: [wave][word][lane], so that lanes at the same depth touch consecutive dwords
    glb_u32 *spill = (glb_u32 *)job.spill + (size_t)(blockIdx.x * (kTraceBlock / 64) + wave_in_block) * 64 * kSpillWords + lane;
    typedef __attribute__((address_space(1))) float glb_f32;
    glb_f32 *wray_g = (glb_f32 *)(spill + (2 * kSpillEntries) * 64);   // (the slab is sized for the deeper of the two walks: both index it the same way)
    // MIX: queue entry qi belongs to sub 0 below c0, to sub 1 below c01, to sub 2 otherwise
    const uint32_t c0 = *job.sub[0].count, c01 = c0 + (MIX ? *job.sub[1].count : 0u);
    const uint32_t count = c01 + (MIX ? *job.sub[2].count : 0u);
    uint32_t ksel = 0; bool lane_any = ANY == 1;   // the lane's ray: its sub and whether it is an any-hit query
    __shared__ uint32_t lds_kcnt[MIX ? (kTraceBlock / 64) * 16 : 1];   // MIX: per wave {nodes, tris, rays} x 3 subs
    uint32_t *kcnt = lds_kcnt + (MIX ? (threadIdx.x >> 6) * 16 : 0u);
    if (MIX && lane_id() < 16u) kcnt[lane_id()] = 0u;
    // MIX: what a lane needs of ITS ray kind when it retires / refills -- queue, ray records, output arrays, strides -- sits in an LDS table of three
    // rows (five quads each, row stride 20 words: the three rows' quads fall into different banks) and is fetched with the lane's `ksel` as the row:
    // two ds_read_b128 per refill, three per retire. Chosen per lane out of the kernel arguments it was 26 + 47 selects, and the 60 scalars of the
    // three TraceSubs overflowed the SGPR file (123 v_readlane / 45 v_writelane in the kernel, 69 of them in this block).
    //   quad 0: queue, ray   quad 1: ray_stride, per_ray_tmax | any << 1, scalar_tmax, first queue index of the kind
    //   quad 2: out_hit, out_hit2   quad 3: out_word, out_t   quad 4: out_hit_stride, out_word_stride, out_t_stride, -
    constexpr int kSubRow = 20;
    __shared__ __attribute__((aligned(16))) uint32_t lds_sub[MIX ? 3 * kSubRow : 4];
    if constexpr (MIX) {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const TraceSub &S = job.sub[k];
                uint32_t *r = lds_sub + k * kSubRow;
                const unsigned long long pq = (unsigned long long)S.queue, pr = (unsigned long long)S.ray, ph = (unsigned long long)S.out_hit, ph2 = (unsigned long long)S.out_hit2,
                                         pw = (unsigned long long)S.out_word, pt = (unsigned long long)S.out_t;
                r[0] = (uint32_t)pq; r[1] = (uint32_t)(pq >> 32); r[2] = (uint32_t)pr; r[3] = (uint32_t)(pr >> 32);
                r[4] = S.ray_stride; r[5] = (S.per_ray_tmax ? 1u : 0u) | (S.any ? 2u : 0u); r[6] = __float_as_uint(S.scalar_tmax); r[7] = k == 0 ? 0u : (k == 1 ? c0 : c01);
                r[8] = (uint32_t)ph; r[9] = (uint32_t)(ph >> 32); r[10] = (uint32_t)ph2; r[11] = (uint32_t)(ph2 >> 32);
                r[12] = (uint32_t)pw; r[13] = (uint32_t)(pw >> 32); r[14] = (uint32_t)pt; r[15] = (uint32_t)(pt >> 32);
                r[16] = S.out_hit_stride; r[17] = S.out_word_stride; r[18] = S.out_t_stride; r[19] = 0u;
            }
        }
        __syncthreads();
    }
    const uint4 *const sub_rows = reinterpret_cast<const uint4 *>(lds_sub);
    // (pointers rebuilt from table words are GLOBAL pointers: left generic they become flat loads / stores, which wait on both memory counters)
#define PT_GPTR(T, lo, hi) ((__attribute__((address_space(1))) T *)(((unsigned long long)(hi) << 32) | (unsigned long long)(lo)))
#define PT_SUB(f) (job.sub[0].f)   // the launches of ONE ray kind (the mixed launch reads its lane's row of lds_sub instead)
    const uint4 *wide4 = reinterpret_cast<const uint4 *>(s.wide);
    // QUAD: records and packets live in ONE allocation, addressed by 32-bit byte offsets from its start through a buffer resource
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<QuadNode *>(s.quad), 0, BIG ? 0 : (int)(s.pool_quads << 4), 0x00020000);
    typedef __attribute__((address_space(1))) const v4u gpool_u32;   // BIG: the pool as global quads
    gpool_u32 *const gpool = (gpool_u32 *)s.quad;
    const uint4 *leaf4 = reinterpret_cast<const uint4 *>(s.leaf);
    uint32_t n_nodes = 0, n_tris = 0, n_rays = 0, n_sph = 0;
#ifdef PT_TRACE_UTIL   // SIMD utilisation study: wave iterations and active lanes of the node phase / the leaf phase
    uint32_t u_it1 = 0, u_act1 = 0, u_it2 = 0, u_act2 = 0, u_it3 = 0, u_act3 = 0;
    uint32_t u_ent = 0, u_rej = 0, u_ihit = 0, u_spill = 0;   // instance entries tried / turned away by the object's root test / left with a hit; stack entries written beyond the LDS ones
    unsigned long long u_cxf = 0, u_cmain = 0;   // wave cycles inside the transform step / the record step
    unsigned long long u_cfetch = 0, u_cnode = 0, u_cleaf = 0, u_cpop = 0;   // of the record step: issue + wait of the loads, the node branch, the leaf branch, the pops
    long long u_cm = 0;
#define PT_UTIL_MARK(acc) do { const long long n_ = clock64(); acc += (unsigned long long)(n_ - u_cm); u_cm = n_; } while (0)
    const unsigned long long u_t0 = wall_clock64(); const long long u_t0c = clock64();
#define PT_UTIL(it, act, pred) do { const unsigned long long m_ = __ballot(pred); if (pred) { act++; it += (lane == (uint32_t)(__ffsll((long long)m_) - 1)); } } while (0)
#else
#define PT_UTIL(it, act, pred) do { } while (0)
#define PT_UTIL_MARK(acc) do { } while (0)
#endif

    // lane state: ST_IDLE (no ray), ST_ENTER (fetch record `cur`), ST_LEAF (test packets from `cur`), ST_DONE
    // ST_INST (an instance packet at `cur` waits to be entered) and ST_RET (the instance's marker was popped, the world ray waits to
    // be restored) are the two transform steps of TransformedPrimitive (primitive.rs:58-88): ~200 instructions that only a few lanes
    // need in any one iteration, so they run in a step of their own once `inst_quorum` lanes wait for it (like leaves, below)
    enum : uint32_t { ST_IDLE = 0, ST_ENTER = 1, ST_LEAF = 2, ST_DONE = 3, ST_INST = 4, ST_RET = 5, ST_LEAFS = 6 };   // ST_LEAFS: at a leaf and being served (leaf_quorum below)
    uint32_t state = ST_IDLE;
#ifndef PT_TRACE_CHUNK
#define PT_TRACE_CHUNK 512   // queue entries a wave reserves per atomic (sweeps: profiles/HISTORY.md)
#endif
    constexpr int kChunk = PT_TRACE_CHUNK;
    // Scenes with instances: the bites shrink near the end of the queue (guided self-scheduling). Their rays are long (S4: 50 records and 20 triangle tests on average, some
    // ten times that), so a wave that bites 512 entries when nothing is left behind them works them off while the chip idles. What is left is known for free: the atomic
    // returns the old head. Within the last `PT_TRACE_TAIL_ROUNDS` rounds of the grid the bite is a quarter, within the last quarter round a sixteenth (C4 trace 932 -> 915 ms).
    // Triangle-only scenes keep the full bite to the end: their launches end on single long rays, not on bites (C2: 131.4 -> 133.5 ms with the small ones, the eighth
    // of the job that rank 0 of 8 renders 22.1 -> 22.7 ms; a plain look at the head before every bite cost more than the tail it saved, round 2).
#ifndef PT_TRACE_TAIL_ROUNDS
#define PT_TRACE_TAIL_ROUNDS 2
#endif
    constexpr bool kTailBites = MODE != 0 && PT_TRACE_TAIL_ROUNDS > 0;
    uint32_t bite = kChunk;   // wave-uniform (only the kernels with shrinking bites ever change it)
    bool exhausted = false;
    uint32_t chunk_next = 0, chunk_left = 0;   // wave-uniform
    uint32_t qwin = 0u, win_base = 0xffffffffu;   // the prefetched queue window (per lane) and the queue index it starts at (wave-uniform); see the refill block
    uint32_t pid = 0, cur = 0, sp = 0, pending = 0;
    V3 ro, rd, inv_dir;
    TriRay tray; tray.kz = 2; tray.Sx = tray.Sy = tray.Sz = 0.0f;   // per-ray half of the triangle test
    V3 rop;              // the ray origin permuted like the packet quads the lane loads (dev_scene.h: TriPacket)
    uint32_t lofs = 0;   // byte offsets of a packet's quads in the ray's order (kx, ky, kz): 16 kx | 16 ky << 8 | 16 kz << 16
#define PT_TRI_RAY() do { tray = tri_ray_setup(rd); rop = tri_permute(ro, tray.kz); const uint32_t kz_ = (uint32_t)tray.kz, kx_ = kz_ == 2u ? 0u : kz_ + 1u, ky_ = kx_ == 2u ? 0u : kx_ + 1u; lofs = (kx_ << 4) | (ky_ << 12) | (kz_ << 20); } while (0)
    bool nx = false, ny = false, nz = false;
    uint32_t sgn3 = 0;   // QUAD: 3 x the ray's sign octant (nx | ny << 1 | nz << 2): the shift that finds the octant's slot order in a record's order word
#define PT_SGN3() do { if (QUAD) sgn3 = 3u * ((nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u)); } while (0)
    float t_max = 0.0f;
    // closest hit so far: its packet (index in DeviceScene::leaf, PT_NONE = no hit yet) -- the primitive id and the flag word are
    // read back from that packet when the ray retires instead of being carried through the loop
    uint32_t hit_pkt = PT_NONE; float hit_t = 0.0f, hb0 = 0.0f, hb1 = 0.0f, hb2 = 0.0f;
    // instancing (primitive.rs:58-88): while inside an instance the lane's ray is the object-space ray
    uint32_t in_inst = PT_NONE, hit_inst = PT_NONE; float t_max_world = 0.0f; bool inst_hit = false;
    uint32_t xf_arg = 0;   // ST_INST: instance index | TP_LAST of its packet in bit 31; ST_RET: word 0 of the popped marker entry
    constexpr uint32_t kMarker = 0xFFC0DEADu;   // stack word 1 of an "end of instance" entry (never a real tmin)
    // PROBE: chain state of the lane. A chain with more than kProbeRing matches after the selected one is walked a second time
    // ("rewalk") up to match number `selected`; the rewalk's work is not counted (the reference indexes its Vec instead).
    uint32_t nfound = 0, seen = 0; bool rewalk = false;
    // the chain's target, the BSSRDF's material and the selection number u1: read ONCE, with the ray record at the refill (one 32-byte probe record, kernels.h: BssSoA), and kept
    // in registers across the chain's segments (until round 6 five dword gathers at every retired segment; the kernel runs three waves per SIMD with registers to spare)
    V3 pr_target(0.0f, 0.0f, 0.0f); uint32_t pr_mat = PT_NONE; float pr_u1n = 0.0f;
    // the medium the chain's current probe ray travels in (volpath only reads it): every hit's MediumInterface is the primitive's own when it is a
    // transition, else this medium on both sides (primitive.rs:139-145); the next probe ray takes its medium from that interface (interaction.rs:38-43);
    // the first ray starts from an interaction without one (bssrdf.rs:362-366). Packed as inside | outside << 16, 0xffff = none.
    uint32_t cur_med = PT_NONE;
    uint32_t sv_nodes = 0, sv_tris = 0, sv_rays = 0, sv_sph = 0;
    uint4 *ring = PROBE ? job.ring + (size_t)(blockIdx.x * (kTraceBlock / 64) + wave_in_block) * (kProbeRing * 3 * 64) + lane : nullptr;

    // Pop entries until one passes its deferred `tmin < t_max` test (or the stack is empty).
    auto pop_next = [&]() {
        for (;;) {
            if (!QUAD) { n_nodes += pending; pending = 0; }   // skipped far children above the top entry: popped + failed
            if (sp == 0) { state = ST_DONE; return; }
            sp--;
            // (the LDS entry is read unconditionally and the rare spilled entry replaces it: written as `sp < kLds ? LDS : HBM` the two became ONE pair
            //  of flat loads through a selected address -- every pop went down the vector-memory path and waited on both counters)
            const uint32_t se = min(sp, (uint32_t)(kLds - 1));
            uint32_t w0 = stack[(2 * se) * 64], w1 = stack[(2 * se + 1) * 64];
            asm volatile("" : "+v"(w0), "+v"(w1));   // keeps the two loads apart
            if (sp >= (uint32_t)kLds) { w0 = spill[(2 * (sp - kLds)) * 64]; w1 = spill[(2 * (sp - kLds) + 1) * 64]; }
            if (INST && w1 == kMarker) {               // the object's BVH is exhausted: back to world space (primitive.rs:70-77), in the transform step
                xf_arg = w0; state = ST_RET;
                return;
            }
            if (!QUAD) { n_nodes++; pending = (w0 >> 25) & 63u; }   // the reference tests the popped node now
            if (__uint_as_float(w1) < t_max) {         // deferred half of intersect_p2
                cur = w0 & kMaskRef;
                state = (w0 & kLeafBit) ? ST_LEAF : ST_ENTER;
                return;
            }
        }
    };

    auto push = [&](uint32_t w0, uint32_t w1) {
        if (sp < (uint32_t)kLds) { stack[(2 * sp) * 64] = w0; stack[(2 * sp + 1) * 64] = w1; }
        else {
            spill[(2 * (sp - kLds)) * 64] = w0; spill[(2 * (sp - kLds) + 1) * 64] = w1;
#ifdef PT_TRACE_UTIL
            u_spill++;
#endif
        }
        sp++;
    };
    const uint32_t root_ref = QUAD ? s.root_ref4 : s.root_ref;

    for (;;) {
        // ---- retire finished rays and refill their lanes, in batches: finished lanes wait (idle) until at least
        //      `refill_min` of them have accumulated or nothing else is running, so the queue atomics below are
        //      paid once per batch instead of once per ray.
        const unsigned long long donem = __ballot(state == ST_DONE || state == ST_IDLE);
        const unsigned long long busy = __ballot(state == ST_ENTER || state == ST_LEAF || state == ST_LEAFS || state == ST_INST || state == ST_RET);
        if (donem != 0ull && ((uint32_t)__popcll(donem) >= job.refill_min || busy == 0ull)) {
            bool retire = state == ST_DONE;
            if constexpr (PROBE) {
                if (retire) {   // one step of the chain loop of bssrdf.rs:373-395, in world space (every instance marker is popped)
                    const V3 target = pr_target;
                    bool finish = false, next_seg = false, chain_end = false;
                    if (hit_pkt != PT_NONE) {
                        SurfaceInteraction si;
                        const uint32_t pfl = fill_hit_pkt<INST>(s, hit_pkt, INST ? hit_inst : PT_NONE, ro, rd, hb0, hb1, hb2, si);
                        const uint32_t hprim = s.leaf[hit_pkt].prim;
                        const MedIface hif = surface_iface(s, hprim, cur_med);
                        const uint32_t hif_packed = (hif.inside & 0xffffu) | (hif.outside << 16);
                        if (packet_material(s, pfl, hprim) == pr_mat) {   // Arc::ptr_eq(material), bssrdf.rs:385-391
                            if (!rewalk) {
                                const uint32_t k = nfound % (uint32_t)kProbeRing;
                                ring[(3 * k + 0) * 64] = make_uint4(hit_pkt, hit_inst, __float_as_uint(hb0), __float_as_uint(hb1));
                                ring[(3 * k + 1) * 64] = make_uint4(__float_as_uint(hb2), __float_as_uint(ro.x), __float_as_uint(ro.y), __float_as_uint(ro.z));
                                ring[(3 * k + 2) * 64] = make_uint4(__float_as_uint(rd.x), __float_as_uint(rd.y), __float_as_uint(rd.z), hif_packed);
                                if (nfound == 0xffffffffu) atomicMax(job.error, (uint32_t)PT_ERR_PROBE_CHAIN); else nfound++;
                            } else {
                                const uint32_t selected = min(f2u32_sat(pr_u1n * (float)nfound), nfound - 1u);
                                if (seen == selected) { finish = true; job.bs.iface(pid) = hif_packed; }
                                seen++;
                            }
                        }
                        if (!finish) {   // base = si.get_data(); the next segment runs base -> target (interaction.rs:38-43)
                            const V3 d = target - si.p;
                            if (d.x == 0.0f && d.y == 0.0f && d.z == 0.0f) chain_end = true;
                            else { ro = offset_ray_origin(si.p, si.p_error, si.n, d); rd = d; next_seg = true; cur_med = medium_toward(hif, si.n, d); }
                        }
                    } else chain_end = true;
                    if (chain_end) {
                        if (!rewalk && nfound > 0u) {
                            // bssrdf.rs:398: selected = clamp((u1 * nfound) as usize, 0, nfound - 1)
                            const uint32_t selected = min(f2u32_sat(pr_u1n * (float)nfound), nfound - 1u);
                            if (nfound - selected <= (uint32_t)kProbeRing) {
                                const uint32_t k = selected % (uint32_t)kProbeRing;
                                const uint4 e0 = ring[(3 * k + 0) * 64], e1 = ring[(3 * k + 1) * 64], e2 = ring[(3 * k + 2) * 64];
                                hit_pkt = e0.x; hit_inst = e0.y; hb0 = __uint_as_float(e0.z); hb1 = __uint_as_float(e0.w); hb2 = __uint_as_float(e1.x);
                                ro = V3(__uint_as_float(e1.y), __uint_as_float(e1.z), __uint_as_float(e1.w));
                                rd = V3(__uint_as_float(e2.x), __uint_as_float(e2.y), __uint_as_float(e2.z));
                                job.bs.iface(pid) = e2.w;
                                finish = true;
                            } else {   // the selected intersection has left the ring: walk again from the start, uncounted
                                rewalk = true; seen = 0u; cur_med = PT_NONE;
                                sv_nodes = n_nodes; sv_tris = n_tris; sv_rays = n_rays; sv_sph = n_sph;
                                { const float4 st = job.bs.probe[(size_t)pid * BssSoA::kProbeQuads]; ro = V3(st.x, st.y, st.z); } rd = target - ro;
                                next_seg = true;
                            }
                        } else { hit_pkt = PT_NONE; finish = true; }   // nfound == 0: S = 0 (bssrdf.rs:397); a rewalk cannot end before `selected`
                    }
                    if (next_seg || finish) {   // the segment's ray goes back to the path state: k_bssrdf rebuilds the exit point from it
                        float4 *rw = const_cast<float4 *>(job.sub[0].ray) + (size_t)pid * job.sub[0].ray_stride;
                        rw[0] = make_float4(ro.x, ro.y, ro.z, rd.x); rw[1] = make_float4(rd.y, rd.z, 0.0f, 0.0f);
                    }
                    if (next_seg) {
                        retire = false;
                        t_max = job.sub[0].scalar_tmax;
                        inv_dir = V3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        nx = inv_dir.x < 0.0f; ny = inv_dir.y < 0.0f; nz = inv_dir.z < 0.0f; PT_SGN3();
                        PT_TRI_RAY();
                        sp = 0; pending = 0;
                        hit_pkt = PT_NONE; hit_t = 0.0f; hb0 = hb1 = hb2 = 0.0f;
                        in_inst = PT_NONE; hit_inst = PT_NONE; inst_hit = false;
                        n_rays++;                                  // Scene::intersect of the next segment
                        state = ST_DONE;                           // (a segment that misses the world bound is a miss: resolved at the next refill round)
                        if (s.n_nodes > 0) {
                            if (QUAD && !(root_ref & kLeafBit)) { cur = root_ref & kMaskRef; state = ST_ENTER; }   // (as at a refill: no root test in the production walk)
                            else {
                                n_nodes++;
                                if (slab_test(s.root_min, s.root_max, ro, inv_dir, nx, ny, nz, t_max)) {
                                    cur = root_ref & kMaskRef;
                                    state = (root_ref & kLeafBit) ? ST_LEAF : ST_ENTER;
                                }
                            }
                        }
                    } else {
                        if (rewalk) { n_nodes = sv_nodes; n_tris = sv_tris; n_rays = sv_rays; n_sph = sv_sph; }
                        job.bs.nfound(pid) = nfound;
                    }
                }
            }
            // The block is ordered for ONE memory round trip: (A) the retiring lanes request the two words of their hit's packet, (B) the lanes to be refilled take
            // their path ids out of the wave's prefetched queue window and request their ray records, (C) the retiring lanes store their hit records, (D) the new rays
            // are unpacked, (E) the window behind the entries just taken is requested for the next refill. Until round 4's second half the wave stood still for three
            // dependent round trips here (packet words -> stores; queue entry -> ray record), once per ~4 record steps.
            typedef float v4f_ __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(1))) v4f_ gf4; typedef __attribute__((address_space(1))) uint32_t gu32; typedef __attribute__((address_space(1))) float gf32;
            typedef __attribute__((address_space(1))) const uint32_t gcu32; typedef __attribute__((address_space(1))) const v4f_ gcf4;
            // (A)
            uint32_t hit_prim = PT_NONE, hit_fl = (uint32_t)kMissClass << kTpClassShift;
            if (retire && !lane_any && hit_pkt != PT_NONE) { hit_prim = s.leaf[hit_pkt].prim; hit_fl = s.leaf[hit_pkt].flags; }
            const uint32_t r_pid = pid, r_ksel = ksel, r_pkt = hit_pkt, r_inst = hit_inst, r_nn = n_nodes, r_nt = n_tris; const bool r_any = lane_any;
            const float r_b0 = hb0, r_b1 = hb1, r_b2 = hb2;
            // (triangle-only scenes: the closest hit's t IS the ray's t_max -- one register less in the loop; inside instances t_max is an object-space value)
            const float r_t = (INST || PROBE) ? hit_t : (hit_pkt != PT_NONE ? t_max : 0.0f);
            if (retire) state = ST_IDLE;
            // (B)
            bool get = false; v4f_ rr0 = {0.0f, 0.0f, 0.0f, 0.0f}, rr1 = rr0, pb0 = rr0, pb1 = rr0; bool per_ray_tmax = PT_SUB(per_ray_tmax) != 0u; float scalar_tmax = PT_SUB(scalar_tmax);
            // the wave's queue window: lane l holds the path id of queue entry win_base + l
            auto load_window = [&](uint32_t base, uint32_t left) {
                const uint32_t qi = base + lane; uint32_t w = 0u;
                if (lane < left) {
                    uint32_t qk = qi; gcu32 *qp = (gcu32 *)PT_SUB(queue);
                    if constexpr (MIX) {
                        const uint32_t kk = (qi >= c0 ? 1u : 0u) + (qi >= c01 ? 1u : 0u);
                        const uint4 r0 = sub_rows[5u * kk], r1 = sub_rows[5u * kk + 1u];
                        qp = PT_GPTR(const uint32_t, r0.x, r0.y); qk = qi - r1.w;
                    }
                    w = qp ? qp[qk] : qk;   // (the loaded word as it is: anything computed from it here would make the wave wait for the prefetch)
                }
                return w;
            };
            if (!exhausted) {
                // work fetch: the wave reserves kChunk consecutive queue entries with one atomic and hands them
                // out over several refills (consecutive entries are spatially coherent rays)
                if (chunk_left == 0) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(job.head, kTailBites ? bite : (uint32_t)kChunk);
                    chunk_next = __shfl(base, 0);
                    chunk_left = (chunk_next < count) ? min(kTailBites ? bite : (uint32_t)kChunk, count - chunk_next) : 0u;
                    if constexpr (kTailBites) {
                        const uint32_t grid_round = gridDim.x * (kTraceBlock / 64) * (uint32_t)kChunk;   // entries one full bite of every wave takes
                        const uint32_t behind = count > chunk_next + bite ? count - chunk_next - bite : 0u;   // entries nobody had taken when this bite was
                        bite = behind < grid_round / 4u ? (uint32_t)kChunk / 16u : behind < (uint32_t)PT_TRACE_TAIL_ROUNDS * grid_round ? (uint32_t)kChunk / 4u : (uint32_t)kChunk;
                    }
                    if (chunk_left == 0) exhausted = true;
                }
                if (chunk_left != 0u && win_base != chunk_next) { qwin = load_window(chunk_next, chunk_left); win_base = chunk_next; }   // a fresh chunk: not prefetched (once per kChunk rays)
                // PROBE: lanes that went on to their chain's next segment are in `donem` but not idle any more
                const unsigned long long idlem = PROBE ? __ballot(state == ST_IDLE) : donem;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idlem >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idlem, 0u));   // set bits below this lane (v_mbcnt: no per-lane mask registers)
                const uint32_t take = min(chunk_left, (uint32_t)__popcll(idlem));
                const uint32_t wsel = (uint32_t)__shfl((int)qwin, (int)(rank & 63u));   // entry chunk_next + rank (the window starts at chunk_next)
                const uint32_t qi = chunk_next + rank;
                get = state == ST_IDLE && rank < take;
                chunk_next += take; chunk_left -= take;
                if (get) {
                    gcf4 *rays = (gcf4 *)PT_SUB(ray); uint32_t ray_stride = PT_SUB(ray_stride);
                    pid = wsel;
                    if constexpr (MIX) {
                        ksel = (qi >= c0 ? 1u : 0u) + (qi >= c01 ? 1u : 0u);
                        const uint4 r0 = sub_rows[5u * ksel], r1 = sub_rows[5u * ksel + 1u];
                        rays = PT_GPTR(const v4f_, r0.z, r0.w);
                        ray_stride = r1.x; per_ray_tmax = (r1.y & 1u) != 0u; lane_any = (r1.y & 2u) != 0u; scalar_tmax = __uint_as_float(r1.z);
                    }
                    gcf4 *const rp = rays + (size_t)pid * ray_stride;
                    rr0 = rp[0]; rr1 = rp[1];   // one 32-byte record
                    if constexpr (PROBE) { gcf4 *const pp = (gcf4 *)job.bs.probe + (size_t)pid * BssSoA::kProbeQuads; pb0 = pp[0]; pb1 = pp[1]; }   // ... and the chain's 32-byte probe record with it
                }
            }
            // (every lane consumes what (A) and (B) requested, here, once: consumed only inside the conditional blocks below, the loads stay "possibly pending" on the
            //  paths around those blocks as far as the compiler's wait-count bookkeeping can tell, and it then waits for EVERYTHING -- the prefetch of (E) included --
            //  at the first reuse of one of their registers in the record step)
            asm volatile("" :: "v"(hit_prim), "v"(hit_fl), "v"(rr0), "v"(rr1));
            if constexpr (PROBE) asm volatile("" :: "v"(pb0), "v"(pb1));
            // (C)
            if (retire) {
                gf4 *o_hit = (gf4 *)PT_SUB(out_hit), *o_hit2 = (gf4 *)PT_SUB(out_hit2); gu32 *o_word = (gu32 *)PT_SUB(out_word); gf32 *o_t = (gf32 *)PT_SUB(out_t);
                uint32_t o_hit_stride = PT_SUB(out_hit_stride), o_word_stride = PT_SUB(out_word_stride), o_t_stride = PT_SUB(out_t_stride);
                if constexpr (MIX) {
                    const uint4 r2 = sub_rows[5u * r_ksel + 2u], r3 = sub_rows[5u * r_ksel + 3u], r4 = sub_rows[5u * r_ksel + 4u];
                    o_hit = PT_GPTR(v4f_, r2.x, r2.y); o_hit2 = PT_GPTR(v4f_, r2.z, r2.w); o_word = PT_GPTR(uint32_t, r3.x, r3.y); o_t = PT_GPTR(float, r3.z, r3.w);
                    o_hit_stride = r4.x; o_word_stride = r4.y; o_t_stride = r4.z;
                }
                if (r_any) o_word[(size_t)r_pid * o_word_stride] = r_pkt != PT_NONE ? 1u : 0u;   // (an any-hit ray records the packet that stopped it)
                else {
                    if (o_hit) o_hit[(size_t)r_pid * o_hit_stride] = v4f_{__uint_as_float(hit_prim), r_b0, r_b1, r_b2};   // one quad
                    else o_word[(size_t)r_pid * o_word_stride] = hit_prim;   // only the hit / miss matters (volpath shadow rays)
                    if (o_hit2) o_hit2[(size_t)r_pid * o_hit_stride] = v4f_{__uint_as_float(INST ? r_inst : PT_NONE), r_t, __uint_as_float(r_pkt), __uint_as_float(hit_fl)};
                    if (o_t) o_t[(size_t)r_pid * o_t_stride] = r_t;
                }
                if constexpr (MIX) {   // per-kind work counters (LDS atomics of the wave's own slots)
                    atomicAdd(&kcnt[3u * r_ksel], r_nn); atomicAdd(&kcnt[3u * r_ksel + 1u], r_nt); atomicAdd(&kcnt[3u * r_ksel + 2u], 1u);   // (MIX: n_nodes / n_tris count the lane's current ray only)
                }
            }
            // (D)
            if (get) {
                if constexpr (MIX) { n_nodes = 0u; n_tris = 0u; }
                ro = V3(rr0.x, rr0.y, rr0.z);
                rd = V3(rr0.w, rr1.x, rr1.y);
                t_max = per_ray_tmax ? rr1.z : scalar_tmax;
                inv_dir = V3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                nx = inv_dir.x < 0.0f; ny = inv_dir.y < 0.0f; nz = inv_dir.z < 0.0f; PT_SGN3();
                PT_TRI_RAY();
                sp = 0; pending = 0;
                hit_pkt = PT_NONE; hit_t = 0.0f; hb0 = hb1 = hb2 = 0.0f;
                in_inst = PT_NONE; hit_inst = PT_NONE; inst_hit = false;
                if (PROBE) { nfound = 0u; seen = 0u; rewalk = false; cur_med = PT_NONE; pr_u1n = pb0.w; pr_target = V3(pb1.x, pb1.y, pb1.z); pr_mat = __float_as_uint(pb1.w); }
                n_rays++;
                state = ST_DONE;
                if (s.n_nodes > 0) {
                    if (QUAD && !(root_ref & kLeafBit)) {
                        // The production walk enters an interior root without the root's own slab test (bvh.rs:725-727): a ray that misses the root's box misses
                        // the boxes of the root record's four slots (they lie inside it and the slab arithmetic is monotone in the planes -- the argument that
                        // lets the walk skip the collapsed L and R), so it reaches no leaf either way; ~50 vector instructions per ray that only rays starting
                        // outside the scene's bounds and pointing away ever needed. (The exact walk counts the test as the reference does.)
                        cur = root_ref & kMaskRef; state = ST_ENTER;
                    } else {   // the root node's own test (bvh.rs:725-727)
                        if (!QUAD) n_nodes++;   // (production walk: the counter counts four-wide RECORDS fetched, nothing else)
                        if (slab_test(s.root_min, s.root_max, ro, inv_dir, nx, ny, nz, t_max)) {
                            cur = root_ref & kMaskRef;
                            state = (root_ref & kLeafBit) ? ST_LEAF : ST_ENTER;
                        }
                    }
                }
            }
            // (E) (a coalesced 256-byte read per refill; it has the record steps until the next refill to arrive)
            if (!exhausted) { if (chunk_left != 0u) { qwin = load_window(chunk_next, chunk_left); win_base = chunk_next; } else win_base = 0xffffffffu; }   // (a drained chunk: the next one may start right where this one ended)
        }
        if (__ballot(state != ST_IDLE) == 0ull) break;   // queue drained and every lane retired

        // ---- one record per lane and iteration: a lane at an interior node fetches its 64-byte two-wide record, a lane at a leaf
        //      its next 48-byte packet; both kinds of fetch are in flight together and nobody waits for a phase change
        // lanes at a leaf join in once `leaf_quorum` of them wait (or no lane is at a node), so the triangle test is not
        // executed for a handful of lanes in every iteration
        const bool at_node = state == ST_ENTER;
        const unsigned long long leaf_m = __ballot(state == ST_LEAF);   // lanes waiting at a leaf
        // (the ballots are taken by the whole wave BEFORE the per-lane test: inside `state == ST_LEAF && (.. || __ballot(at_node) == 0)` the
        //  short-circuit ran the ballot under the leaf lanes' exec mask only, where it is always 0 -- the quorum never held anything back;
        //  found in round 2 with the PT_TRACE_UTIL counters: 88 % of all wave iterations ran the triangle test for 5 lanes)
        const bool no_node_lane = __ballot(at_node) == 0ull;
        // The quorum is sticky: the lanes that formed it (ST_LEAFS) are served until each has left its leaf (a leaf holds up to
        // max_node_prims packets, one per iteration -- without the hysteresis a lane in a three-packet leaf waits for three quorums); lanes
        // that arrive meanwhile wait for the next quorum (in S4 some lane is at a leaf in 99 % of the iterations: "until none is left"
        // would never end).
        if (__ballot(state == ST_LEAFS) == 0ull && ((uint32_t)__popcll(leaf_m) >= job.leaf_quorum || (no_node_lane && leaf_m != 0ull))) { if (state == ST_LEAF) state = ST_LEAFS; }
        const bool at_leaf = state == ST_LEAFS;
        if constexpr (INST) {
            // ---- transform step: enter instances / return from them, once enough lanes wait (or nothing else can run)
            const unsigned long long xf_m = __ballot(state == ST_INST || state == ST_RET);
            if (xf_m != 0ull && ((uint32_t)__popcll(xf_m) >= job.inst_quorum || __ballot(at_node || at_leaf) == 0ull)) {
                bool need_pop = false;
#ifdef PT_TRACE_UTIL
                PT_UTIL(u_it3, u_act3, state == ST_INST || state == ST_RET);
                const long long u_c0 = clock64();
#endif
                if (state == ST_RET) {
                    const uint32_t w0 = xf_arg;
                    if (!QUAD) pending = (w0 >> 25) & 63u;            // the outer traversal's skipped entries
                    if constexpr (kWrayHbm) { ro = V3(wray_g[0], wray_g[64], wray_g[128]); rd = V3(wray_g[192], wray_g[256], wray_g[320]); }
                    else { ro = V3(wray_l[0], wray_l[64], wray_l[128]); rd = V3(wray_l[192], wray_l[256], wray_l[320]); }
                    inv_dir = V3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                    nx = inv_dir.x < 0.0f; ny = inv_dir.y < 0.0f; nz = inv_dir.z < 0.0f; PT_SGN3();
                    PT_TRI_RAY();
#ifdef PT_TRACE_UTIL
                    if (inst_hit) u_ihit++;
#endif
                    t_max = inst_hit ? t_max : t_max_world;  // r.t_max = ray.t_max only when the instance was hit
                    in_inst = PT_NONE; inst_hit = false;
                    if (w0 & kLeafBit) { cur = w0 & kMaskRef; state = ST_LEAF; }  // remaining packets of the outer leaf
                    else need_pop = true;
                } else if (state == ST_INST) {   // TransformedPrimitive::intersect / intersect_p (primitive.rs:58-88)
                    const uint32_t li = cur;
                    const bool more = !(xf_arg & 0x80000000u);
                    const uint32_t ii = xf_arg & 0x7fffffffu;
                    const DevInstance &I = s.instances[ii];
                    // ray = inverse(prim_to_world).transform_ray(r)  (transform.rs:543-577, t_max -= dt)
                    const M4 w2i = ldm4g(I.world_to_instance);
                    V3 oerr; V3 o2 = xf_point_err(w2i, ro, oerr); const V3 d2 = xf_vector(w2i, rd);
                    const float l2 = length_squared(d2);
                    float tm2 = t_max;
                    if (l2 > 0.0f) { const float dt = dot(vabs(d2), oerr) / l2; o2 = o2 + d2 * dt; tm2 -= dt; }
                    const V3 inv2(1.0f / d2.x, 1.0f / d2.y, 1.0f / d2.z);
                    const bool nx2 = inv2.x < 0.0f, ny2 = inv2.y < 0.0f, nz2 = inv2.z < 0.0f;
                    bool enter = true;
                    if (!I.single) { if (!QUAD) n_nodes++; enter = slab_test(I.root_min, I.root_max, o2, inv2, nx2, ny2, nz2, tm2); }  // object BVH root (bvh.rs:725-727)
#ifdef PT_TRACE_UTIL
                    u_ent++; if (!enter) u_rej++;
#endif
                    if (enter && ((!QUAD && pending > 63u) || sp >= (uint32_t)kMaxS)) { atomicMax(job.error, (uint32_t)PT_ERR_STACK_OVERFLOW); enter = false; }
                    if (enter) {
                        // remember where to resume: the rest of this leaf (if any) and the outer skip count
                        push((more ? (kLeafBit | ((li + 1u) & kMaskRef)) : 0u) | (QUAD ? 0u : (pending << 25)), kMarker);
                        pending = 0;
                        if constexpr (kWrayHbm) { wray_g[0] = ro.x; wray_g[64] = ro.y; wray_g[128] = ro.z; wray_g[192] = rd.x; wray_g[256] = rd.y; wray_g[320] = rd.z; }
                        else { wray_l[0] = ro.x; wray_l[64] = ro.y; wray_l[128] = ro.z; wray_l[192] = rd.x; wray_l[256] = rd.y; wray_l[320] = rd.z; }
                        t_max_world = t_max; in_inst = ii; inst_hit = false;
                        ro = o2; rd = d2; inv_dir = inv2; nx = nx2; ny = ny2; nz = nz2; PT_SGN3(); t_max = tm2;
                        PT_TRI_RAY();
                        const uint32_t iroot = QUAD ? I.root_ref4 : I.root_ref;
                        cur = iroot & kMaskRef;
                        state = (iroot & kLeafBit) ? ST_LEAF : ST_ENTER;
                    } else if (more) { cur = li + 1u; state = ST_LEAF; }
                    else need_pop = true;
                }
                if (need_pop) pop_next();
#ifdef PT_TRACE_UTIL
                u_cxf += (unsigned long long)(clock64() - u_c0);
#endif
                continue;   // states changed: re-evaluate which step runs next
            }
        }
        PT_UTIL(u_it1, u_act1, at_node || at_leaf);
        PT_UTIL(u_it2, u_act2, at_leaf);
#ifdef PT_TRACE_UTIL
        const long long u_c1 = clock64(); u_cm = u_c1;
#endif
        if (at_node || at_leaf) {
            uint4 q0, q1, q2, q3;
            bool need_pop = false;
            if constexpr (QUAD) {
                // Buffer loads off ONE resource (records and packets in one allocation, 32-bit byte offsets; an offset beyond the allocation reads zeros
                // without touching memory). Three loads serve both kinds of lane: a packet's three quads, or the NEAR planes of a record's four boxes along
                // x, y, z -- the ray's sign along an axis picks which of the record's lo / hi quads that is, so the planes arrive as
                // Bounds3f::intersect_p2 indexes them (`bounds[dir_is_neg[k]]`, bounds.rs:561-566) without a select per plane. A node lane's other five
                // loads of the same 128-byte line (far planes, references, order word) go out with them -- a leaf lane aims them past the end --, so a step is
                // one round trip and nobody waits inside a branch.
                v4u a0, a1, a2, b0, b1, b2, rf; uint32_t meta;
                if constexpr (!BIG) {
                    const uint32_t onx = nx ? 48u : 0u, ony = ny ? 64u : 16u, onz = nz ? 80u : 32u;
                    const uint32_t c3x = (cur << 1) + cur, qb = at_leaf ? (c3x << 4) + (s.leaf_off << 4) : (cur << 7);
                    const uint32_t nb = at_node ? qb : 0xffffff00u;
                    a0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, qb + (at_leaf ? (lofs & 0xffu) : onx), 0, 0);
                    a1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, qb + (at_leaf ? ((lofs >> 8) & 0xffu) : ony), 0, 0);
                    a2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, qb + (at_leaf ? (lofs >> 16) : onz), 0, 0);
                    b0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, nb + (48u - onx), 0, 0);
                    b1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, nb + (80u - ony), 0, 0);
                    b2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, nb + (112u - onz), 0, 0);
                    rf = __builtin_amdgcn_raw_buffer_load_b128(rsrc, nb + 96u, 0, 0);
                    meta = __builtin_amdgcn_raw_buffer_load_b32(rsrc, nb + 112u, 0, 0);
                } else {   // quads of the pool through 64-bit addresses; a leaf lane aims its five node loads at record 0 (one line the whole wave shares)
                    const uint32_t onx = nx ? 3u : 0u, ony = ny ? 4u : 1u, onz = nz ? 5u : 2u;
                    gpool_u32 *const rec = gpool + (at_leaf ? (size_t)cur * 3u + s.leaf_off : (size_t)cur * 8u), *const nrec = at_node ? rec : gpool;
                    a0 = rec[at_leaf ? ((lofs >> 4) & 0xfu) : onx]; a1 = rec[at_leaf ? ((lofs >> 12) & 0xfu) : ony]; a2 = rec[at_leaf ? (lofs >> 20) : onz];
                    b0 = nrec[3u - onx]; b1 = nrec[5u - ony]; b2 = nrec[7u - onz]; rf = nrec[6]; meta = nrec[7].x;
                }
                asm volatile("" :: "v"(a0.x), "v"(a1.x), "v"(a2.x), "v"(a2.w), "v"(b0.x), "v"(b1.x), "v"(b2.x), "v"(rf.x), "v"(meta));   // all eight before the node / leaf branch
                PT_UTIL_MARK(u_cfetch);
                q0 = make_uint4(a0.x, a0.y, a0.z, a0.w); q1 = make_uint4(a1.x, a1.y, a1.z, a1.w); q2 = make_uint4(a2.x, a2.y, a2.z, a2.w); q3 = make_uint4(0u, 0u, 0u, 0u);
                if (at_node) {
                    n_nodes++;   // records fetched
                    // Bounds3f::intersect_p2 (bounds.rs:559-580) for the four slots, the arithmetic of slab_geo, two slots per packed instruction:
                    // t = (plane - o) * inv_dir, far planes times 1 + 2 gamma(3)
                    const f2 kk2 = {1.0f + 2.0f * gammaf(3), 1.0f + 2.0f * gammaf(3)};
                    const f2 ox = {ro.x, ro.x}, oy = {ro.y, ro.y}, oz = {ro.z, ro.z}, ix = {inv_dir.x, inv_dir.x}, iy = {inv_dir.y, inv_dir.y}, iz = {inv_dir.z, inv_dir.z};
#define PT_F2(v, lo, hi) f2{__uint_as_float(v.lo), __uint_as_float(v.hi)}
                    const f2 tnx[2] = {(PT_F2(a0, x, y) - ox) * ix, (PT_F2(a0, z, w) - ox) * ix}, tfx[2] = {((PT_F2(b0, x, y) - ox) * ix) * kk2, ((PT_F2(b0, z, w) - ox) * ix) * kk2};
                    const f2 tny[2] = {(PT_F2(a1, x, y) - oy) * iy, (PT_F2(a1, z, w) - oy) * iy}, tfy[2] = {((PT_F2(b1, x, y) - oy) * iy) * kk2, ((PT_F2(b1, z, w) - oy) * iy) * kk2};
                    const f2 tnz[2] = {(PT_F2(a2, x, y) - oz) * iz, (PT_F2(a2, z, w) - oz) * iz}, tfz[2] = {((PT_F2(b2, x, y) - oz) * iz) * kk2, ((PT_F2(b2, z, w) - oz) * iz) * kk2};
#undef PT_F2
                    float T[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // the reference's conditional updates `if tymin > tmin { tmin = tymin }` are max / min except when the x-axis value is a NaN (a ray
                        // with d.x == 0 starting on a plane of the box): the reference then keeps the NaN and rejects the box -- `ord` below; a NaN along y or
                        // z is ignored by both forms
                        const float tminx = tnx[k >> 1][k & 1], tmaxx = tfx[k >> 1][k & 1], tymin = tny[k >> 1][k & 1], tymax = tfy[k >> 1][k & 1], tzmin = tnz[k >> 1][k & 1], tzmax = tfz[k >> 1][k & 1];
                        const bool miss_xy = (tminx > tymax) | (tymin > tmaxx);
                        const float tmin1 = __builtin_fmaxf(tminx, tymin), tmax1 = __builtin_fminf(tmaxx, tymax);
                        const bool miss_z = (tmin1 > tzmax) | (tzmin > tmax1);
                        const float tmin = __builtin_fmaxf(tmin1, tzmin), tmax = __builtin_fminf(tmax1, tzmax);
                        const bool ord = !__builtin_isunordered(tminx, tmaxx);
                        const bool ok = !(miss_xy | miss_z) & (tmax > 0.0f) & ord;   // (an empty slot's planes are +inf / -inf: it misses)
                        T[k] = ok ? tmin : __builtin_inff();   // a slot that fails gets entry distance +inf: it fails every `tmin < t_max` below
                    }
                    // the reference's order of the four slots (bvh.rs:728-751: the near child of a node first, near = the second child when the ray is
                    // negative along the node's split axis): inside each pair by the axis of L / R, between the pairs by the axis of N -- the record
                    // holds the three decisions for each of the eight sign octants
                    const uint32_t ord3 = meta >> sgn3;
                    const bool n0 = (ord3 & 1u) != 0u, n1 = (ord3 & 2u) != 0u, n2 = (ord3 & 4u) != 0u;
                    const float ta = n1 ? T[1] : T[0], tb = n1 ? T[0] : T[1], tc = n2 ? T[3] : T[2], td = n2 ? T[2] : T[3];
                    const uint32_t ra = n1 ? rf.y : rf.x, rb = n1 ? rf.x : rf.y, rc = n2 ? rf.w : rf.z, rd4 = n2 ? rf.z : rf.w;
                    const float t0 = n0 ? tc : ta, t1 = n0 ? td : tb, t2 = n0 ? ta : tc, t3 = n0 ? tb : td;
                    const uint32_t r0 = n0 ? rc : ra, r1 = n0 ? rd4 : rb, r2 = n0 ? ra : rc, r3 = n0 ? rb : rd4;
                    // the first slot that passes `tmin < t_max` now is entered (the reference tests it with this very t_max: the slots before it fail at this
                    // moment too, nothing is visited in between); later ones are pushed with their entry distance and pass or fail the same comparison when
                    // popped, against the t_max of then. A later slot that fails NOW may only be dropped if it fails then too -- and t_max is NOT monotone in
                    // the reference: Triangle::intersect accepts tscaled < t_max * det in scaled space (triangle.rs:202-206) and then rounds t = tscaled *
                    // (1 / det) (:208-213), which can land up to three roundings ABOVE the t_max it was tested against; `r.t_max = thit` (primitive.rs:137) then
                    // RAISES t_max by at most (1 + 3 * 2^-24) per accepted hit. A ray through a vertex shared by six triangles does it (the reference's own
                    // tests/shapes.rs:132-144, "shoot directly at a vertex": round 6 found 6 of its 200 000 rays returning another triangle of the fan than the
                    // reference, whose walk re-admitted a box it would have culled one hit earlier). So a later slot is dropped only beyond a slack of
                    // 2^-10: below that it is pushed and meets the exact comparison at its pop. t_max would have to be rounded upwards by more than 5 400
                    // SUCCESSIVE hits of one ray, each within three ulps of the last, to cross the slack (round-6 notes).
                    const bool m0 = t0 < t_max, m1 = t1 < t_max, m2 = t2 < t_max, m3 = t3 < t_max;
                    const float t_slack = t_max * (1.0f + 0x1p-10f);
                    const bool c3 = (t3 < t_slack) & (m0 | m1 | m2), c2 = (t2 < t_slack) & (m0 | m1), c1 = (t1 < t_slack) & m0;
                    if (sp + 3u <= (uint32_t)kLds) {   // the usual case: all three land in LDS, at slots known without a chain of sp updates
                        const uint32_t p2 = sp + (c3 ? 1u : 0u), p1 = p2 + (c2 ? 1u : 0u);
                        if (c3) { stack[(2 * sp) * 64] = r3; stack[(2 * sp + 1) * 64] = __float_as_uint(t3); }
                        if (c2) { stack[(2 * p2) * 64] = r2; stack[(2 * p2 + 1) * 64] = __float_as_uint(t2); }
                        if (c1) { stack[(2 * p1) * 64] = r1; stack[(2 * p1 + 1) * 64] = __float_as_uint(t1); }
                        sp = p1 + (c1 ? 1u : 0u);
                    } else if (sp + (c1 ? 1u : 0u) + (c2 ? 1u : 0u) + (c3 ? 1u : 0u) > (uint32_t)kMaxS) atomicMax(job.error, (uint32_t)PT_ERR_STACK_OVERFLOW);
                    else {
                        if (c3) push(r3, __float_as_uint(t3));
                        if (c2) push(r2, __float_as_uint(t2));
                        if (c1) push(r1, __float_as_uint(t1));
                    }
                    if (m0 | m1 | m2 | m3) {
                        const uint32_t nr = m0 ? r0 : (m1 ? r1 : (m2 ? r2 : r3));
                        cur = nr & kMaskRef;
                        state = (nr & kLeafBit) ? ST_LEAF : ST_ENTER;
                    } else need_pop = true;
                }
                PT_UTIL_MARK(u_cnode);
            } else {
                const uint4 *rec = at_leaf ? leaf4 + 3 * (size_t)cur : wide4 + 4 * (size_t)cur;
                // (a packet's three quads in the ray's order kx, ky, kz; q3 of a packet = start of the next one, the array is padded)
                q0 = rec[at_leaf ? (lofs >> 4) & 3u : 0u]; q1 = rec[at_leaf ? (lofs >> 12) & 3u : 1u]; q2 = rec[at_leaf ? lofs >> 20 : 2u]; q3 = rec[3];
                // Pin the whole record in front of the node / leaf branch: left alone, the compiler sinks the fields only the node path
                // reads (q2.z, q3.z) below the branch as two more dword loads, i.e. a second dependent L1 round trip in every node step
                // (measured: extend 143 -> 126 ms per step).
                asm volatile("" :: "v"(q2.z), "v"(q3.x), "v"(q3.y), "v"(q3.z));
            }
            if (QUAD && at_node) {
                // (the four-wide step above)
            } else if (at_node) {
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
                    if (pending > 63u || sp >= (uint32_t)kMaxS) atomicMax(job.error, (uint32_t)PT_ERR_STACK_OVERFLOW);
                    else { push(far_ref | (pending << 25), __float_as_uint(tmin_far)); pending = 0; }
                } else pending++;
                n_nodes++;  // the near child's test
                if (geo_near && tmin_near < t_max) {
                    cur = near_ref & kMaskRef;
                    state = (near_ref & kLeafBit) ? ST_LEAF : ST_ENTER;
                } else need_pop = true;
            } else {
                // leaf packets in ordered_prims order
                const uint32_t fl = tp_aux(q0.w, q1.w, q2.w, tray.kz, 2), li = cur;   // the packet's quads came in the ray's axis order
                const uint32_t shw = tp_aux(q0.w, q1.w, q2.w, tray.kz, 1);
                bool advance = true;   // false: the lane left the leaf (entered an instance / finished an any-hit ray)
                if (fl & TP_INSTANCE) {
                    if constexpr (INST) {
                        xf_arg = shw | ((fl & TP_LAST) ? 0x80000000u : 0u); state = ST_INST; advance = false;   // entered in the transform step below
                    }
                } else if (fl & TP_SPHERE) {
                    if constexpr (SPH) {  // GeometricPrimitive -> Sphere::intersect / intersect_p (sphere.rs:59-286)
                        n_sph++;
                        float t, phi; V3 ph, dobj;
                        if (sphere_hit(s.spheres[shw & 0x3fffffffu], ro, rd, t_max, lane_any, t, ph, phi, dobj)) {
                            if (lane_any) { hit_pkt = li; state = ST_DONE; advance = false; }
                            else {
                                t_max = t;
                                hit_pkt = li; hit_t = t; hb0 = hb1 = hb2 = 0.0f;
                                hit_inst = in_inst; inst_hit = in_inst != PT_NONE;
                            }
                        }
                    }
                } else {
                    n_tris++;
                    // vertices arrive permuted: quad j = the three vertices' coordinate along the ray's j-th permuted axis
                    const V3 p0t(__uint_as_float(q0.x) - rop.x, __uint_as_float(q1.x) - rop.y, __uint_as_float(q2.x) - rop.z);
                    const V3 p1t(__uint_as_float(q0.y) - rop.x, __uint_as_float(q1.y) - rop.y, __uint_as_float(q2.y) - rop.z);
                    const V3 p2t(__uint_as_float(q0.z) - rop.x, __uint_as_float(q1.z) - rop.y, __uint_as_float(q2.z) - rop.z);
                    float t, b0, b1, b2;
                    bool hit = tri_hit_core(p0t, p1t, p2t, tray, t_max, t, b0, b1, b2);
                    if constexpr (ALPHA) {
                        // Triangle::intersect (triangle.rs:275-285) / intersect_p (:497-545) with an alpha mask: the hit is
                        // discarded where the mask evaluates to 0; intersect_p then also rejects degenerate triangles
                        if (hit && (fl & TP_ALPHA) && !(fl & TP_BOGUS)) {
                            const uint32_t tri = shw & 0x3fffffffu;
                            const V3 p0 = ld3(s.P, s.indices[3 * tri]), p1 = ld3(s.P, s.indices[3 * tri + 1]), p2 = ld3(s.P, s.indices[3 * tri + 2]);   // (world-space vertices for the mask lookup)
                            P2 uv[3]; tri_uvs(s, tri, s.indices[3 * tri], s.indices[3 * tri + 1], s.indices[3 * tri + 2], uv);
                            TexCtx c; c.dpdx = V3(0.0f, 0.0f, 0.0f); c.dpdy = V3(0.0f, 0.0f, 0.0f); c.dudx = c.dvdx = c.dudy = c.dvdy = 0.0f;
                            c.p = p0 * b0 + p1 * b1 + p2 * b2;
                            c.uv = P2(uv[0].x * b0 + uv[1].x * b1 + uv[2].x * b2, uv[0].y * b0 + uv[1].y * b1 + uv[2].y * b2);
                            const int32_t a = s.tri_alpha ? s.tri_alpha[tri] : -1;
                            if (a >= 0 && tex_eval(s, a, c).r == 0.0f) hit = false;
                            if (lane_any && hit) { const int32_t sa = s.tri_shadow_alpha ? s.tri_shadow_alpha[tri] : -1; if (sa >= 0 && tex_eval(s, sa, c).r == 0.0f) hit = false; }
                        } else if (lane_any && hit && (fl & TP_ALPHA) && (fl & TP_BOGUS)) hit = false;
                    }
                    if (hit) {
                        if (lane_any) { hit_pkt = li; state = ST_DONE; advance = false; }
                        else if (!(fl & TP_BOGUS)) {  // triangle.rs:258-261
                            t_max = t;  // primitive.rs:137
                            hit_pkt = li; if (INST || PROBE) hit_t = t; hb0 = b0; hb1 = b1; hb2 = b2;
                            if (INST) { hit_inst = in_inst; inst_hit = in_inst != PT_NONE; }
                        }
                    }
                }
                if (advance) { if (fl & TP_LAST) need_pop = true; else cur = li + 1u; }
            }
            PT_UTIL_MARK(u_cleaf);
            if (need_pop) pop_next();
            PT_UTIL_MARK(u_cpop);
        }
#ifdef PT_TRACE_UTIL
        u_cmain += (unsigned long long)(clock64() - u_c1);
#endif
    }
    if (!MIX) { counter_add(&job.counters->nodes, n_nodes); counter_add(&job.counters->tri_tests, n_tris); }
    if (SPH) counter_add(&job.counters->sphere_tests, n_sph);
    if constexpr (MIX) {   // rays per kind from the wave's LDS slots: shadow_tests counts Scene::intersect_p calls, intersect_tests Scene::intersect calls
        if (lane < 3u) {
            const uint32_t kd = lane == 0u ? job.sub[0].kind : (lane == 1u ? job.sub[1].kind : job.sub[2].kind);
            const uint32_t is_any = lane == 0u ? job.sub[0].any : (lane == 1u ? job.sub[1].any : job.sub[2].any);
            const uint32_t kn = kcnt[3u * lane], kt = kcnt[3u * lane + 1u], kr = kcnt[3u * lane + 2u];
            if (kr) {
                atomicAdd(&job.counters->nodes, (unsigned long long)kn); atomicAdd(&job.counters->tri_tests, (unsigned long long)kt);
                atomicAdd(is_any ? &job.counters->shadow_tests : &job.counters->intersect_tests, (unsigned long long)kr);
                atomicAdd(&job.counters->k_nodes[kd], (unsigned long long)kn); atomicAdd(&job.counters->k_tris[kd], (unsigned long long)kt); atomicAdd(&job.counters->k_rays[kd], (unsigned long long)kr);
            }
        }
    } else {
        counter_add(ANY ? &job.counters->shadow_tests : &job.counters->intersect_tests, n_rays);
        counter_add(&job.counters->k_nodes[job.sub[0].kind], n_nodes);
        counter_add(&job.counters->k_tris[job.sub[0].kind], n_tris);
        counter_add(&job.counters->k_rays[job.sub[0].kind], n_rays);
    }
#ifdef PT_TRACE_UTIL
    if (blockIdx.x == 0 && threadIdx.x == 0) { job.counters->dbg[0] = job.leaf_quorum; job.counters->dbg[1] = job.refill_min; }
    if (lane == 0) {   // how long the launch's wave slots were occupied: a wave leaves when the queue is drained and its own rays are done
        const unsigned long long u_t1 = wall_clock64();
        atomicMin(&job.counters->tail[0], u_t0); atomicMax(&job.counters->tail[1], u_t1); atomicAdd(&job.counters->tail[4 + 2 * (job.sub[0].kind & 3)], u_t1 - u_t0);
    }
    for (int o = 32; o > 0; o >>= 1) { u_it3 += __shfl_xor(u_it3, o); u_act3 += __shfl_xor(u_act3, o); u_ent += __shfl_xor(u_ent, o); u_rej += __shfl_xor(u_rej, o); u_ihit += __shfl_xor(u_ihit, o); u_spill += __shfl_xor(u_spill, o); }
    if (lane == 0) { atomicAdd(&job.counters->util2[4], (unsigned long long)u_ent); atomicAdd(&job.counters->util2[5], (unsigned long long)u_rej); atomicAdd(&job.counters->util2[6], (unsigned long long)u_ihit); atomicAdd(&job.counters->util2[7], (unsigned long long)u_spill); }
    if (lane == 0) {
        atomicAdd(&job.counters->tail[12], (unsigned long long)u_it3); atomicAdd(&job.counters->tail[13], (unsigned long long)u_act3);
        atomicAdd(&job.counters->tail[14], u_cxf); atomicAdd(&job.counters->tail[15], u_cmain);
        atomicAdd(&job.counters->util2[0], u_cfetch); atomicAdd(&job.counters->util2[1], u_cnode); atomicAdd(&job.counters->util2[2], u_cleaf); atomicAdd(&job.counters->util2[3], u_cpop); atomicAdd(&job.counters->tail[2], (unsigned long long)(clock64() - (long long)u_t0c));
    }
    for (int o = 32; o > 0; o >>= 1) { u_it1 += __shfl_xor(u_it1, o); u_act1 += __shfl_xor(u_act1, o); u_it2 += __shfl_xor(u_it2, o); u_act2 += __shfl_xor(u_act2, o); }
    if (lane == 0) {
        atomicAdd(&job.counters->regions[4 * (job.sub[0].kind & 3) + 0], (unsigned long long)u_it1); atomicAdd(&job.counters->regions[4 * (job.sub[0].kind & 3) + 1], (unsigned long long)u_act1);
        atomicAdd(&job.counters->regions[4 * (job.sub[0].kind & 3) + 2], (unsigned long long)u_it2); atomicAdd(&job.counters->regions[4 * (job.sub[0].kind & 3) + 3], (unsigned long long)u_act2);
    }
#endif
}
#undef PT_SUB
#undef PT_GPTR
#undef PT_UTIL
#undef PT_UTIL_MARK
#undef PT_SGN3
#undef PT_TRI_RAY
