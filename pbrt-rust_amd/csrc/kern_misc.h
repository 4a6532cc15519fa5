// kern_misc.h -- split out of the former single-file kernels.hip so that the translation units compile in parallel.
#pragma once
#include "kern_common.h"
#include "kern_film.h"
// ---- scene preparation ---------------------------------------------------------------------------
// Triangle packets in leaf order + the per-triangle "degenerate -> intersect() always fails" flag
// (triangle.rs:254-261, evaluated once here instead of per accepted candidate).
__global__ void k_build_packets(DeviceScene s, const uint32_t *ordered, uint32_t n_refs, TriPacket *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_refs) return;
    uint32_t prim = ordered[i];      // primitive index, or PT_TOP_INSTANCE | instance index
    TriPacket p;
    if (prim & PT_TOP_INSTANCE) {
        p.x[0] = p.x[1] = p.x[2] = p.y[0] = p.y[1] = p.y[2] = p.z[0] = p.z[1] = p.z[2] = 0.0f;
        p.prim = PT_NONE; p.shape = prim & ~PT_TOP_INSTANCE; p.flags = TP_INSTANCE;
        out[i] = p;
        return;
    }
    uint32_t shape = s.prim_shape[prim];
    p.prim = prim; p.shape = shape; p.flags = 0;
    // material index + its shade-queue class ride in the flag word (dev_scene.h: kTpClassShift / kTpMatShift)
    const uint32_t mat = s.prim_material[prim];
    const uint32_t hi = ((mat == PT_NONE ? 0u : (uint32_t)s.mat_class[mat]) << kTpClassShift) | ((mat == PT_NONE || mat >= kTpMatNone ? kTpMatNone : mat) << kTpMatShift);
    if ((shape >> 30) == PT_SHAPE_TRIANGLE) {
        uint32_t tri = shape & 0x3fffffffu;
        uint32_t i0 = s.indices[3 * tri], i1 = s.indices[3 * tri + 1], i2 = s.indices[3 * tri + 2];
        V3 p0 = ld3(s.P, i0), p1 = ld3(s.P, i1), p2 = ld3(s.P, i2);
        P2 uv[3]; tri_uvs(s, tri, i0, i1, i2, uv);
        V3 dpdu, dpdv;
        bool ok = tri_partials(p0, p1, p2, uv, dpdu, dpdv);
        p.x[0] = p0.x; p.x[1] = p1.x; p.x[2] = p2.x; p.y[0] = p0.y; p.y[1] = p1.y; p.y[2] = p2.y; p.z[0] = p0.z; p.z[1] = p1.z; p.z[2] = p2.z;
        p.flags = (uint32_t)s.tri_flags[tri] | (ok ? 0u : (uint32_t)TP_BOGUS) | hi;
        if ((s.tri_alpha && s.tri_alpha[tri] >= 0) || (s.tri_shadow_alpha && s.tri_shadow_alpha[tri] >= 0)) p.flags |= TP_ALPHA;
    } else {
        p.x[0] = p.x[1] = p.x[2] = p.y[0] = p.y[1] = p.y[2] = p.z[0] = p.z[1] = p.z[2] = 0.0f;
        p.flags = TP_SPHERE | hi;
    }
    out[i] = p;
}

// Mark the last packet of every leaf (offsets of the last primitive of each leaf, computed on the host).
__global__ void k_mark_leaf_ends(TriPacket *leaf, const uint32_t *last_index, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) leaf[last_index[i]].flags |= TP_LAST;
}

__global__ void k_light_area(DeviceScene s, float *area, float4 *rec) {  // DiffuseAreaLight::new -> shape.area(); DeviceScene::light_rec
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n_lights) return;
    float a = 0.0f;
    const PtLight &L = s.lights[i];
    float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0, r2 = r0, r4 = r0, r5 = r0;
    if (L.type == PT_LIGHT_DIFFUSE_AREA) {
        uint32_t shape = s.prim_shape[L.prim];
        if ((shape >> 30) == PT_SHAPE_TRIANGLE) {
            uint32_t tri = shape & 0x3fffffffu;
            const V3 p0 = ld3(s.P, s.indices[3 * tri]), p1 = ld3(s.P, s.indices[3 * tri + 1]), p2 = ld3(s.P, s.indices[3 * tri + 2]);
            a = tri_area(p0, p1, p2);
            const uint32_t fl = s.tri_flags[tri];
            const bool flip = ((fl & PT_TRI_REVERSE_ORIENTATION) != 0) != ((fl & PT_TRI_SWAPS_HANDEDNESS) != 0);
            V3 ns = normalize(cross(p1 - p0, p2 - p0));                      // Triangle::sample (triangle.rs:566-577)
            if (!(fl & PT_TRI_HAS_N) && flip) ns = ns * -1.0f;
            V3 np = normalize(cross(p0 - p2, p1 - p2));                      // Triangle::intersect (triangle.rs:283-300), no shape
            if (flip) np = -np;
            P2 uv[3]; tri_uvs(s, tri, s.indices[3 * tri], s.indices[3 * tri + 1], s.indices[3 * tri + 2], uv);
            V3 dpdu, dpdv;
            const bool degenerate = !tri_partials(p0, p1, p2, uv, dpdu, dpdv);
            r4 = make_float4(ns.x, ns.y, ns.z, 1.0f / a); r5 = make_float4(np.x, np.y, np.z, 0.0f);
            r0 = make_float4(p0.x, p0.y, p0.z, __uint_as_float(fl | 0x100u | (degenerate ? 0x200u : 0u)));
            r1 = make_float4(p1.x, p1.y, p1.z, __uint_as_float(tri)); r2 = make_float4(p2.x, p2.y, p2.z, a);
        } else {  // Sphere::area (sphere.rs:291-293)
            const PtSphere &S = s.spheres[shape & 0x3fffffffu];
            a = S.kind == PT_QUADRIC_DISK ? S.phi_max * 0.5f * (S.radius * S.radius - S.inner_radius * S.inner_radius)   // Disk::area (disk.rs:120-122)
                                          : S.phi_max * S.radius * (S.z_max - S.z_min);
        }
    }
    area[i] = a;
    float4 *r = rec + 6 * (size_t)i;
    r[0] = r0; r[1] = r1; r[2] = r2; r[3] = make_float4(L.L[0], L.L[1], L.L[2], __uint_as_float(L.two_sided)); r[4] = r4; r[5] = r5;
}
// ---- material-class routing (material-sorted shade queues) ------------------------------------------------------
// Reads the hit record of every traced continuation ray and appends the path id to the shade queue of the hit
// material's class (escaped rays -> the miss class). Block-level staged appends, one global atomic per ~1800 entries per class: the returning atomics on the two or three
// class counters every block hammers are most of what the kernel waits on (tools/microbench/route_bench.hip: 14 ps per entry with 1024-entry staging queues WHATEVER the
// class is read from, 6.7 ps with 2048-entry ones; the gather of the hit records' flag words adds 6 ps) -- three blocks per CU is what 48 KB of staging queues per block leave room for.
// One staging queue per class the SCENE uses (RouteJob: at most NQ of them): <6, 2048> for scenes of up to six classes (every config of BASELINE.json), <12, 1024> beyond that
// (more classes spread the appends over more counters, which is what the longer queues were for).
template <int NQ, int CAP>
__global__ __launch_bounds__(256) void k_route(DeviceScene s, const uint32_t *queue, const uint32_t *count_ptr, PathSoA ps, uint32_t *class_count, RouteJob rj) {
    __shared__ LdsQueue<CAP> q[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) lq_init(q[k]);
    __syncthreads();
    const uint32_t count = *count_ptr;
    const uint32_t rounded = (count + 255u) & ~255u;
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < rounded; qi += gridDim.x * blockDim.x) {
        const bool valid = qi < count;
        uint32_t pid = 0, cls = (uint32_t)kMissClass;   // escaped rays: their own light kernel (k_shade_miss)
        if (valid) {
            pid = queue[qi];
            cls = (ps.hit_pflags(pid) >> kTpClassShift) & kTpClassMask;   // written by k_trace with the hit: the packet's class bits, kMissClass for a miss
        }
        const uint32_t slot = (uint32_t)(rj.slot_map >> (4u * cls)) & 15u;
        if (valid && slot == 15u && cls != rj.drop_cls) atomicMax(rj.error, (uint32_t)PT_ERR_UNSUPPORTED);   // a hit whose material class has no queue in this render would vanish silently (and k_film_final would take it for an escaped ray)
#pragma unroll
        for (int k = 0; k < NQ; ++k) if ((uint32_t)k < rj.n_slots) lq_push(q[k], pid, valid && slot == (uint32_t)k);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NQ; ++k) if ((uint32_t)k < rj.n_slots) lq_flush_nosync(q[k], class_count + rj.cls_of_slot[k], rj.buf[k], 256u, false);
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NQ; ++k) if ((uint32_t)k < rj.n_slots) lq_flush_nosync(q[k], class_count + rj.cls_of_slot[k], rj.buf[k], 0u, true);
}
// ---- camera rays -------------------------------------------------------------------------------------
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

// A fresh path's core record (64 B) and ray record (32 B) are staged in LDS by the lane that computed them and written by the block
// TRANSPOSED: thread t stores quad (t & 3) of path (t >> 2), so every store instruction of a wave covers 1 KB of consecutive HBM
// (a lane writing its own record quad by quad strides 64 B between lanes: measured 2.5 TB/s against 3.6 TB/s for the old 4-byte arrays).
// meta = dimension 5 after the camera sample, bounces 0, flags: camera ray.
constexpr int kGenPlane = 257;   // quads per LDS plane (256 paths + 1: the four planes of a path land in different banks)
struct GenStage { float4 core[4 * kGenPlane]; float4 ray[2 * kGenPlane]; uint32_t alive[256]; };
PT_DEV void stage_path(GenStage &g, uint32_t t, float pfx, float pfy, V3 o, V3 d, uint64_t index, uint32_t medium) {
    g.core[0 * kGenPlane + t] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);                                               // L, etascale
    g.core[1 * kGenPlane + t] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(5u | (PF_CAMERA_RAY << 24)));        // beta, meta
    g.core[2 * kGenPlane + t] = make_float4(__uint_as_float((uint32_t)index), __uint_as_float((uint32_t)(index >> 32)), pfx, pfy);
    g.core[3 * kGenPlane + t] = make_float4(__uint_as_float(medium), __uint_as_float(PT_NONE), 0.0f, 0.0f);        // medium (volpath: the camera's, perspective.rs:114), mis_medium
    g.ray[0 * kGenPlane + t] = make_float4(o.x, o.y, o.z, d.x); g.ray[1 * kGenPlane + t] = make_float4(d.y, d.z, 0.0f, 0.0f);
}
PT_DEV void flush_paths(const GenStage &g, const PathSoA &ps, uint32_t pid0, uint32_t t) {   // all 256 threads; pid0 = first path id of the block's batch
    float4 *core = reinterpret_cast<float4 *>(ps.core) + 4 * (size_t)pid0, *ray = reinterpret_cast<float4 *>(ps.ray) + 2 * (size_t)pid0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) { const uint32_t gq = j * 256u + t, p = gq >> 2, q = gq & 3u; if (g.alive[p]) core[gq] = g.core[q * kGenPlane + p]; }
#pragma unroll
    for (uint32_t j = 0; j < 2; ++j) { const uint32_t gq = j * 256u + t, p = gq >> 1, q = gq & 1u; if (g.alive[p]) ray[gq] = g.ray[q * kGenPlane + p]; }
}

__global__ __launch_bounds__(256) void k_generate(RenderConst rc, SobolTables tabs, PathSoA ps, uint32_t *q_ext, uint32_t *q_ext_count, DevCounters *counters) {
    // LDS copies of what every lane needs, folded by NIBBLE of the word they are indexed with (the XORs of lowdiscrepancy.rs:512-569 regrouped, as in the shade
    // kernels' Sobol' windows): the generator matrices of dimensions 0..4 (SobolTables::nib) and the two van-der-Corput matrices of m. A look-up per nibble
    // replaces a count-trailing-zeros step per set bit -- the kernel issued vector instructions 99.7 % of its cycles, two thirds of them in those bit loops.
    constexpr uint32_t kVdcNibbles = 13;   // 52 columns
    __shared__ uint32_t s_nib[kSobolNibbles * 5 * 16];
    __shared__ uint64_t s_vdc[2 * kVdcNibbles * 16];   // [M | MI][nibble][16]
    __shared__ LdsQueue<1024> s_q;
    __shared__ GenStage s_gen;
    lq_init(s_q);
    const uint32_t m = (uint32_t)rc.sobol.log2_resolution;
    sobol_stage_lds(s_nib, tabs.nib, 5u, threadIdx.x, blockDim.x);
    if (m > 0) for (uint32_t i = threadIdx.x; i < 2 * kVdcNibbles * 16; i += blockDim.x) {
        const uint32_t which = i / (kVdcNibbles * 16), e = i - which * (kVdcNibbles * 16), j = e >> 4, n = e & 15u;
        const uint64_t *col = (which ? tabs.vdc_inv : tabs.vdc) + (m - 1) * 52 + 4 * j;
        uint64_t v = 0;
        for (uint32_t b = 0; b < 4; ++b) if (n & (1u << b)) v ^= col[b];
        s_vdc[i] = v;
    }
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
                    stage_path(s_gen, threadIdx.x, pfx, pfy, o, d, index, rc.volpath ? rc.camera_medium : PT_NONE);
                    alive = true;
                } else {
                // sobol_interval_to_index (lowdiscrepancy.rs:512-543) through the nibble tables
                uint64_t index = 0;
                if (m != 0) {
                    index = sample << (m << 1);
                    uint64_t delta = 0;
                    for (uint64_t f = sample, j = 0; f != 0; f >>= 4, ++j) delta ^= s_vdc[j * 16 + (f & 15u)];
                    uint64_t b = ((uint64_t)((uint32_t)(px - rc.sobol.sb_min[0]) << m) | (uint64_t)(uint32_t)(py - rc.sobol.sb_min[1])) ^ delta;
                    for (uint32_t j = 0; b != 0; b >>= 4, ++j) index ^= s_vdc[kVdcNibbles * 16 + j * 16 + (b & 15u)];
                }
                // get_camera_sample (sampler.rs:170-180): pfilm = get_2d, time = get_1d, plens = get_2d
                const uint32_t v0 = sobol_bits_nib(s_nib, 5u * 16u, tabs.m32, index), v1 = sobol_bits_nib(s_nib + 16, 5u * 16u, tabs.m32 + 52, index), v2 = sobol_bits_nib(s_nib + 32, 5u * 16u, tabs.m32 + 104, index),
                               v3 = sobol_bits_nib(s_nib + 48, 5u * 16u, tabs.m32 + 156, index), v4 = sobol_bits_nib(s_nib + 64, 5u * 16u, tabs.m32 + 208, index);
                // sobol.rs:77-81: film dimensions are remapped to the pixel
                float fx = sobol_to_float(v0) * (float)rc.sobol.resolution + (float)rc.sobol.sb_min[0];
                fx = clampf(fx - (float)px, 0.0f, kOneMinusEps);
                float fy = sobol_to_float(v1) * (float)rc.sobol.resolution + (float)rc.sobol.sb_min[1];
                fy = clampf(fy - (float)py, 0.0f, kOneMinusEps);
                const float pfx = (float)px + fx, pfy = (float)py + fy;
                V3 o, d;
                camera_ray(rc, pfx, pfy, sobol_to_float(v2), P2(sobol_to_float(v3), sobol_to_float(v4)), o, d);
                stage_path(s_gen, threadIdx.x, pfx, pfy, o, d, index, rc.volpath ? rc.camera_medium : PT_NONE);
                alive = true;
                }
            }
        }
        s_gen.alive[threadIdx.x] = alive ? 1u : 0u;
        __syncthreads();
        flush_paths(s_gen, ps, pid - threadIdx.x, threadIdx.x);   // (the queue flush below holds the barrier that protects s_gen for the next round)
        lq_push(s_q, pid, alive);
        lq_sync_flush(s_q, q_ext_count, q_ext, 256u, false);
        n_alive += alive ? 1ull : 0ull;
    }
    lq_sync_flush(s_q, q_ext_count, q_ext, 0u, true);
    counter_add(&counters->camera_rays, n_alive);
}
// ---- film: kern_film.h (film_slot) -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_film(RenderConst rc, PathSoA ps, const float *filter_table, float *film_rgbw, DevCounters *counters) {
    film_slot(rc, ps, filter_table, film_rgbw, counters, [](uint32_t, RGB &) {});
}

// Film::merge_film_tile (film.rs:142-161): RGB sums -> XYZ, added to the caller's film.
__global__ void k_film_finish(const float *film_rgbw, float *film_xyzw, uint32_t npix) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    float rgb[3] = {film_rgbw[4 * i], film_rgbw[4 * i + 1], film_rgbw[4 * i + 2]}, xyz[3];
    rgb_to_xyz(rgb, xyz);
    film_xyzw[4 * i] += xyz[0]; film_xyzw[4 * i + 1] += xyz[1]; film_xyzw[4 * i + 2] += xyz[2]; film_xyzw[4 * i + 3] += film_rgbw[4 * i + 3];
}
// Film merge of the one-process multi-device path (pt_multi_render): the films of all replicas (the first device's own and the landing
// buffers of the peer copies) are summed quad by quad in replica order; accumulate != 0 adds the sum to dst, else dst = sum (dst may
// be src[0]).
__global__ void k_film_sum(FilmSumArgs a, float4 *dst, int accumulate, size_t n_quads) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_quads) return;
    float4 s = a.src[0][i];
    for (uint32_t k = 1; k < a.n; ++k) { const float4 b = a.src[k][i]; s = make_float4(s.x + b.x, s.y + b.y, s.z + b.z, s.w + b.w); }
    if (accumulate) { const float4 d = dst[i]; s = make_float4(d.x + s.x, d.y + s.y, d.z + s.z, d.w + s.w); }
    dst[i] = s;
}

// ---- spatial light distribution (lightdistrib.rs:151-228), all voxels precomputed ---------------------------
// `cells` == NULL: every voxel of the grid, func[cell][n_lights] (the eager form). Otherwise the n_cells voxels listed in `cells` (first
// touched in this wavefront iteration), written into blocks of `stride` floats: {func_int, -, -, -, func[n_lights], cdf[n_lights + 1]}.
__global__ __launch_bounds__(256) void k_light_grid_contrib(DeviceScene s, uint32_t nvx, uint32_t nvy, uint32_t nvz, float *func, const uint32_t *cells, size_t n_cells, size_t stride) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t ncell = cells ? n_cells : (size_t)nvx * nvy * nvz;
    if (gid >= ncell * s.n_lights) return;
    const uint32_t j = (uint32_t)(gid % s.n_lights);
    const size_t slot = gid / s.n_lights;
    const size_t cell = cells ? (size_t)cells[slot] : slot;
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
    func[cells ? slot * stride + 4 + j : gid] = contrib;
}
// Per voxel: floor at 0.001*avg, then Distribution1D::new (sampling.rs:12-34)
// (`cells` != NULL: the blocks of k_light_grid_contrib's listed form; the finished block's address is published in cell_ptr[cell])
__global__ __launch_bounds__(256) void k_light_grid_finish(uint32_t n_lights, size_t ncell, float *func, float *cdf, float *func_int, const uint32_t *cells, size_t stride, unsigned long long *cell_ptr) {
    const size_t cell = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= ncell) return;
    float *f = cells ? func + cell * stride + 4 : func + cell * n_lights, *c = cells ? f + n_lights : cdf + cell * (n_lights + 1);
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
    if (cells) { func[cell * stride] = fi; __threadfence(); cell_ptr[cells[cell]] = (unsigned long long)(func + cell * stride); }
    else func_int[cell] = fi;
}

// SpatialLightDistribution::lookup's "first touch" (lightdistrib.rs:233-337), once per wavefront iteration: every vertex of `queue` that is
// about to look its voxel up (kind 0: surface hits of a shade class below max_depth, path.rs:120-131; 1: medium vertices, volpath.rs:113-119;
// 2: the exit points of finished BSSRDF probe chains, path.rs:188) names the voxel; voxels without a distribution are listed once.
template <bool SPH>
__global__ __launch_bounds__(256) void k_light_touch(DeviceScene s, LightGrid g, PathSoA ps, const uint32_t *queue, const uint32_t *count_ptr, uint32_t kind, uint32_t max_depth,
                                                     uint32_t *req_flag, uint32_t *req_list, uint32_t *req_count) {
    const uint32_t count = *count_ptr;
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < count; qi += gridDim.x * blockDim.x) {
        const uint32_t pid = queue[qi];
        const uint32_t meta = ps.meta(pid);
        if ((meta >> 24) & PF_DEAD) continue;
        if (kind != 2u && ((meta >> 16) & 0xffu) >= max_depth) continue;
        const V3 ro(ps.ox(pid), ps.oy(pid), ps.oz(pid)), rd(ps.dx(pid), ps.dy(pid), ps.dz(pid));
        V3 p;
        if (kind == 1u) p = ro + rd * ps.hit_t(pid);
        else {
            if (ps.hit_prim(pid) == PT_NONE) continue;
            SurfaceInteraction si;
            fill_hit_pkt<SPH>(s, ps.hit_pkt(pid), SPH ? ps.hit_inst(pid) : PT_NONE, ro, rd, ps.hit_b0(pid), ps.hit_b1(pid), ps.hit_b2(pid), si);
            p = si.p;
        }
        const size_t cell = light_grid_cell(g, s, p);
        if (g.cell_ptr[cell] == g.zero_block && atomicExch(&req_flag[cell], 1u) == 0u) req_list[atomicAdd(req_count, 1u)] = (uint32_t)cell;
    }
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

// Dimensions >= 2 go through the shade kernels' own Sampler (dev_sampler.h): windows of eight dimensions from the LDS nibble tables, then from
// their HBM copy, the last dimensions one at a time -- so this parity entry checks the product's sampling code, index bits >= 32 and >= 40 included.
__global__ __launch_bounds__(256) void k_sobol_samples(SobolTables tabs, SobolParams sp, uint32_t n, const int32_t *pixel_xy, const uint32_t *sample_num,
                                                       uint32_t n_dims, float *out, uint64_t *out_index) {
    constexpr uint32_t LDS_DIMS = 56u;
    __shared__ uint32_t s_sobol[LDS_DIMS * kSobolNibWords];
    sobol_stage_lds(s_sobol, tabs.nib, LDS_DIMS, threadIdx.x, blockDim.x);
    __syncthreads();
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t px = pixel_xy[2 * i], py = pixel_xy[2 * i + 1];
    uint64_t index = sobol_interval_to_index(tabs, (uint32_t)sp.log2_resolution, sample_num[i], (uint32_t)(px - sp.sb_min[0]), (uint32_t)(py - sp.sb_min[1]));
    if (out_index) out_index[i] = index;
    for (uint32_t d = 0; d < 2 && d < n_dims; ++d) out[(size_t)i * n_dims + d] = sobol_pixel_dim(tabs.m32, sp, index, (int)d, d == 0 ? px : py);
    Sampler smp; smp.index = index; smp.m32 = tabs.m32; smp.nib = tabs.nib; smp.lds = s_sobol; smp.lds_dims = LDS_DIMS; smp.overflow = false; smp.halton = false;
    smp.prime = tabs.prime; smp.prime_sum = tabs.prime_sum; smp.perm = tabs.perm;
    for (uint32_t d = 2; d < n_dims;) {
        smp.dim = d;
        smp.load_window();
        for (uint32_t k = 0; k < 8 && d < n_dims; ++k, ++d) out[(size_t)i * n_dims + d] = smp.get_1d();
    }
}
__global__ void k_camera_rays(RenderConst rc, uint32_t n, const float *cs, float *out_o, float *out_d) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 o, d;
    camera_ray(rc, cs[5 * i], cs[5 * i + 1], cs[5 * i + 2], P2(cs[5 * i + 3], cs[5 * i + 4]), o, d);
    out_o[3 * i] = o.x; out_o[3 * i + 1] = o.y; out_o[3 * i + 2] = o.z;
    out_d[3 * i] = d.x; out_d[3 * i + 1] = d.y; out_d[3 * i + 2] = d.z;
}
// Parity entry (pt_dist1d_sample): the device's Distribution1D -- dist_sample_continuous (the environment map's rows and marginal, sampling.rs:38-64) or
// dist_sample_discrete (the light choice, sampling.rs:66-85) -- on a distribution the host built with Distribution1D::new (scene_create.hip: dist1d).
__global__ void k_dist1d_sample(const float *func, const float *cdf, float func_int, int n, int discrete, uint32_t n_u, const float *u, float *out_x, float *out_pdf, int32_t *out_off) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_u) return;
    Dist1D d{func, cdf, func_int, n};
    float pdf = 0.0f; int off = 0; float x = 0.0f;
    if (discrete) off = dist_sample_discrete(d, u[i], pdf);
    else x = dist_sample_continuous(d, u[i], pdf, off);
    out_x[i] = x; out_pdf[i] = pdf; out_off[i] = off;
}
