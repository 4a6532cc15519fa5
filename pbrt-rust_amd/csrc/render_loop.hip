// render_loop.hip -- the wavefront scheduler behind pt_render: workspace, light grids, launches, one pass, counters (host_common.h has the map).
#include "host_common.h"

namespace pth {
#ifdef PT_TRACE_UTIL
__global__ void k_trace_util_fold(DevCounters *dc, uint32_t kind, uint32_t waves, int reset) {   // per launch: span x waves, then re-arm min / max
    if (!reset) dc->tail[5 + 2 * kind] += (dc->tail[1] - dc->tail[0]) * waves;
    dc->tail[0] = ~0ull; dc->tail[1] = 0ull;
}
#endif

// Shade queues exist for the seven lobe-count / service classes always (the volumetric integrator uses all of them) and for the lobe-set classes the scene's materials map to.
static bool class_has_queue(const pt_scene *sc, int c) {
    // matte and the three service classes always (4 dead / escaped paths, 5 medium vertices, 6 also the waiting exit points of subsurface chains under volpath); a material
    // class when the scene has such materials, a lobe-count class also when a lobe-set class of the scene folds into it under volpath (class_general)
    if (c == 0 || c == kMissClass || c == kMediumClass || c == kSpecClass || sc->class_used[c]) return true;
    if (c == 1 && sc->class_used[kSpecClass]) return true;   // (the volumetric router folds the specular class into class 1)
    for (int k = kSpecClass + 1; k < kNumClasses; ++k) if (sc->class_used[k] && class_general((uint32_t)k) == (uint32_t)c) return true;
    return false;
}
static size_t n_class_queues(const pt_scene *sc) { size_t n = 0; for (int c = 0; c < kNumClasses; ++c) n += class_has_queue(sc, c) ? 1 : 0; return n; }

// any: 0 closest hit, 1 any hit (rays of job.sub[0]); 2 mixed: the queues of job.sub[0..2] in one launch (n_upper covers all three)
int launch_trace(pt_scene *sc, int any, TraceJob job, uint32_t n_upper, bool probe) {
    if (n_upper == 0) return PT_OK;
    const uint32_t knob = job.sub[0].kind == 4 ? 0 : (job.sub[0].kind & 3);
    job.refill_min = (probe && !g_refill_from_env) ? 24u : g_refill_min[knob]; job.leaf_quorum = g_leaf_quorum[knob];   // (probe chains: 24 measured best on C5, 16: +1.6 %)
    if (sc->ds.n_instances > 0 && !g_refill_from_env) job.refill_min = 8;   // rays through instanced scenes are long (S4: 200 node visits): idle lanes are refilled early (measured 24 -> 8: +9 %)
    uint32_t waves = (n_upper + 63) / 64;
    uint32_t blocks = std::min<uint32_t>((waves + 3) / 4, sc->spill_waves / 4);
    const int mode = (sc->ds.tri_alpha || sc->ds.tri_shadow_alpha) ? 2 : sc->ds.n_spheres > 0 ? 1 : sc->ds.n_instances > 0 ? 3 : 0;  // kern_trace.h: k_trace MODE
    job.inst_quorum = g_inst_quorum;
    if (sc->quad_walk_only && (g_trace_exact || sc->exact_walk_only)) return fail(PT_ERR_UNSUPPORTED, "the two-wide (exact) walk addresses 2^25 records / packets: this scene has the production walk only");
#ifdef PT_TRACE_UTIL
    hipLaunchKernelGGL(k_trace_util_fold, dim3(1), dim3(1), 0, sc->stream, sc->dc, job.sub[0].kind & 3u, 0u, 1);
#endif
    const bool quad = !g_trace_exact && !sc->exact_walk_only;   // production: the four-wide records; pt_set_trace_exact(1): the two-wide walk with the reference's node-visit counter
    const bool big = sc->pool_big;   // records + packets beyond 4 GB: the production walk through 64-bit addresses
    #define PT_LAUNCH_TRACE(A, M, P) do { if (quad && big) hipLaunchKernelGGL((k_trace<A, M, P, 2>), dim3(blocks), dim3(kTraceBlock), 0, sc->stream, sc->ds, job); \
                                          else if (quad) hipLaunchKernelGGL((k_trace<A, M, P, 1>), dim3(blocks), dim3(kTraceBlock), 0, sc->stream, sc->ds, job); \
                                          else hipLaunchKernelGGL((k_trace<A, M, P, 0>), dim3(blocks), dim3(kTraceBlock), 0, sc->stream, sc->ds, job); } while (0)
    #define PT_LAUNCH_TRACE_MODE(A, P) do { if (mode == 3) PT_LAUNCH_TRACE(A, 3, P); else if (mode == 2) PT_LAUNCH_TRACE(A, 2, P); else if (mode == 1) PT_LAUNCH_TRACE(A, 1, P); else PT_LAUNCH_TRACE(A, 0, P); } while (0)
    if (probe) PT_LAUNCH_TRACE_MODE(0, true);
    else if (any == 2) PT_LAUNCH_TRACE_MODE(2, false);
    else if (any == 1) PT_LAUNCH_TRACE_MODE(1, false);
    else PT_LAUNCH_TRACE_MODE(0, false);
    #undef PT_LAUNCH_TRACE_MODE
    #undef PT_LAUNCH_TRACE
#ifdef PT_TRACE_UTIL
    hipLaunchKernelGGL(k_trace_util_fold, dim3(1), dim3(1), 0, sc->stream, sc->dc, job.sub[0].kind & 3u, blocks * (kTraceBlock / 64), 0);
#endif
    sc->set_kernel(std::string("k_trace<") + std::to_string(any) + ", " + std::to_string(mode) + ", " + (probe ? "true" : "false") + ", " + (quad ? (big ? "2" : "1") : "0") + ">");
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

int ensure_workspace(pt_scene *sc, size_t capacity, size_t film_px) {
    if (!sc->stream) HIP_TRY(hipStreamCreate(&sc->stream));
    if (!sc->qc) {
        int st;
        if ((st = sc->dalloc(&sc->qc, 1))) return st;
        if ((st = sc->dalloc(&sc->dc, 1))) return st;
        sc->spill_waves = (uint32_t)g_num_cus * g_trace_waves_per_cu;  // resident persistent waves (LDS: 5 KB per wave)
        if ((st = sc->dalloc(&sc->spill, (size_t)sc->spill_waves * 64 * kSpillWords))) return st;
        if ((st = sc->dalloc(&sc->d_filter, 256))) return st;
        if (sc->has_bssrdf && (st = sc->dalloc(&sc->probe_ring, (size_t)sc->spill_waves * 64 * kProbeRing * 3))) return st;
    }
    if (capacity > sc->capacity) {
        // the present workspace goes first (its memory is needed for the larger one); from here to the last allocation the scene HAS no workspace: a failure
        // leaves capacity 0 and every pointer null, so the next render allocates afresh instead of running on a freed slab behind the old capacity
        auto drop = [&]() {
            if (sc->slab) hipFree(sc->slab); if (sc->qbuf) hipFree(sc->qbuf); if (sc->ext_slab) hipFree(sc->ext_slab); if (sc->bss_slab) hipFree(sc->bss_slab);
            sc->slab = nullptr; sc->qbuf = nullptr; sc->ext_slab = nullptr; sc->bss_slab = nullptr; sc->ext_capacity = 0; sc->capacity = 0;
        };
        auto oom = [&](const char *what, hipError_t e) { (void)hipGetLastError(); drop(); return fail(PT_ERR_OUT_OF_MEMORY, std::string(what) + ": " + hipGetErrorString(e)); };
        drop();
        size_t bytes = capacity * (size_t)kPathBytes + 4096;
        hipError_t e = hipMalloc(&sc->slab, bytes);
        if (e != hipSuccess) { sc->slab = nullptr; return oom("path-state slab", e); }
        char *p = (char *)sc->slab;   // hipMalloc returns 256-byte aligned memory; every record array starts on a 64-byte line
        PathSoA &ps = sc->ps;
        ps.core = (float *)p; p += capacity * (size_t)PathSoA::kCoreWords * 4;
        ps.nee = (float *)p; p += capacity * (size_t)PathSoA::kNeeWords * 4;
        ps.mis = (float *)p; p += capacity * (size_t)PathSoA::kMisWords * 4;
        ps.ray = (float *)p; p += capacity * (size_t)PathSoA::kRayWords * 4;
        ps.hit = (float *)p; p += capacity * (size_t)PathSoA::kHitWords * 4;
        if (sc->has_bssrdf) {
            e = hipMalloc(&sc->bss_slab, capacity * (size_t)kBssBytes);
            if (e != hipSuccess) { sc->bss_slab = nullptr; return oom("BSSRDF probe state", e); }
            BssSoA &bs = sc->bs;
            float4 *bp = (float4 *)sc->bss_slab;
            bs.probe = bp; bp += capacity * (size_t)BssSoA::kProbeQuads; bs.frame = bp; bp += capacity * (size_t)BssSoA::kFrameQuads; bs.coef = bp;
        }
        // queues: ext[2] + shade[2][classes] + shadow + mis (+ probe[2])
        size_t nq = 2 + 2 * n_class_queues(sc) + 2 + (sc->has_bssrdf ? 2 : 0);
        e = hipMalloc((void **)&sc->qbuf, nq * capacity * 4);
        if (e != hipSuccess) { sc->qbuf = nullptr; return oom("queues", e); }
        uint32_t *qp = sc->qbuf;
        for (int i = 0; i < 2; ++i) { sc->q.ext[i] = qp; qp += capacity; }
        for (int i = 0; i < 2; ++i) for (int c = 0; c < kNumClasses; ++c) { sc->q.shade[i][c] = nullptr; if (class_has_queue(sc, c)) { sc->q.shade[i][c] = qp; qp += capacity; } }
        sc->q.shadow = qp; qp += capacity; sc->q.mis = qp; qp += capacity;
        sc->q.probe[0] = sc->q.probe[1] = nullptr;
        if (sc->has_bssrdf) { sc->q.probe[0] = qp; qp += capacity; sc->q.probe[1] = qp; }
        sc->capacity = capacity;
    }
    if (film_px > sc->film_px) {
        if (sc->film_rgbw) hipFree(sc->film_rgbw);
        sc->film_rgbw = nullptr; sc->film_px = 0;
        if (hipError_t e = hipMalloc((void **)&sc->film_rgbw, film_px * 16); e != hipSuccess) { (void)hipGetLastError(); sc->film_rgbw = nullptr; return fail(PT_ERR_OUT_OF_MEMORY, std::string("device film: ") + hipGetErrorString(e)); }
        sc->film_px = film_px;
    }
    return PT_OK;
}

// SpatialLightDistribution::new's voxel counts (lightdistrib.rs:112-128)
void spatial_voxels(const pt_scene *sc, uint32_t nvox[3]) {
    float diag[3] = {sc->ds.wb_max[0] - sc->ds.wb_min[0], sc->ds.wb_max[1] - sc->ds.wb_min[1], sc->ds.wb_max[2] - sc->ds.wb_min[2]};
    int me = (diag[0] > diag[1] && diag[0] > diag[2]) ? 0 : (diag[1] > diag[2] ? 1 : 2);
    float bmax = diag[me];
    for (int i = 0; i < 3; ++i) {
        float v = std::round(diag[i] / bmax * 64.0f);
        uint32_t nv = (v > 0.0f) ? (uint32_t)v : 0u;  // `as usize` saturates, NaN -> 0
        nvox[i] = std::max<uint32_t>(1u, nv);
    }
}
constexpr size_t kEagerGridEntries = (size_t)1 << 25;   // voxels x lights up to which PT_LS_SPATIAL precomputes every voxel

int ensure_light_grid(pt_scene *sc, int requested, int &effective) {
    effective = requested;
    if (requested == PT_LS_UNIFORM || sc->n_lights == 1) effective = PT_LS_UNIFORM;  // lightdistrib.rs:21
    if (requested > PT_LS_SPATIAL_LAZY) effective = PT_LS_SPATIAL;
    if (effective == PT_LS_SPATIAL) {   // the form is the library's choice (include/mi355pt.h: PtLightStrategy)
        uint32_t nv[3]; spatial_voxels(sc, nv);
        effective = (size_t)nv[0] * nv[1] * nv[2] * std::max(1u, sc->n_lights) <= kEagerGridEntries ? PT_LS_SPATIAL_EAGER : PT_LS_SPATIAL_LAZY;
    }
    LightGrid &g = sc->grid[effective];
    if (sc->grid_ready[effective]) return PT_OK;
    g.strategy = effective >= PT_LS_SPATIAL ? (int)PT_LS_SPATIAL : effective; g.n_lights = sc->n_lights; g.nvox[0] = g.nvox[1] = g.nvox[2] = 1;
    g.cell_ptr = nullptr; g.zero_block = 0; g.missing = nullptr;
    const uint32_t nl = sc->n_lights;
    if (nl == 0) { g.strategy = PT_LS_UNIFORM; g.func = g.cdf = g.func_int = nullptr; sc->grid_ready[effective] = true; return PT_OK; }
    if (effective == PT_LS_UNIFORM || effective == PT_LS_POWER) {
        std::vector<float> func(nl, 1.0f), cdf; float fi;
        if (effective == PT_LS_POWER) {  // compute_light_power_distribution (integrator.rs:239-247): Light::power().y() per light
            std::vector<float> area(nl);
            HIP_TRY(hipMemcpy(area.data(), sc->ds.light_area, nl * sizeof(float), hipMemcpyDeviceToHost));
            const float wr = sc->ds.world_radius;
            for (uint32_t i = 0; i < nl; ++i) {
                const PtLight &L = sc->host_lights[i];
                RGB c(L.L[0], L.L[1], L.L[2]), p(0.0f);
                switch (L.type) {
                case PT_LIGHT_DIFFUSE_AREA: p = c * area[i] * kPi; break;                                  // diffuse.rs:82-84
                case PT_LIGHT_DISTANT: p = c * kPi * wr * wr; break;                                      // distant.rs:47-50
                case PT_LIGHT_POINT: p = c * 4.0f * kPi; break;                                           // point.rs:44-46
                case PT_LIGHT_SPOT: p = c * 2.0f * kPi * (1.0f - 0.5f * (L.cos_falloff_start + L.cos_total_width)); break;  // spot.rs:64-66
                case PT_LIGHT_INFINITE: {                                                                  // infinite.rs:103-109
                    p = RGB(sc->env_texel0[0], sc->env_texel0[1], sc->env_texel0[2]) * wr * wr * kPi; break;   // env_texel0 = PtSceneDesc.env_power_lookup
                }
                default: break;
                }
                func[i] = p.y();
            }
        }
        dist1d(func, cdf, fi);
        int st;
        if ((st = sc->upload(&g.func, func.data(), nl))) return st;
        if ((st = sc->upload(&g.cdf, cdf.data(), nl + 1))) return st;
        if ((st = sc->upload(&g.func_int, &fi, 1))) return st;
    } else if (effective == PT_LS_SPATIAL_LAZY) {   // voxels filled on first touch (lazy_light_fill, called from run_pass)
        spatial_voxels(sc, g.nvox);
        pt_scene::LazyGrid &z = sc->lazy;
        z.ncell = (size_t)g.nvox[0] * g.nvox[1] * g.nvox[2];
        z.stride = ((size_t)4 + nl + nl + 1 + 3) & ~(size_t)3;   // {func_int, -, -, -} func[nl] cdf[nl + 1], whole quads
        int st;
        if ((st = sc->dalloc(&z.cell_ptr, z.ncell))) return st;
        if ((st = sc->dalloc(&z.zero_block, z.stride))) return st;
        if ((st = sc->dalloc(&z.req_flag, z.ncell))) return st;
        if ((st = sc->dalloc(&z.req_list, z.ncell))) return st;
        if ((st = sc->dalloc(&z.req_count, 2))) return st;
        z.missing = z.req_count + 1;
        HIP_TRY(hipMemset(z.zero_block, 0, z.stride * 4));
        HIP_TRY(hipMemset(z.req_flag, 0, z.ncell * 4));
        HIP_TRY(hipMemset(z.req_count, 0, 8));
        std::vector<unsigned long long> init(z.ncell, (unsigned long long)z.zero_block);
        HIP_TRY(hipMemcpy(z.cell_ptr, init.data(), z.ncell * 8, hipMemcpyHostToDevice));
        g.func = g.cdf = g.func_int = nullptr;
        g.cell_ptr = z.cell_ptr; g.zero_block = (unsigned long long)z.zero_block; g.missing = z.missing;
    } else {  // SpatialLightDistribution::new (lightdistrib.rs:112-128), every voxel precomputed on device
        spatial_voxels(sc, g.nvox);
        size_t ncell = (size_t)g.nvox[0] * g.nvox[1] * g.nvox[2];
        if (ncell * nl > ((size_t)1 << 31)) return fail(PT_ERR_UNSUPPORTED, "PT_LS_SPATIAL_EAGER: voxels x lights > 2^31 (PT_LS_SPATIAL picks the first-touch form for such scenes)");
        float *func, *cdf, *fint; int st;
        if ((st = sc->dalloc(&func, ncell * nl))) return st;
        if ((st = sc->dalloc(&cdf, ncell * (nl + 1)))) return st;
        if ((st = sc->dalloc(&fint, ncell))) return st;
        size_t total = ncell * nl;
        sc->begin("light_grid", total);
        sc->set_kernel("k_light_grid_contrib");
        hipLaunchKernelGGL(k_light_grid_contrib, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sc->stream, sc->ds, g.nvox[0], g.nvox[1], g.nvox[2], func, (const uint32_t *)nullptr, (size_t)0, (size_t)0);
        hipLaunchKernelGGL(k_light_grid_finish, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, sc->stream, nl, ncell, func, cdf, fint, (const uint32_t *)nullptr, (size_t)0, (unsigned long long *)nullptr);
        sc->end();
        HIP_TRY(hipGetLastError());
        g.func = func; g.cdf = cdf; g.func_int = fint;
    }
    sc->grid_ready[effective] = true;
    return PT_OK;
}

void fill_render_const(const PtRenderParams *rp, RenderConst &rc) {
    std::memset(&rc, 0, sizeof rc);
    std::memcpy(rc.sample_bounds, rp->sample_bounds, 16);
    std::memcpy(rc.pixel_bounds, rp->pixel_bounds, 16);
    std::memcpy(rc.crop, rp->cropped_pixel_bounds, 16);
    int32_t dx = rp->sample_bounds[2] - rp->sample_bounds[0], dy = rp->sample_bounds[3] - rp->sample_bounds[1];
    rc.ntx = (uint32_t)((dx + 15) / 16); rc.nty = (uint32_t)((dy + 15) / 16);
    // SobolSampler::new (sobol.rs:42-44)
    int32_t v = std::max(dx, dy); v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++;
    rc.sobol.resolution = v; rc.sobol.log2_resolution = 31 - __builtin_clz((uint32_t)v);
    rc.sobol.sb_min[0] = rp->sample_bounds[0]; rc.sobol.sb_min[1] = rp->sample_bounds[1];
    if (rp->sampler_type == PT_SAMPLER_HALTON) {   // HaltonSampler::new (halton.rs:62-110), kMaxResolution = 128
        rc.halton.enabled = 1; rc.halton.at_center = rp->sample_at_pixel_center ? 1u : 0u;
        const int32_t res[2] = {dx, dy};
        for (int i = 0; i < 2; ++i) {
            const uint32_t base = i == 0 ? 2u : 3u;
            uint32_t scale = 1, e = 0;
            while ((int64_t)scale < (int64_t)std::min(res[i], 128)) { scale *= base; e++; }
            rc.halton.base_scale[i] = scale; rc.halton.base_exp[i] = e;
        }
        rc.halton.stride = rc.halton.base_scale[0] * rc.halton.base_scale[1];
        auto mult_inverse = [](int64_t a, int64_t n) {   // extended_gcd + mod_ (halton.rs:19-35)
            int64_t x0 = 1, x1 = 0, aa = a, bb = n;      // iterative form of the same recurrence: x with a*x = gcd (mod n)
            while (bb != 0) { const int64_t q = aa / bb; int64_t t = aa - q * bb; aa = bb; bb = t; t = x0 - q * x1; x0 = x1; x1 = t; }
            return ((x0 % n) + n) % n;
        };
        rc.halton.mult_inv[0] = (uint32_t)mult_inverse(rc.halton.base_scale[1], rc.halton.base_scale[0]);
        rc.halton.mult_inv[1] = (uint32_t)mult_inverse(rc.halton.base_scale[0], rc.halton.base_scale[1]);
    }
    std::memcpy(rc.raster_to_camera.m, rp->raster_to_camera, 64);
    std::memcpy(rc.camera_to_world.m, rp->camera_to_world, 64);
    rc.lens_radius = rp->lens_radius; rc.focal_distance = rp->focal_distance;
    {   // PerspectiveCamera::new (perspective.rs:64-70): dx_camera / dy_camera
        const V3 p2t = xf_point(rc.raster_to_camera, V3(0.0f, 0.0f, 0.0f));
        const V3 dx = xf_point(rc.raster_to_camera, V3(1.0f, 0.0f, 0.0f)) - p2t, dy = xf_point(rc.raster_to_camera, V3(0.0f, 1.0f, 0.0f)) - p2t;
        rc.dx_camera[0] = dx.x; rc.dx_camera[1] = dx.y; rc.dx_camera[2] = dx.z;
        rc.dy_camera[0] = dy.x; rc.dy_camera[1] = dy.y; rc.dy_camera[2] = dy.z;
        rc.inv_sqrt_spp = 1.0f / std::sqrt((float)rp->spp);
    }
    rc.shutter_open = rp->shutter_open; rc.shutter_close = rp->shutter_close;
    rc.max_depth = rp->max_depth; rc.rr_threshold = rp->rr_threshold;
    rc.volpath = rp->integrator == PT_INTEGRATOR_VOLPATH ? 1u : 0u; rc.camera_medium = rc.volpath ? rp->camera_medium : PT_NONE;
    rc.filter_radius[0] = rp->filter_radius[0]; rc.filter_radius[1] = rp->filter_radius[1];
    rc.max_sample_luminance = rp->max_sample_luminance;
    rc.film_w = (uint32_t)(rp->cropped_pixel_bounds[2] - rp->cropped_pixel_bounds[0]);
    rc.film_h = (uint32_t)(rp->cropped_pixel_bounds[3] - rp->cropped_pixel_bounds[1]);
    rc.spp = rp->spp;
    rc.tile_world = rp->tile_world ? rp->tile_world : 1; rc.tile_rank = rp->tile_rank;
}

__global__ void k_reset(QCounters *qc, uint32_t mask, int cur) {
    // mask bit0: next ext + next shade queues; bit1: shadow + mis; bit2: trace heads; bit3: current ext + shade
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int nxt = 1 - cur;
    if (mask & 1u) { qc->ext[nxt] = 0; qc->probe[nxt] = 0; for (int c = 0; c < kNumClasses; ++c) qc->shade[nxt][c] = 0; }
    if (mask & 2u) { qc->shadow = 0; qc->mis = 0; }
    if (mask & 4u) { for (int i = 0; i < 4; ++i) qc->head[i] = 0; }
    if (mask & 8u) { qc->ext[cur] = 0; qc->probe[cur] = 0; for (int c = 0; c < kNumClasses; ++c) qc->shade[cur][c] = 0; }
}

template <int MAXL, int DIFF = 0> void launch_shade(pt_scene *sc, const RenderConst &rc, const LightGrid &grid, const ShadeJob &job, uint32_t upper) {
#ifndef PT_SHADE_BLOCKS_PER_CU
#define PT_SHADE_BLOCKS_PER_CU 24u   // experiment hook. Each block walks a fixed stride of the queue: 24 a CU (2, 3 or 4 resident at a time) even out the per-vertex cost differences; 8 left the matte kernel's third round two-thirds full (80.1 -> 76.3 ms on C2)
#endif
    const uint32_t blocks = std::min<uint32_t>((upper + 255) / 256, (uint32_t)g_num_cus * PT_SHADE_BLOCKS_PER_CU);  // persistent blocks: the LDS Sobol' table is staged once per block
    const int mode = rc.volpath ? 3 : sc->ds.n_textures > 0 ? 2 : (sc->ds.n_spheres > 0 || sc->ds.n_instances > 0 || rc.halton.enabled) ? 1 : 0;
    if (DIFF >= 3 && (mode == 3 || (DIFF == 6 && mode == 2))) { launch_shade<MAXL, 0>(sc, rc, grid, job, upper); return; }   // (volpath folds the lobe-set classes back; the smooth-subsurface form is untextured only)
    sc->set_kernel("k_shade<" + std::to_string(MAXL) + ", " + std::to_string(mode) + ", " + std::to_string(DIFF) + ">");
    if (rc.volpath) hipLaunchKernelGGL((k_shade<MAXL, 3, (DIFF >= 2) ? 0 : DIFF>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, rc, g_tabs, grid, sc->ps, job);
    else if (sc->ds.n_textures > 0) hipLaunchKernelGGL((k_shade<MAXL, 2, DIFF == 6 ? 0 : DIFF>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, rc, g_tabs, grid, sc->ps, job);
    else if (sc->ds.n_spheres > 0 || sc->ds.n_instances > 0 || rc.halton.enabled) hipLaunchKernelGGL((k_shade<MAXL, 1, DIFF>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, rc, g_tabs, grid, sc->ps, job);
    else hipLaunchKernelGGL((k_shade<MAXL, 0, DIFF>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, rc, g_tabs, grid, sc->ps, job);
}

// Samples per pass when the caller leaves the choice to the library (PtRenderParams.spp_per_pass = 0): as many paths in flight as the
// memory allows, up to 2^28 (69 GB of path state + 17-36 GB of queues / probe state out of 288 GB). Every wavefront iteration ends
// in a tail of straggling rays (~0.8 ms on S2, whatever the launch size), so fewer, larger iterations spend less of the render in tails:
// S2 at 1080p x 256 spp: 32 samples per pass 1267, 64: 1367, 128: 1439, 256: 1468 Msamples/s. `share` = renders that will hold a
// workspace on this device at the same time (pt_multi_render with a device listed more than once).
#ifndef PT_PASS_MAX_PATHS_LOG2
#define PT_PASS_MAX_PATHS_LOG2 29   // round 3: 2^29 paths = 256 samples per pixel at 1080p in ONE pass (137 GB of path state + 39 GB of queues of the 288 GB): half the iterations of 2^28
#endif
constexpr size_t kPassMaxPaths = (size_t)1 << PT_PASS_MAX_PATHS_LOG2;
constexpr double kPassMemFraction = 0.65;   // of the device's free memory
uint32_t choose_pass_size(const pt_scene *sc, uint32_t n_pix_slots, uint32_t spp, uint32_t share, bool volpath) {
    // per path: the five state records, the queues, the probe state of scenes with subsurface materials and -- volpath through material-less shells -- the
    // 128-byte chain record (PathSoA::ext, allocated after the main slab: left out of this sum, a shell scene asked for ~1.4x its budget)
    const size_t per_path = (size_t)kPathBytes + 4u * (2 + 2 * n_class_queues(sc) + 2 + (sc->has_bssrdf ? 2 : 0)) + (sc->has_bssrdf ? (size_t)kBssBytes : 0u)
                            + ((volpath && sc->has_null_material) ? 4u * (size_t)PathSoA::kExtWords : 0u);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
    // (the present workspace is freed before a larger one is allocated)
    const size_t afford = std::max(sc->capacity, (size_t)((double)(free_b / std::max(1u, share) + sc->capacity * per_path) * kPassMemFraction) / per_path);
    const size_t paths = std::min(afford, kPassMaxPaths / std::max(1u, share));
    uint32_t S = (uint32_t)std::min<size_t>(spp, std::max<size_t>(1, paths / std::max(1u, n_pix_slots)));
    const uint32_t n_pass = (spp + S - 1) / S;
    return (spp + n_pass - 1) / n_pass;   // passes of equal size
}

// First-touch voxels of PT_LS_SPATIAL_LAZY. touch: the vertices of one queue name their voxels; fill: the voxels named since the last fill are
// computed (k_light_grid_contrib over the list: 128 Halton points x every light each, lightdistrib.rs:151-228) and published in cell_ptr.
int lazy_light_touch(pt_scene *sc, const RenderConst &rc, const LightGrid &grid, const uint32_t *queue, const uint32_t *count, uint32_t n_upper, uint32_t kind) {
    if (!grid.cell_ptr || n_upper == 0) return PT_OK;
    pt_scene::LazyGrid &z = sc->lazy;
    const unsigned blocks = std::min<uint32_t>((n_upper + 255) / 256, (uint32_t)g_num_cus * 16u);
    const bool sph = sc->ds.n_spheres > 0 || sc->ds.n_instances > 0;
    sc->begin("light_touch", n_upper); sc->set_kernel(sph ? "k_light_touch<true>" : "k_light_touch<false>");
    if (sph) hipLaunchKernelGGL((k_light_touch<true>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, grid, sc->ps, queue, count, kind, rc.max_depth, z.req_flag, z.req_list, z.req_count);
    else hipLaunchKernelGGL((k_light_touch<false>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, grid, sc->ps, queue, count, kind, rc.max_depth, z.req_flag, z.req_list, z.req_count);
    sc->end();
    return PT_OK;
}
int lazy_light_fill(pt_scene *sc, const LightGrid &grid) {
    if (!grid.cell_ptr) return PT_OK;
    pt_scene::LazyGrid &z = sc->lazy;
    uint32_t n_new = 0;
    HIP_TRY(hipMemcpyAsync(&n_new, z.req_count, 4, hipMemcpyDeviceToHost, sc->stream));
    HIP_TRY(hipStreamSynchronize(sc->stream));
    if (n_new == 0) return PT_OK;
    if (n_new > z.ncell) return fail(PT_ERR_HIP, "light grid: more voxels requested than the grid holds");
    const uint32_t nl = grid.n_lights;
    // in batches of at most 2^31 (voxel, light) pairs per launch and 1 GiB of blocks per allocation
    const size_t per_batch = std::max<size_t>(1, std::min<size_t>(((size_t)1 << 31) / std::max(1u, nl), ((size_t)1 << 28) / z.stride));
    for (size_t first = 0; first < n_new; first += per_batch) {
        const size_t n = std::min<size_t>(per_batch, n_new - first);
        // blocks live as long as the scene (their addresses are published in cell_ptr): carved from slabs of >= 64 MB, so that a long render that keeps
        // touching a few new voxels per iteration makes a handful of allocations instead of one per iteration
        const size_t need = n * z.stride;
        if (need > z.arena_left) {
            const size_t slab = std::max<size_t>(need, (size_t)16 << 20);   // floats
            int st;
            if ((st = sc->dalloc(&z.arena, slab))) return st;
            z.arena_left = slab;
        }
        float *blocks = z.arena; z.arena += need; z.arena_left -= need;
        const size_t total = n * nl;
        sc->begin("light_grid", total); sc->set_kernel("k_light_grid_contrib");
        hipLaunchKernelGGL(k_light_grid_contrib, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, sc->stream, sc->ds, grid.nvox[0], grid.nvox[1], grid.nvox[2], blocks, (const uint32_t *)(z.req_list + first), n, z.stride);
        hipLaunchKernelGGL(k_light_grid_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sc->stream, nl, n, blocks, (float *)nullptr, (float *)nullptr, (const uint32_t *)(z.req_list + first), z.stride, z.cell_ptr);
        sc->end();
    }
    HIP_TRY(hipMemsetAsync(z.req_count, 0, 4, sc->stream));
    HIP_TRY(hipGetLastError());
    z.filled += n_new;
    return PT_OK;
}

// The text behind a status a kernel raised (QCounters::error; the largest code wins when several were raised)
const char *device_error_text(uint32_t code) {
    switch (code) {
    case PT_ERR_STACK_OVERFLOW: return "BVH traversal stack overflow: a ray needed more than 96 pending entries in the four-wide production walk / 64 in the two-wide exact walk (the reference's 64-entry stack, accelerators/bvh.rs:722, has no check)";
    case PT_ERR_PROBE_CHAIN: return "BSSRDF probe chain with more than 2^32 - 1 intersections";
    case PT_ERR_SOBOL_DIMENSIONS: return "a path asked for sampler dimension >= 1024 (Sobol') / 1000 (Halton): the reference panics there (samplers/sobol.rs:69-73)";
    case PT_ERR_UNSUPPORTED: return "a hit's material class has no shade queue in this render (internal: k_route)";
    default: return "error raised on the device";
    }
}

int run_pass(pt_scene *sc, RenderConst &rc, const LightGrid &grid, bool rp_profile_exact) {
    const uint32_t total = rc.n_pix_slots * rc.s_count;
    QCounters *qc = sc->qc;
    HIP_TRY(hipMemsetAsync(qc, 0, offsetof(QCounters, error), sc->stream));
    sc->begin("generate", total);
        sc->set_kernel("k_generate");
    #ifndef PT_GEN_BLOCKS_PER_CU
#define PT_GEN_BLOCKS_PER_CU 40u   // experiment hook (16 -> 40: k_generate 12.2 -> 11.7 ms on C2; the miss kernel does not care)
#endif
    hipLaunchKernelGGL(k_generate, dim3(std::min<uint32_t>((total + 255) / 256, (uint32_t)g_num_cus * PT_GEN_BLOCKS_PER_CU)), dim3(256), 0, sc->stream, rc, g_tabs, sc->ps, sc->q.ext[0], &qc->ext[0], sc->dc);
    sc->end();
    // The plain path integrator's paths are ended by the film kernel (k_film_final, kern_aux.h): no miss class, no k_shade_miss pass. Volpath keeps the pass (its
    // transmittance estimates draw sampler dimensions in iteration order).
    const bool fin = g_film_final && !rc.volpath;
    int cur = 0;
    static const char *shade_names[kNumClasses] = {"shade_matte", "shade_1lobe", "shade_2lobe", "shade_manylobe", "shade_miss", "shade_medium", "shade_specular", "shade_metal", "shade_plastic", "shade_uber", "shade_sss"};
    const int kMaxIterations = g_test_max_iterations > 0 ? g_test_max_iterations : 1 << 20;   // a path needs <= max_depth + null-surface skips + probe segments iterations (PT_TEST_MAX_ITERATIONS: the error path's test)
    for (int iter = 0; iter <= kMaxIterations; ++iter) {
        QCounters h;
        HIP_TRY(hipMemcpyAsync(&h, qc, sizeof h, hipMemcpyDeviceToHost, sc->stream));
        HIP_TRY(hipStreamSynchronize(sc->stream));
        if (h.error) return fail((int)h.error, device_error_text(h.error));
        if (iter == kMaxIterations) return fail(PT_ERR_PROBE_CHAIN, "pass did not finish within " + std::to_string(kMaxIterations) + " wavefront iterations");
        if (iter > 0 && iter % 2048 == 0 && getenv("PT_DEBUG_ITER")) {
            fprintf(stderr, "[iter %d] ext %u shadow %u mis %u probe %u shade:", iter, h.ext[cur], h.shadow, h.mis, h.probe[cur]);
            for (int c = 0; c < kNumClasses; ++c) fprintf(stderr, " %u", h.shade[cur][c]);
            fprintf(stderr, "\n");
        }
        const uint32_t n_ext = h.ext[cur], n_resolve = h.shade[cur][kMissClass], n_shadow = h.shadow, n_mis = h.mis, n_probe = h.probe[cur];
        // volpath with grid media: vertices that did their NEE set-up last iteration wait in their own shade class for stage B
        uint32_t n_stage_b = 0;
        if (rc.volpath && (sc->ds.has_grid || sc->ds.has_shells)) for (int c = 0; c < kNumClasses; ++c) if (c != kMissClass) n_stage_b += h.shade[cur][c];
        if (n_ext == 0 && n_resolve == 0 && n_probe == 0 && n_stage_b == 0 && (!fin || (n_shadow == 0 && n_mis == 0))) break;   // (fin: the last vertices' shadow / MIS rays are still to be traced)
        hipLaunchKernelGGL(k_reset, dim3(1), dim3(64), 0, sc->stream, qc, 4u | 1u, cur);
        TraceJob tj{};
        tj.spill = sc->spill; tj.error = &qc->error; tj.counters = sc->dc; tj.head = &qc->head[0];
        PathSoA &ps = sc->ps;
        // continuation rays -> hit record (routed to the material classes below)
        TraceSub ext{};
        ext.queue = sc->q.ext[cur]; ext.count = &qc->ext[cur]; ext.scalar_tmax = INFINITY;
        ext.ray = (const float4 *)ps.ray; ext.ray_stride = PathSoA::kRayWords / 4;
        ext.out_hit = (float4 *)ps.hit; ext.out_hit_stride = PathSoA::kHitWords / 4; ext.out_hit2 = (float4 *)ps.hit + 1;   // {inst, t, packet, packet flags}
        ext.kind = (iter == 0) ? 3 : 0;
        // MIS rays of the previous vertex (closest hit, integrator.rs:215)
        TraceSub mis{};
        mis.queue = sc->q.mis; mis.count = &qc->mis; mis.scalar_tmax = INFINITY;
        mis.ray = (const float4 *)ps.mis; mis.ray_stride = PathSoA::kMisWords / 4;
        mis.out_hit = (float4 *)&ps.mis_prim(0); mis.out_hit_stride = PathSoA::kMisWords / 4;
        mis.out_t = rc.volpath ? &ps.mis_t(0) : nullptr; mis.out_t_stride = PathSoA::kMisWords;
        mis.kind = 1;
        const bool shells = rc.volpath && ps.ext != nullptr;   // the chains of VisibilityTester::tr / Scene::intersect_tr need every segment's full hit record
        if (shells) { mis.out_hit = (float4 *)ps.ext + 6; mis.out_hit_stride = PathSoA::kExtWords / 4; mis.out_hit2 = (float4 *)ps.ext + 7; mis.out_t = nullptr; }
        // shadow rays (any hit, light.rs:120-123). volpath: VisibilityTester::tr (light.rs:125-150) calls Scene::intersect, a closest-hit
        // query counted as one; without out_hit the primitive goes to out_word = nee.sh_prim, the slot `occluded` uses otherwise
        TraceSub sh{};
        sh.queue = sc->q.shadow; sh.count = &qc->shadow; sh.scalar_tmax = 1.0f - 0.0001f;
        sh.ray = (const float4 *)ps.nee; sh.ray_stride = PathSoA::kNeeWords / 4;
        sh.out_word = &ps.occluded(0); sh.out_word_stride = PathSoA::kNeeWords;
        sh.kind = 2; sh.any = rc.volpath ? 0u : 1u;
        if (shells) { sh.out_hit = (float4 *)ps.ext + 4; sh.out_hit_stride = PathSoA::kExtWords / 4; sh.out_hit2 = (float4 *)ps.ext + 5; }
        int st = PT_OK;
        if (n_mis + n_shadow == 0 || g_trace_split) {   // camera rays (nothing else to trace in the first iteration) / PT_TRACE_SPLIT=1: one launch per kind
            if (n_ext) {   // (a launch kind with no work is not a launch: the per-launch averages of bench.py / rocprofv3 count real dispatches)
                tj.sub[0] = ext;
                sc->begin(iter == 0 ? "extend_camera" : "extend", n_ext);
                st = launch_trace(sc, 0, tj, n_ext);
                sc->end();
                if (st) return st;
            }
            if (n_mis) {
                tj.sub[0] = mis; tj.head = &qc->head[1];
                sc->begin("extend_mis", n_mis);
                st = launch_trace(sc, 0, tj, n_mis);
                sc->end();
                if (st) return st;
            }
            if (n_shadow) {
                tj.sub[0] = sh; tj.head = &qc->head[2];
                sc->begin("shadow", n_shadow);
                st = launch_trace(sc, (int)sh.any, tj, n_shadow);
                sc->end();
                if (st) return st;
            }
        } else {   // the three ray kinds of this iteration in one launch: one tail of straggling rays instead of three
            tj.sub[0] = ext; tj.sub[1] = mis; tj.sub[2] = sh;
            sc->begin("trace", (uint64_t)n_ext + n_mis + n_shadow);
            st = launch_trace(sc, 2, tj, n_ext + n_mis + n_shadow);
            sc->end();
            if (st) return st;
        }
        if (n_ext && rc.volpath) {  // medium sampling (volpath.rs:98-105) + material-sorted shade queues + the medium-vertex queue
            sc->begin("route", n_ext); sc->set_kernel("k_medium_route");
            hipLaunchKernelGGL(k_medium_route, dim3(std::min<uint32_t>((n_ext + 255) / 256, (uint32_t)g_num_cus * 8u)), dim3(256), 0, sc->stream, sc->ds, rc, g_tabs, sc->ps,
                               (const uint32_t *)sc->q.ext[cur], (const uint32_t *)&qc->ext[cur], &qc->shade[cur][0],
                               sc->q.shade[cur][0], sc->q.shade[cur][1], sc->q.shade[cur][2], sc->q.shade[cur][3], sc->q.shade[cur][4], sc->q.shade[cur][5], &qc->error);
            sc->end();
        } else if (n_ext) {  // material-sorted shade queues
            sc->begin("route", n_ext); sc->set_kernel("k_route<6, 2048>");
            #ifndef PT_ROUTE_BLOCKS_PER_CU
#define PT_ROUTE_BLOCKS_PER_CU 3u   // what a CU's LDS holds of this kernel (six 8 KB staging queues per block)
#endif
            RouteJob rj{}; rj.slot_map = ~0ull; rj.error = &qc->error; rj.drop_cls = fin ? (uint32_t)kMissClass : ~0u;
            for (int c = 0; c < kNumClasses; ++c) if (c != kMediumClass && ((c == kMissClass && !fin) || (c != kMissClass && sc->class_used[c]))) {   // (fin: escaped rays are dropped here, the film kernel ends them)
                rj.slot_map = (rj.slot_map & ~(15ull << (4 * c))) | ((unsigned long long)rj.n_slots << (4 * c));
                rj.cls_of_slot[rj.n_slots] = (uint32_t)c; rj.buf[rj.n_slots] = sc->q.shade[cur][c]; rj.n_slots++;
            }
            const dim3 rgrid(std::min<uint32_t>((n_ext + 255) / 256, (uint32_t)g_num_cus * PT_ROUTE_BLOCKS_PER_CU));
            if (rj.n_slots <= 6) hipLaunchKernelGGL((k_route<6, 2048>), rgrid, dim3(256), 0, sc->stream, sc->ds, (const uint32_t *)sc->q.ext[cur], (const uint32_t *)&qc->ext[cur], sc->ps, &qc->shade[cur][0], rj);
            else { sc->set_kernel("k_route<12, 1024>"); hipLaunchKernelGGL((k_route<12, 1024>), rgrid, dim3(256), 0, sc->stream, sc->ds, (const uint32_t *)sc->q.ext[cur], (const uint32_t *)&qc->ext[cur], sc->ps, &qc->shade[cur][0], rj); }
            sc->end();
        }
        hipLaunchKernelGGL(k_reset, dim3(1), dim3(64), 0, sc->stream, qc, 2u, cur);
        if (grid.cell_ptr) {   // first-touch voxels: the vertices of every shade class name theirs, then the new ones are computed
            const uint32_t upper0 = n_ext + n_resolve;
            for (int c = 0; c < kNumClasses; ++c) {
                if (c == kMissClass || !class_has_queue(sc, c)) continue;
                if ((st = lazy_light_touch(sc, rc, grid, sc->q.shade[cur][c], &qc->shade[cur][c], upper0 + n_stage_b, c == kMediumClass ? 1u : 0u))) return st;
            }
            if ((st = lazy_light_fill(sc, grid))) return st;
        }
        if (n_probe) {  // subsurface probe chains (bssrdf.rs:367-402): each lane of k_trace<.., PROBE> walks a whole chain, then k_bssrdf
            TraceSub pr{};
            pr.queue = sc->q.probe[cur]; pr.count = &qc->probe[cur]; pr.scalar_tmax = 1.0f - 0.0001f;
            pr.ray = (const float4 *)ps.ray; pr.ray_stride = PathSoA::kRayWords / 4;
            pr.out_hit = (float4 *)ps.hit; pr.out_hit_stride = PathSoA::kHitWords / 4; pr.out_hit2 = (float4 *)ps.hit + 1;
            pr.kind = 4;
            tj.sub[0] = pr; tj.head = &qc->head[3]; tj.bs = sc->bs; tj.ring = sc->probe_ring;
            sc->begin("extend_probe", n_probe);
            st = launch_trace(sc, 0, tj, n_probe, true);
            sc->end();
            if (st) return st;
            if (grid.cell_ptr) {   // the chains' exit points look their voxels up in k_bssrdf
                if ((st = lazy_light_touch(sc, rc, grid, sc->q.probe[cur], &qc->probe[cur], n_probe, 2u))) return st;
                if ((st = lazy_light_fill(sc, grid))) return st;
            }
            BssrdfJob bj{};
            bj.queue = sc->q.probe[cur]; bj.count = &qc->probe[cur];
            bj.self_next = sc->q.shade[1 - cur][kSpecClass]; bj.self_next_count = &qc->shade[1 - cur][kSpecClass];   // (volpath has no specular-only class: its queue serves the waiting exit-point vertices)
            bj.ext_next = sc->q.ext[1 - cur]; bj.ext_next_count = &qc->ext[1 - cur];
            if (!fin) { bj.shade_next0 = sc->q.shade[1 - cur][kMissClass]; bj.shade_next0_count = &qc->shade[1 - cur][kMissClass]; }
            bj.shadow = sc->q.shadow; bj.shadow_count = &qc->shadow; bj.mis = sc->q.mis; bj.mis_count = &qc->mis;
            bj.error = &qc->error; bj.counters = sc->dc; bj.bs = sc->bs;
            const uint32_t blocks = std::min<uint32_t>((n_probe + 255) / 256, (uint32_t)g_num_cus * PT_SHADE_BLOCKS_PER_CU);
            sc->begin("bssrdf", n_probe);
            const bool bsph = sc->ds.n_spheres > 0 || sc->ds.n_instances > 0;
            sc->set_kernel(rc.volpath ? "k_bssrdf<true, true>" : bsph ? "k_bssrdf<true, false>" : "k_bssrdf<false, false>");
            if (rc.volpath) hipLaunchKernelGGL((k_bssrdf<true, true>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, rc, g_tabs, grid, sc->ps, bj);
            else if (bsph) hipLaunchKernelGGL((k_bssrdf<true, false>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, rc, g_tabs, grid, sc->ps, bj);
            else hipLaunchKernelGGL((k_bssrdf<false, false>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, rc, g_tabs, grid, sc->ps, bj);
            sc->end();
        }
        uint32_t class_n[kNumClasses];
        const uint32_t upper = n_ext + n_resolve + n_stage_b;
        if (rp_profile_exact) {  // exact per-class item counts for the statistics (costs one extra sync per iteration)
            QCounters h2;
            HIP_TRY(hipMemcpyAsync(&h2, qc, sizeof h2, hipMemcpyDeviceToHost, sc->stream));
            HIP_TRY(hipStreamSynchronize(sc->stream));
            for (int c = 0; c < kNumClasses; ++c) class_n[c] = h2.shade[cur][c];
        } else for (int c = 0; c < kNumClasses; ++c) class_n[c] = upper;
        for (int c = 0; c < kNumClasses; ++c) {
            bool used = sc->class_used[c] || (c == 1 && rc.volpath && sc->class_used[kSpecClass]);   // (the volumetric router folds class 6 into class 1)
            if (rc.volpath) { if (c > kSpecClass) used = false; else for (int k = kSpecClass + 1; k < kNumClasses; ++k) if (sc->class_used[k] && class_general((uint32_t)k) == (uint32_t)c) used = true; }   // (... and the lobe-set classes into their lobe-count class)
            if (c == kSpecClass && rc.volpath) {   // exit-point vertices of subsurface chains in stage B (k_bssrdf put them here an iteration ago)
                if (!sc->has_bssrdf || !(sc->ds.has_grid || sc->ds.has_shells) || class_n[c] == 0) continue;
                BssrdfJob bj{};
                bj.queue = sc->q.shade[cur][c]; bj.count = &qc->shade[cur][c];
                bj.ext_next = sc->q.ext[1 - cur]; bj.ext_next_count = &qc->ext[1 - cur];
                bj.shade_next0 = sc->q.shade[1 - cur][kMissClass]; bj.shade_next0_count = &qc->shade[1 - cur][kMissClass];
                bj.shadow = sc->q.shadow; bj.shadow_count = &qc->shadow; bj.mis = sc->q.mis; bj.mis_count = &qc->mis;
                bj.error = &qc->error; bj.counters = sc->dc; bj.bs = sc->bs;
                bj.self_next = sc->q.shade[1 - cur][c]; bj.self_next_count = &qc->shade[1 - cur][c]; bj.stage_b = 1u;
                sc->begin("bssrdf_stage_b", rp_profile_exact ? class_n[c] : 0);
                sc->set_kernel("k_bssrdf<true, true>");
                hipLaunchKernelGGL((k_bssrdf<true, true>), dim3(std::min<uint32_t>((class_n[c] + 255) / 256, (uint32_t)g_num_cus * PT_SHADE_BLOCKS_PER_CU)), dim3(256), 0, sc->stream, sc->ds, rc, g_tabs, grid, sc->ps, bj);
                sc->end();
                continue;
            }
            if (!used || class_n[c] == 0 || (c == kMissClass && fin)) continue;
            ShadeJob sj{};
            sj.queue = sc->q.shade[cur][c]; sj.count = &qc->shade[cur][c];
            sj.ext_next = sc->q.ext[1 - cur]; sj.ext_next_count = &qc->ext[1 - cur];
            if (!fin) { sj.shade_next0 = sc->q.shade[1 - cur][kMissClass]; sj.shade_next0_count = &qc->shade[1 - cur][kMissClass]; }
            sj.shadow = sc->q.shadow; sj.shadow_count = &qc->shadow; sj.mis = sc->q.mis; sj.mis_count = &qc->mis;
            sj.error = &qc->error; sj.counters = sc->dc; sj.cls = (uint32_t)c;
            sj.self_next = sc->q.shade[1 - cur][c]; sj.self_next_count = &qc->shade[1 - cur][c];
            if ((c == 3 || c == kSssClass) && sc->has_bssrdf) { sj.probe_next = sc->q.probe[1 - cur]; sj.probe_next_count = &qc->probe[1 - cur]; sj.bs = sc->bs; }
            sc->begin(shade_names[c], rp_profile_exact ? class_n[c] : 0);
            if (c == kMediumClass) {
                const uint32_t blocks = std::min<uint32_t>((class_n[c] + 255) / 256, (uint32_t)g_num_cus * 8u);
                sc->set_kernel("k_shade_medium");
                hipLaunchKernelGGL(k_shade_medium, dim3(blocks), dim3(256), 0, sc->stream, sc->ds, rc, g_tabs, grid, sc->ps, sj);
            }
            else if (c == kMissClass) {
                #ifndef PT_MISS_BLOCKS_PER_CU
#define PT_MISS_BLOCKS_PER_CU 16u   // experiment hook
#endif
                const uint32_t blocks = std::min<uint32_t>((class_n[c] + 255) / 256, (uint32_t)g_num_cus * PT_MISS_BLOCKS_PER_CU);
                sc->set_kernel(rc.volpath ? "k_shade_miss<true, true>" : (sc->ds.n_spheres > 0 || sc->ds.n_instances > 0) ? "k_shade_miss<true, false>" : "k_shade_miss<false, false>");
                if (rc.volpath) hipLaunchKernelGGL((k_shade_miss<true, true>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, rc, sc->ps, sj);
                else if (sc->ds.n_spheres > 0 || sc->ds.n_instances > 0) hipLaunchKernelGGL((k_shade_miss<true, false>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, rc, sc->ps, sj);
                else hipLaunchKernelGGL((k_shade_miss<false, false>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, rc, sc->ps, sj);
            }
            else if (c == 0) launch_shade<1, 1>(sc, rc, grid, sj, class_n[c]);
            else if (c == kSpecClass) launch_shade<1, 2>(sc, rc, grid, sj, class_n[c]);
            else if (c == 1) launch_shade<1>(sc, rc, grid, sj, class_n[c]);
            else if (c == 2) launch_shade<2>(sc, rc, grid, sj, class_n[c]);
            else if (c == kMetalClass) launch_shade<1, 3>(sc, rc, grid, sj, class_n[c]);     // (the lobe-set classes exist in untextured scenes only and never reach this loop under volpath)
            else if (c == kPlasticClass) launch_shade<2, 4>(sc, rc, grid, sj, class_n[c]);
            else if (c == kUberClass) launch_shade<5, 5>(sc, rc, grid, sj, class_n[c]);
            else if (c == kSssClass) launch_shade<1, 6>(sc, rc, grid, sj, class_n[c]);
            else launch_shade<5>(sc, rc, grid, sj, class_n[c]);
            sc->end();
        }
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_reset, dim3(1), dim3(64), 0, sc->stream, qc, 8u, cur);
        cur = 1 - cur;
    }
    sc->begin("film", total);
    const bool fsph = sc->ds.n_spheres > 0 || sc->ds.n_instances > 0;
    sc->set_kernel(fin ? (fsph ? "k_film_final<true>" : "k_film_final<false>") : "k_film");
    const dim3 fgrid((unsigned)(((size_t)rc.n_pix_slots * kFilmLanes + 255) / 256));   // kFilmLanes threads per pixel slot (kern_film.h)
    if (fin && fsph) hipLaunchKernelGGL((k_film_final<true>), fgrid, dim3(256), 0, sc->stream, sc->ds, rc, sc->ps, sc->d_filter, sc->film_rgbw, sc->dc);
    else if (fin) hipLaunchKernelGGL((k_film_final<false>), fgrid, dim3(256), 0, sc->stream, sc->ds, rc, sc->ps, sc->d_filter, sc->film_rgbw, sc->dc);
    else hipLaunchKernelGGL(k_film, fgrid, dim3(256), 0, sc->stream, rc, sc->ps, sc->d_filter, sc->film_rgbw, sc->dc);
    sc->end();
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

void read_counters(pt_scene *sc) {
    DevCounters d;
    hipMemcpy(&d, sc->dc, sizeof d, hipMemcpyDeviceToHost);
    PtCounters &c = sc->counters;
    std::memset(&c, 0, sizeof c);
    c.camera_rays = d.camera_rays; c.intersect_tests = d.intersect_tests; c.shadow_tests = d.shadow_tests;
    c.bvh_nodes_visited = d.nodes; c.triangle_tests = d.tri_tests; c.sphere_tests = d.sphere_tests;
    c.zero_radiance_paths_num = d.zero_num; c.zero_radiance_paths_den = d.zero_den;
    for (int i = 0; i < 16; ++i) c.path_length_hist[i] = d.path_len[i];
    c.sanitized_nan = d.san_nan; c.sanitized_negative = d.san_neg; c.sanitized_infinite = d.san_inf;
    c.film_splats = d.splats; c.wavefront_stages = d.stages; c.reference_asserts = d.ref_asserts;
    static const char *sn[kNumClasses] = {"shade_matte", "shade_1lobe", "shade_2lobe", "shade_manylobe", "shade_miss", "shade_medium", "shade_specular", "shade_metal", "shade_plastic", "shade_uber", "shade_sss"};
    for (int k = 0; k < kNumClasses; ++k) for (auto &s : sc->stats) if (s.name == sn[k]) { s.items = d.shade_items[k]; s.nodes = d.shade_bytes[k]; }
    static const char *kn[5] = {"extend", "extend_mis", "shadow", "extend_camera", "extend_probe"};
    for (int k = 0; k < 5; ++k) for (auto &s : sc->stats) if (s.name == kn[k]) { s.nodes = d.k_nodes[k]; s.tris = d.k_tris[k]; if (k == 4) s.items = d.k_rays[k]; }   // probe chains: items = segments traced
    for (auto &s : sc->stats) if (s.name == "trace") { s.nodes = d.k_nodes[0] + d.k_nodes[1] + d.k_nodes[2]; s.tris = d.k_tris[0] + d.k_tris[1] + d.k_tris[2]; }   // the mixed launches: all three kinds
    if (!g_trace_split) {   // what the mixed launches did per ray kind (no time of their own: launches = 0)
        for (int k = 0; k < 3; ++k) if (d.k_rays[k]) { bool have = false; for (auto &s : sc->stats) have = have || s.name == kn[k];
            if (!have) { Stat s2{std::string("trace:") + kn[k]}; s2.items = d.k_rays[k]; s2.nodes = d.k_nodes[k]; s2.tris = d.k_tris[k]; sc->stats.push_back(s2); } }
    }
    for (auto &s : sc->stats) if (s.name == "bssrdf") { s.items = d.bss_items; s.nodes = d.bss_bytes; }
#ifdef PT_TRACE_UTIL
    fprintf(stderr, "[trace-util] kernel saw leaf_quorum = %llu, refill_min = %llu (last launch)\n", d.dbg[0], d.dbg[1]);
    for (int k = 0; k < 4; ++k) if (d.tail[5 + 2 * k])
        fprintf(stderr, "[trace-util] %-14s wave slots busy %.1f %% of launch span x resident waves (the rest: launch ramp + tail after the queue drained)\n", kn[k], 100.0 * (double)d.tail[4 + 2 * k] / (double)d.tail[5 + 2 * k]);
    if (d.tail[2]) fprintf(stderr, "[trace-util] all trace launches: transform step %.3e wave iterations, %.1f %% lanes active, %.1f %% of the waves' cycles; record step (fetch + node / leaf + pop) %.1f %% of the cycles\n",
                           (double)d.tail[12], d.tail[12] ? 100.0 * (double)d.tail[13] / (64.0 * (double)d.tail[12]) : 0.0, 100.0 * (double)d.tail[14] / (double)d.tail[2], 100.0 * (double)d.tail[15] / (double)d.tail[2]);
    if (d.tail[2]) fprintf(stderr, "[trace-util] instance entries tried %.4e, turned away by the object's root test %.4e (%.1f %%), left with a hit %.4e (%.1f %%); stack entries written beyond the LDS ones %.4e\n",
                           (double)d.util2[4], (double)d.util2[5], d.util2[4] ? 100.0 * (double)d.util2[5] / (double)d.util2[4] : 0.0, (double)d.util2[6], d.util2[4] ? 100.0 * (double)d.util2[6] / (double)d.util2[4] : 0.0, (double)d.util2[7]);
    if (d.tail[2]) fprintf(stderr, "[trace-util] record step by part, %% of the waves' cycles: loads issued + waited for %.1f, node branch %.1f, leaf branch %.1f, pops %.1f\n",
                           100.0 * (double)d.util2[0] / (double)d.tail[2], 100.0 * (double)d.util2[1] / (double)d.tail[2], 100.0 * (double)d.util2[2] / (double)d.tail[2], 100.0 * (double)d.util2[3] / (double)d.tail[2]);
    for (int k = 0; k < 4; ++k)
        fprintf(stderr, "[trace-util] %-14s node phase: %.3e wave iterations, %.1f %% lanes active; leaf phase: %.3e iterations, %.1f %% lanes active\n", kn[k], (double)d.regions[4 * k],
                d.regions[4 * k] ? 100.0 * (double)d.regions[4 * k + 1] / (64.0 * (double)d.regions[4 * k]) : 0.0, (double)d.regions[4 * k + 2], d.regions[4 * k + 2] ? 100.0 * (double)d.regions[4 * k + 3] / (64.0 * (double)d.regions[4 * k + 2]) : 0.0);
#endif
#ifdef PT_REGION_PROFILE
    {
        static const char *rn[16] = {"0 loop/queue read", "1 resolve", "2 resolve env le", "3 load ray/hit + fill_hit + Le", "4 sobol window", "5 light choice",
                                     "6 light sample_li", "7 bsdf f/pdf + shadow ray", "8 MIS bsdf sample + store", "9 MIS light pdf_li", "10 bsdf build", "11 continuation sample + RR",
                                     "12 state write-back", "13 queue push/flush", "14 tail", "15 prologue"};
        unsigned long long tot = 0; for (int i = 0; i < 16; ++i) tot += d.regions[i];
        for (int i = 0; i < 16; ++i) fprintf(stderr, "[region] %-34s %6.2f %%  %.3e cycles\n", rn[i], tot ? 100.0 * (double)d.regions[i] / (double)tot : 0.0, (double)d.regions[i]);
    }
#endif
}

}  // namespace pth

extern "C" {

// What pt_render checks before it touches the device and the pass size it then uses -- ONE function for pt_render and pt_pass_size, so that the size pt_pass_size
// reports is the size of a render that would be accepted (the two had drifted apart). Fills rc (incl. n_tile_slots / n_pix_slots); *S = samples per pixel
// per wavefront pass (0 when this rank owns no tile).
static int render_geometry(pt_scene *sc, const PtRenderParams *rp, RenderConst &rc, uint32_t *S_out) {
    if (rp->spp == 0) return fail(PT_ERR_INVALID_ARG, "spp must be > 0");
    if (!(rp->filter_radius[0] > 0.0f) || !(rp->filter_radius[1] > 0.0f)) return fail(PT_ERR_INVALID_ARG, "filter radius must be > 0");
    if (rp->tile_world > 1 && rp->tile_rank >= rp->tile_world) return fail(PT_ERR_INVALID_ARG, "tile_rank >= tile_world");
    if (sc->device != g_device) { int bst = bind_device(sc->device); if (bst) return bst; }
    fill_render_const(rp, rc);
    if (rc.film_w == 0 || rc.film_h == 0 || rc.ntx == 0 || rc.nty == 0) return fail(PT_ERR_INVALID_ARG, "empty film or sample bounds");
    if (rc.sobol.log2_resolution > 25) return fail(PT_ERR_INVALID_ARG, "sample bounds exceed the 2^25 Sobol' pixel grid");
    if (rp->max_depth > 254) return fail(PT_ERR_INVALID_ARG, "maxdepth must be <= 254 (the bounce count of a path is kept in 8 bits)");
    if (rc.volpath) {   // VolPathIntegrator (volpath.rs): what this back end takes
        if (sc->has_bssrdf && sc->ds.n_media >= 0xffffu) return fail(PT_ERR_UNSUPPORTED, "volpath: subsurface materials with more than 65534 media");
        if (rp->camera_medium != PT_NONE && rp->camera_medium >= sc->ds.n_media) return fail(PT_ERR_INVALID_ARG, "camera_medium out of range");
    }
    const uint32_t ntiles = rc.ntx * rc.nty;
    rc.n_tile_slots = rc.tile_rank < ntiles ? (ntiles - rc.tile_rank + rc.tile_world - 1) / rc.tile_world : 0;
    rc.n_pix_slots = rc.n_tile_slots * 256u;
    uint32_t S = 0;
    if (rc.n_pix_slots > 0) {
        S = rp->spp_per_pass;
        if (S == 0) S = choose_pass_size(sc, rc.n_pix_slots, rp->spp, 1, rc.volpath != 0);
        S = std::min(S, rp->spp);
        if ((size_t)rc.n_pix_slots * S > ((size_t)1 << 31)) return fail(PT_ERR_INVALID_ARG, "pass too large: pixel slots x samples per pass > 2^31 paths (lower spp_per_pass)");
    }
    *S_out = S;
    return PT_OK;
}

int pt_pass_size(pt_scene *sc, const PtRenderParams *rp, uint32_t *spp_per_pass) {
    if (!sc || !rp || !spp_per_pass) return fail(PT_ERR_INVALID_ARG, "null argument");
    RenderConst rc; uint32_t S = 0;
    if (int st = render_geometry(sc, rp, rc, &S)) return st;
    *spp_per_pass = S ? S : rp->spp;   // (a rank that owns no tile renders nothing: any size)
    return PT_OK;
}

int pt_render(pt_scene *sc, const PtRenderParams *rp, float *film_xyzw, int film_is_device) {
    if (!sc || !rp || !film_xyzw) return fail(PT_ERR_INVALID_ARG, "null argument");
    RenderConst rc; uint32_t S = 0;
    if (int gst = render_geometry(sc, rp, rc, &S)) return gst;
    g_test_max_iterations = 0;
    if (const char *e = getenv("PT_TEST_MAX_ITERATIONS")) { const int v = atoi(e); if (v > 0) g_test_max_iterations = v; }
    const size_t film_px = (size_t)rc.film_w * rc.film_h;
    sc->profile = rp->profile != 0;
    sc->drop_timings();
    sc->stats.clear();
    int st = PT_OK;
    if (rc.n_pix_slots > 0) {
        if ((st = ensure_workspace(sc, (size_t)rc.n_pix_slots * S, film_px))) return st;
        if (rc.volpath && sc->has_null_material && sc->ext_capacity < sc->capacity) {   // the shells' chain state (PathSoA::ext)
            if (sc->ext_slab) { hipFree(sc->ext_slab); sc->ext_slab = nullptr; sc->ext_capacity = 0; }
            if (hipMalloc(&sc->ext_slab, sc->capacity * (size_t)PathSoA::kExtWords * 4) != hipSuccess) { (void)hipGetLastError(); return fail(PT_ERR_OUT_OF_MEMORY, "volpath: chain state of material-less shells"); }
            sc->ext_capacity = sc->capacity;
        }
        sc->ps.ext = (rc.volpath && sc->has_null_material) ? (float *)sc->ext_slab : nullptr;
        int eff;
        if ((st = ensure_light_grid(sc, (int)rp->light_strategy, eff))) return st;
        if (sc->grid[eff].cell_ptr) {   // first-touch voxels: a render that failed half way may have named voxels it never computed -- start from a clean request list
            HIP_TRY(hipMemsetAsync(sc->lazy.req_flag, 0, sc->lazy.ncell * 4, sc->stream));
            HIP_TRY(hipMemsetAsync(sc->lazy.req_count, 0, 8, sc->stream));
        }
        HIP_TRY(hipMemcpyAsync(sc->d_filter, rp->filter_table, 256 * 4, hipMemcpyHostToDevice, sc->stream));
        HIP_TRY(hipMemsetAsync(sc->film_rgbw, 0, film_px * 16, sc->stream));
        HIP_TRY(hipMemsetAsync(sc->dc, 0, sizeof(DevCounters), sc->stream));
        HIP_TRY(hipMemsetAsync(sc->qc, 0, sizeof(QCounters), sc->stream));
        for (uint32_t s0 = 0; s0 < rp->spp; s0 += S) {
            rc.s_begin = s0; rc.s_count = std::min(S, rp->spp - s0);
            if ((st = run_pass(sc, rc, sc->grid[eff], rp->profile >= 2))) return st;
        }
        if (sc->grid[eff].cell_ptr) {   // a vertex that looked up a voxel nobody had computed: cannot happen (k_light_touch names every voxel first)
            uint32_t missing = 0;
            HIP_TRY(hipMemcpyAsync(&missing, sc->lazy.missing, 4, hipMemcpyDeviceToHost, sc->stream));
            HIP_TRY(hipStreamSynchronize(sc->stream));
            if (missing) { HIP_TRY(hipMemset(sc->lazy.missing, 0, 4)); return fail(PT_ERR_HIP, "internal: " + std::to_string(missing) + " light-distribution lookups hit a voxel that had not been computed"); }
        }
        float *dst = film_xyzw, *tmp = nullptr; DevTmp film_tmp;
        std::vector<float> host;
        if (!film_is_device) {
            HIP_TRY(film_tmp.alloc(&tmp, film_px * 16));
            HIP_TRY(hipMemsetAsync(tmp, 0, film_px * 16, sc->stream));
            dst = tmp;
        }
        sc->begin("film_finish", film_px);
        sc->set_kernel("k_film_finish");
        hipLaunchKernelGGL(k_film_finish, dim3((unsigned)((film_px + 255) / 256)), dim3(256), 0, sc->stream, sc->film_rgbw, dst, (uint32_t)film_px);
        sc->end();
        HIP_TRY(hipStreamSynchronize(sc->stream));
        if (!film_is_device) {
            host.resize(film_px * 4);
            HIP_TRY(hipMemcpy(host.data(), tmp, film_px * 16, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < film_px * 4; ++i) film_xyzw[i] += host[i];
        }
        sc->resolve_timings();
        read_counters(sc);
    }
    return PT_OK;
}

int pt_film_resolve(const float *xyzw, uint32_t n, float scale, float *rgb) {  // film.rs:217-258 (host arithmetic)
    if (!xyzw || !rgb) return fail(PT_ERR_INVALID_ARG, "null argument");
    for (uint32_t i = 0; i < n; ++i) {
        float c[3]; xyz_to_rgb(xyzw + 4 * (size_t)i, c);
        float w = xyzw[4 * (size_t)i + 3];
        if (w != 0.0f) { float inv = 1.0f / w; for (int k = 0; k < 3; ++k) c[k] = std::fmax(c[k] * inv, 0.0f); }
        for (int k = 0; k < 3; ++k) rgb[3 * (size_t)i + k] = c[k] * scale;
    }
    return PT_OK;
}

int pt_get_counters(const pt_scene *sc, PtCounters *out) {
    if (!sc || !out) return fail(PT_ERR_INVALID_ARG, "null argument");
    *out = sc->counters;
    return PT_OK;
}
int pt_get_kernel_stats(const pt_scene *sc, PtKernelStat *out, uint32_t max_entries, uint32_t *n_out) {
    if (!sc || !out || !n_out) return fail(PT_ERR_INVALID_ARG, "null argument");
    uint32_t n = (uint32_t)std::min<size_t>(max_entries, sc->stats.size());
    for (uint32_t i = 0; i < n; ++i) {
        std::memset(&out[i], 0, sizeof out[i]);
        std::snprintf(out[i].name, sizeof out[i].name, "%s", sc->stats[i].name.c_str());
        std::snprintf(out[i].kernel, sizeof out[i].kernel, "%s", sc->stats[i].kernel.c_str());
        out[i].launches = sc->stats[i].launches; out[i].total_ms = sc->stats[i].ms; out[i].items = sc->stats[i].items;
        out[i].bvh_nodes = sc->stats[i].nodes; out[i].triangle_tests = sc->stats[i].tris;
    }
    *n_out = n;
    return PT_OK;
}

}  // extern "C"
