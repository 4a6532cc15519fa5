// parity_api.hip -- entry points that expose single stages of the path to the parity tests (host_common.h has the map).
#include "host_common.h"

extern "C" {

static int trace_api(pt_scene *sc, bool any, uint32_t n, const float *o, const float *d, const float *tmax, uint32_t *prim, float *t, float *b, uint8_t *hit) {
    DevTmp scratch;
    if (!sc || !o || !d || !tmax) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (n == 0) return PT_OK;
    if (sc->device != g_device) { int bst = bind_device(sc->device); if (bst) return bst; }
    int st = ensure_workspace(sc, 0, 0);
    if (st) return st;
    std::vector<float> recs(8 * (size_t)n, 0.0f);   // 32-byte ray records {o.xyz, d.x} {d.y, d.z, t_max, -}
    for (uint32_t i = 0; i < n; ++i) {
        float *r = recs.data() + 8 * (size_t)i;
        r[0] = o[3 * (size_t)i]; r[1] = o[3 * (size_t)i + 1]; r[2] = o[3 * (size_t)i + 2];
        r[3] = d[3 * (size_t)i]; r[4] = d[3 * (size_t)i + 1]; r[5] = d[3 * (size_t)i + 2]; r[6] = tmax[i];
    }
    float *din = nullptr, *dout = nullptr, *dt = nullptr; uint32_t *docc = nullptr; uint32_t *dcount = nullptr;
    HIP_TRY(scratch.alloc(&din, recs.size() * 4));
    HIP_TRY(scratch.alloc(&dout, 4 * (size_t)n * 4));
    HIP_TRY(scratch.alloc(&dt, (size_t)n * 4));
    HIP_TRY(scratch.alloc(&docc, (size_t)n * 4));
    HIP_TRY(scratch.alloc(&dcount, 4));
    HIP_TRY(hipMemcpy(din, recs.data(), recs.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dcount, &n, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(sc->dc, 0, sizeof(DevCounters), sc->stream));
    HIP_TRY(hipMemsetAsync(sc->qc, 0, sizeof(QCounters), sc->stream));
    TraceJob tj{};
    TraceSub &ts = tj.sub[0];
    ts.queue = nullptr; ts.count = dcount; tj.head = &sc->qc->head[0];
    ts.ray = (const float4 *)din; ts.ray_stride = 2; ts.per_ray_tmax = 1;
    ts.out_hit = (float4 *)dout; ts.out_hit_stride = 1; ts.out_t = dt; ts.out_t_stride = 1;
    ts.out_word = docc; ts.out_word_stride = 1; ts.out_hit2 = nullptr;
    tj.spill = sc->spill; tj.error = &sc->qc->error; tj.counters = sc->dc;
    sc->profile = true; sc->drop_timings(); sc->stats.clear();
    ts.kind = any ? 2 : 0; ts.any = any ? 1u : 0u;
    sc->begin(any ? "trace_any_api" : "trace_closest_api", n);
    st = launch_trace(sc, any ? 1 : 0, tj, n);
    sc->end();
    if (st) return st;
    HIP_TRY(hipStreamSynchronize(sc->stream));
    sc->resolve_timings();
    QCounters h;
    HIP_TRY(hipMemcpy(&h, sc->qc, sizeof h, hipMemcpyDeviceToHost));
    if (any) {
        std::vector<uint32_t> occ(n);
        HIP_TRY(hipMemcpy(occ.data(), docc, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; ++i) hit[i] = occ[i] ? 1 : 0;
    } else {
        std::vector<float> res(4 * (size_t)n);
        HIP_TRY(hipMemcpy(res.data(), dout, res.size() * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(t, dt, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; ++i) { std::memcpy(&prim[i], &res[4 * (size_t)i], 4); for (int k = 0; k < 3; ++k) b[3 * (size_t)i + k] = res[4 * (size_t)i + 1 + k]; }
    }
    read_counters(sc);
    if (h.error) return fail((int)h.error, device_error_text(h.error));
    return PT_OK;
}

int pt_trace_closest(pt_scene *sc, uint32_t n, const float *o, const float *d, const float *tmax, uint32_t *prim, float *t, float *b) {
    if (!prim || !t || !b) return fail(PT_ERR_INVALID_ARG, "null output");
    return trace_api(sc, false, n, o, d, tmax, prim, t, b, nullptr);
}
int pt_trace_any(pt_scene *sc, uint32_t n, const float *o, const float *d, const float *tmax, uint8_t *hit) {
    if (!hit) return fail(PT_ERR_INVALID_ARG, "null output");
    return trace_api(sc, true, n, o, d, tmax, nullptr, nullptr, nullptr, hit);
}

int pt_sobol_samples(const int32_t sb[4], uint32_t n, const int32_t *pixel_xy, const uint32_t *sample_num, uint32_t n_dims, float *out, uint64_t *out_index) {
    DevTmp scratch;
    if (!sb || !pixel_xy || !sample_num || !out) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (n_dims > 1024) return fail(PT_ERR_SOBOL_DIMENSIONS, "SobolSampler can only sample up to 1024 dimensions");
    int st = ensure_device();
    if (st || n == 0) return st;
    PtRenderParams rp{}; std::memcpy(rp.sample_bounds, sb, 16);
    RenderConst rc; fill_render_const(&rp, rc);
    int32_t *dxy; uint32_t *dsn; float *dout; uint64_t *didx;
    HIP_TRY(scratch.alloc(&dxy, (size_t)n * 8)); HIP_TRY(scratch.alloc(&dsn, (size_t)n * 4));
    HIP_TRY(scratch.alloc(&dout, (size_t)n * n_dims * 4 + 4)); HIP_TRY(scratch.alloc(&didx, (size_t)n * 8));
    HIP_TRY(hipMemcpy(dxy, pixel_xy, (size_t)n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dsn, sample_num, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sobol_samples, dim3((n + 255) / 256), dim3(256), 0, 0, g_tabs, rc.sobol, n, dxy, dsn, n_dims, dout, didx);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout, (size_t)n * n_dims * 4, hipMemcpyDeviceToHost));
    if (out_index) HIP_TRY(hipMemcpy(out_index, didx, (size_t)n * 8, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_halton_samples(const int32_t sb[4], uint32_t sample_at_pixel_center, uint32_t n, const int32_t *pixel_xy, const uint32_t *sample_num, uint32_t n_dims, float *out, uint64_t *out_index) {
    DevTmp scratch;
    if (!sb || !pixel_xy || !sample_num || !out) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (n_dims > kHaltonMaxDims) return fail(PT_ERR_SOBOL_DIMENSIONS, "HaltonSampler can only sample 1000 dimensions");
    int st = ensure_device();
    if (st || n == 0) return st;
    PtRenderParams rp{}; std::memcpy(rp.sample_bounds, sb, 16); rp.sampler_type = PT_SAMPLER_HALTON; rp.sample_at_pixel_center = sample_at_pixel_center;
    RenderConst rc; fill_render_const(&rp, rc);
    int32_t *dxy; uint32_t *dsn; float *dout; uint64_t *didx;
    HIP_TRY(scratch.alloc(&dxy, (size_t)n * 8)); HIP_TRY(scratch.alloc(&dsn, (size_t)n * 4));
    HIP_TRY(scratch.alloc(&dout, (size_t)n * n_dims * 4 + 4)); HIP_TRY(scratch.alloc(&didx, (size_t)n * 8));
    HIP_TRY(hipMemcpy(dxy, pixel_xy, (size_t)n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dsn, sample_num, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_halton_samples, dim3((n + 255) / 256), dim3(256), 0, 0, g_tabs, rc.halton, n, dxy, dsn, n_dims, dout, didx);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout, (size_t)n * n_dims * 4, hipMemcpyDeviceToHost));
    if (out_index) HIP_TRY(hipMemcpy(out_index, didx, (size_t)n * 8, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_camera_rays(const PtRenderParams *rp, uint32_t n, const float *cs, float *out_o, float *out_d) {
    DevTmp scratch;
    if (!rp || !cs || !out_o || !out_d) return fail(PT_ERR_INVALID_ARG, "null argument");
    int st = ensure_device();
    if (st || n == 0) return st;
    RenderConst rc; fill_render_const(rp, rc);
    float *dcs, *dout;
    HIP_TRY(scratch.alloc(&dcs, (size_t)n * 20)); HIP_TRY(scratch.alloc(&dout, (size_t)n * 24));
    HIP_TRY(hipMemcpy(dcs, cs, (size_t)n * 20, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_camera_rays, dim3((n + 255) / 256), dim3(256), 0, 0, rc, n, dcs, dout, dout + 3 * (size_t)n);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_o, dout, (size_t)n * 12, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_d, dout + 3 * (size_t)n, (size_t)n * 12, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_dist1d_sample(const float *func, uint32_t n, int discrete, uint32_t n_u, const float *u, float *out_x, float *out_pdf, int32_t *out_offset) {
    DevTmp scratch;
    if (!func || !u || !out_x || !out_pdf || !out_offset) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (n == 0) return fail(PT_ERR_INVALID_ARG, "a Distribution1D needs at least one function value");
    int st = ensure_device();
    if (st || n_u == 0) return st;
    std::vector<float> f(func, func + n), cdf; float fint = 0.0f;
    dist1d(f, cdf, fint);   // Distribution1D::new (sampling.rs:12-34), the construction the environment map and the light distributions use
    float *dfunc, *dcdf, *du, *dx, *dpdf; int32_t *doff;
    HIP_TRY(scratch.alloc(&dfunc, (size_t)n * 4)); HIP_TRY(scratch.alloc(&dcdf, ((size_t)n + 1) * 4)); HIP_TRY(scratch.alloc(&du, (size_t)n_u * 4));
    HIP_TRY(scratch.alloc(&dx, (size_t)n_u * 4)); HIP_TRY(scratch.alloc(&dpdf, (size_t)n_u * 4)); HIP_TRY(scratch.alloc(&doff, (size_t)n_u * 4));
    HIP_TRY(hipMemcpy(dfunc, f.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dcdf, cdf.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(du, u, (size_t)n_u * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dist1d_sample, dim3((n_u + 255) / 256), dim3(256), 0, 0, dfunc, dcdf, fint, (int)n, discrete, n_u, du, dx, dpdf, doff);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_x, dx, (size_t)n_u * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_pdf, dpdf, (size_t)n_u * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_offset, doff, (size_t)n_u * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

}  // extern "C"
