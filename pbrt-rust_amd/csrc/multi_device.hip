// multi_device.hip -- pt_multi_*: one process, several devices (host_common.h has the map).
#include "host_common.h"

extern "C" {

// ---- one process, several devices ---------------------------------------------------------------------------------------
// The reference is ONE process that fans the 16x16 tiles out over its worker threads (integrator.rs:294-296) and merges the tiles
// into one Film (integrator.rs:392-396). Same shape here with GPUs as the workers: the scene is replicated on every listed device,
// replica i renders the tiles with tile % (world * n) == rank + i * world on its own host thread and stream, and the replicas'
// films are summed onto the first device (peer copies over xGMI + an add kernel) before they are added to the caller's film.
void pt_multi_tile_shard(uint32_t tile_rank, uint32_t tile_world, uint32_t replica, uint32_t n_replicas, uint32_t *rank_out, uint32_t *world_out) {
    const uint32_t w = tile_world ? tile_world : 1u, n = n_replicas ? n_replicas : 1u;
    *world_out = w * n; *rank_out = tile_rank + replica * w;    // t % (w n) == r + i w  =>  t % w == r : the caller's own shard, split n ways
}

int pt_multi_scene_create(const PtSceneDesc *desc, const int *device_ordinals, uint32_t n_devices, pt_multi_scene **out) {
    if (!desc || !device_ordinals || !out || n_devices == 0 || n_devices > (uint32_t)kMaxDevices) return fail(PT_ERR_INVALID_ARG, "pt_multi_scene_create: bad arguments");
    int st = ensure_device();
    if (st) return st;
    const int home = g_device;
    pt_multi_scene *ms = new pt_multi_scene();
    auto bail = [&](int code) { const std::string msg = g_error; pt_multi_scene_destroy(ms); bind_device(home); g_error = msg; return code; };
    PtSceneDesc d = *desc;
    using clk = std::chrono::steady_clock;
    const clk::time_point t_begin = clk::now();
    ms->sc.assign(n_devices, nullptr); ms->dev.assign(device_ordinals, device_ordinals + n_devices);
    ms->film.assign(n_devices, nullptr); ms->film_cap.assign(n_devices, 0); ms->stage.assign(n_devices, nullptr); ms->stage_cap.assign(n_devices, 0);
    ms->render_ms.assign(n_devices, 0); ms->copy_ms.assign(n_devices, 0); ms->create_ms.assign(n_devices, 0);
    // The first replica builds the top-level tree on the calling thread; the others ADOPT it (no second build) and are created concurrently, one host thread per replica bound
    // to its device -- as pt_multi_render runs them -- so that seven uploads of a 0.5 GB scene overlap instead of queueing behind each other.
    {
        if ((st = bind_device(device_ordinals[0]))) return bail(st);
        if ((st = pt_scene_create(&d, &ms->sc[0]))) return bail(st);
        ms->create_ms[0] = std::chrono::duration<double, std::milli>(clk::now() - t_begin).count();
        d.nodes = ms->sc[0]->nodes.data(); d.n_nodes = (uint32_t)ms->sc[0]->nodes.size(); d.ordered_prims = ms->sc[0]->ordered.data();
    }
    if (n_devices > 1) {
        std::vector<int> status(n_devices, PT_OK); std::vector<std::string> message(n_devices);
        std::vector<std::thread> workers;
        for (uint32_t i = 1; i < n_devices; ++i) workers.emplace_back([&, i]() {
            const clk::time_point t0 = clk::now();
            int s2 = bind_device(device_ordinals[i]);
            if (!s2) s2 = pt_scene_create(&d, &ms->sc[i]);
            if (s2) { status[i] = s2; message[i] = g_error; }   // (g_error is thread-local: handed back to the caller's thread below)
            ms->create_ms[i] = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        });
        for (auto &w : workers) w.join();
        for (uint32_t i = 1; i < n_devices; ++i) if (status[i]) { g_error = "replica " + std::to_string(i) + " (device " + std::to_string(device_ordinals[i]) + "): " + message[i]; return bail(status[i]); }
    }
    ms->create_wall_ms = std::chrono::duration<double, std::milli>(clk::now() - t_begin).count();
    // peer access first device <-> the others (the film merge copies device to device; without access the runtime stages the copies through the
    // host): the outcome per replica is kept and reported (pt_multi_get_peer_access), so that a run which fell back says so instead of just being slow
    ms->peer.assign(n_devices, PT_PEER_SAME_DEVICE);
    for (uint32_t i = 1; i < n_devices; ++i) {
        if (ms->dev[i] == ms->dev[0]) continue;
        auto enable = [&](int from, int to) {   // `from` may address memory of `to`
            int can = 0;
            if (bind_device(from) != PT_OK || hipDeviceCanAccessPeer(&can, from, to) != hipSuccess || !can) { (void)hipGetLastError(); return false; }
            const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
            (void)hipGetLastError();
            return e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
        };
        const bool a = enable(ms->dev[0], ms->dev[i]), b = enable(ms->dev[i], ms->dev[0]);
        ms->peer[i] = (a && b) ? PT_PEER_ENABLED : PT_PEER_STAGED;
    }
    if ((st = bind_device(home))) return bail(st);
    *out = ms;
    return PT_OK;
}

void pt_multi_scene_destroy(pt_multi_scene *ms) {
    if (!ms) return;
    const int home = g_device;
    if (!ms->dev.empty() && bind_device(ms->dev[0]) == PT_OK) for (float *p : ms->stage) if (p) hipFree(p);
    for (size_t i = 0; i < ms->sc.size(); ++i) {
        if (bind_device(ms->dev[i]) == PT_OK && ms->film[i]) hipFree(ms->film[i]);
        if (ms->sc[i]) pt_scene_destroy(ms->sc[i]);
    }
    if (home >= 0) bind_device(home);
    delete ms;
}

int pt_multi_render(pt_multi_scene *ms, const PtRenderParams *rp, float *film_xyzw, int film_is_device) {
    if (!ms || !rp || !film_xyzw || ms->sc.empty()) return fail(PT_ERR_INVALID_ARG, "null argument");
    const int home = g_device;
    const uint32_t n = (uint32_t)ms->sc.size();
    const int64_t fw = (int64_t)rp->cropped_pixel_bounds[2] - rp->cropped_pixel_bounds[0], fh = (int64_t)rp->cropped_pixel_bounds[3] - rp->cropped_pixel_bounds[1];
    if (fw <= 0 || fh <= 0) return fail(PT_ERR_INVALID_ARG, "empty film");
    const size_t film_px = (size_t)fw * (size_t)fh;
    std::vector<int> status(n, PT_OK); std::vector<std::string> message(n);
    using clk = std::chrono::steady_clock;
    auto ms_between = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    // landing buffers on the first device, one per replica that lives elsewhere (a replica sharing the first device is summed in place)
    int st = bind_device(ms->dev[0]);
    if (st) return st;
    for (uint32_t i = 1; i < n; ++i) {
        if (ms->dev[i] == ms->dev[0] || ms->stage_cap[i] >= film_px) continue;
        if (ms->stage[i]) { hipFree(ms->stage[i]); ms->stage[i] = nullptr; }
        ms->stage_cap[i] = 0;
        if (hipMalloc((void **)&ms->stage[i], film_px * 16) != hipSuccess) { (void)hipGetLastError(); if (home >= 0) bind_device(home); return fail(PT_ERR_OUT_OF_MEMORY, "pt_multi_render: film landing buffer"); }
        ms->stage_cap[i] = film_px;
    }
    // replicas that share a device share its memory: their pass sizes are chosen here, before any of them allocates
    std::vector<uint32_t> pass_size(n, rp->spp_per_pass);
    if (rp->spp_per_pass == 0) for (uint32_t i = 0; i < n; ++i) {
        uint32_t share = 0; for (uint32_t k = 0; k < n; ++k) share += ms->dev[k] == ms->dev[i];
        if (share > 1 && bind_device(ms->dev[i]) == PT_OK) {
            PtRenderParams p = *rp; RenderConst rc;
            pt_multi_tile_shard(rp->tile_rank, rp->tile_world, i, n, &p.tile_rank, &p.tile_world);
            fill_render_const(&p, rc);
            const uint32_t ntiles = rc.ntx * rc.nty, slots = rc.tile_rank < ntiles ? (ntiles - rc.tile_rank + rc.tile_world - 1) / rc.tile_world * 256u : 0u;
            if (slots) pass_size[i] = choose_pass_size(ms->sc[i], slots, rp->spp, share, rc.volpath != 0);
        }
    }
    std::vector<clk::time_point> t_rendered(n);
    auto worker = [&](uint32_t i) {
        const clk::time_point t0 = clk::now();
        int st = bind_device(ms->dev[i]);
        if (!st && ms->film_cap[i] < film_px) {
            if (ms->film[i]) { hipFree(ms->film[i]); ms->film[i] = nullptr; }
            ms->film_cap[i] = 0;
            if (hipMalloc((void **)&ms->film[i], film_px * 16) != hipSuccess) { (void)hipGetLastError(); st = fail(PT_ERR_OUT_OF_MEMORY, "pt_multi_render: film replica"); }
            else ms->film_cap[i] = film_px;
        }
        if (!st && hipMemset(ms->film[i], 0, film_px * 16) != hipSuccess) st = fail(PT_ERR_HIP, "pt_multi_render: memset");
        if (!st) {
            PtRenderParams p = *rp;
            pt_multi_tile_shard(rp->tile_rank, rp->tile_world, i, n, &p.tile_rank, &p.tile_world);
            p.spp_per_pass = pass_size[i];
            st = pt_render(ms->sc[i], &p, ms->film[i], 1);
        }
        t_rendered[i] = clk::now();
        ms->render_ms[i] = ms_between(t0, t_rendered[i]); ms->copy_ms[i] = 0;
        // merge_film_tile across devices, first half: every replica pushes its film to its own landing buffer on the first device as soon
        // as it has finished -- the copies of different replicas travel on different xGMI links at the same time, and an early
        // finisher's copy hides behind the others' rendering.
        if (!st && i > 0 && ms->dev[i] != ms->dev[0]) {
            hipStream_t cs = ms->sc[i]->stream;
            if (hipMemcpyPeerAsync(ms->stage[i], ms->dev[0], ms->film[i], ms->dev[i], film_px * 16, cs) != hipSuccess || hipStreamSynchronize(cs) != hipSuccess)
                st = fail(PT_ERR_HIP, std::string("pt_multi_render: peer copy: ") + hipGetErrorString(hipGetLastError()));
            ms->copy_ms[i] = ms_between(t_rendered[i], clk::now());
        }
        status[i] = st; if (st) message[i] = g_error;
    };
    std::vector<std::thread> threads;
    for (uint32_t i = 1; i < n; ++i) threads.emplace_back(worker, i);
    worker(0);                                   // the calling thread drives the first replica
    for (auto &t : threads) t.join();
    for (uint32_t i = 0; i < n; ++i) if (status[i]) { bind_device(home >= 0 ? home : ms->dev[0]); return fail(status[i], "replica " + std::to_string(i) + " (device " + std::to_string(ms->dev[i]) + "): " + message[i]); }
    clk::time_point t_last = t_rendered[0];
    for (uint32_t i = 1; i < n; ++i) if (t_rendered[i] > t_last) t_last = t_rendered[i];
    // second half: ONE kernel on the first device sums all films pixel by pixel, float adds in replica order (so the result does not
    // depend on which replica finished first), and adds the sum to the caller's film.
    if ((st = bind_device(ms->dev[0]))) return st;
    pt_scene *s0 = ms->sc[0];
    FilmSumArgs fa; fa.n = n;
    for (uint32_t i = 0; i < n; ++i) fa.src[i] = (const float4 *)((i == 0 || ms->dev[i] == ms->dev[0]) ? ms->film[i] : ms->stage[i]);
    const unsigned blocks = (unsigned)((film_px + 255) / 256);
    if (film_is_device) {
        hipLaunchKernelGGL(k_film_sum, dim3(blocks), dim3(256), 0, s0->stream, fa, (float4 *)film_xyzw, 1, film_px);
        HIP_TRY(hipStreamSynchronize(s0->stream));
    } else {
        hipLaunchKernelGGL(k_film_sum, dim3(blocks), dim3(256), 0, s0->stream, fa, (float4 *)ms->film[0], 0, film_px);
        HIP_TRY(hipStreamSynchronize(s0->stream));
        std::vector<float> host(film_px * 4);
        HIP_TRY(hipMemcpy(host.data(), ms->film[0], film_px * 16, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < film_px * 4; ++k) film_xyzw[k] += host[k];
    }
    HIP_TRY(hipGetLastError());
    ms->merge_ms = ms_between(t_last, clk::now());
    // counters: the work of all replicas
    PtCounters &c = ms->counters; std::memset(&c, 0, sizeof c);
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t *src = reinterpret_cast<const uint64_t *>(&ms->sc[i]->counters); uint64_t *dst = reinterpret_cast<uint64_t *>(&c);
        for (size_t k = 0; k < sizeof(PtCounters) / 8; ++k) dst[k] += src[k];
    }
    if (home >= 0 && home != ms->dev[0]) bind_device(home);
    return PT_OK;
}

int pt_multi_get_peer_access(const pt_multi_scene *ms, int *peer, uint32_t max_replicas) {
    if (!ms || !peer) return fail(PT_ERR_INVALID_ARG, "null argument");
    for (uint32_t i = 0; i < max_replicas && i < ms->sc.size(); ++i) peer[i] = ms->peer[i];
    return PT_OK;
}

int pt_multi_get_timing(const pt_multi_scene *ms, double *merge_ms, double *render_ms, double *copy_ms, uint32_t max_replicas) {
    if (!ms) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (merge_ms) *merge_ms = ms->merge_ms;
    for (uint32_t i = 0; i < max_replicas && i < ms->sc.size(); ++i) { if (render_ms) render_ms[i] = ms->render_ms[i]; if (copy_ms) copy_ms[i] = ms->copy_ms[i]; }
    return PT_OK;
}

int pt_multi_get_create_timing(const pt_multi_scene *ms, double *wall_ms, double *replica_ms, uint32_t max_replicas) {
    if (!ms) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (wall_ms) *wall_ms = ms->create_wall_ms;
    for (uint32_t i = 0; replica_ms && i < max_replicas && i < ms->create_ms.size(); ++i) replica_ms[i] = ms->create_ms[i];
    return PT_OK;
}

int pt_multi_get_counters(const pt_multi_scene *ms, PtCounters *out) {
    if (!ms || !out) return fail(PT_ERR_INVALID_ARG, "null argument");
    *out = ms->counters;
    return PT_OK;
}
int pt_multi_get_kernel_stats(const pt_multi_scene *ms, uint32_t replica, PtKernelStat *out, uint32_t max_entries, uint32_t *n_out) {
    if (!ms || replica >= ms->sc.size()) return fail(PT_ERR_INVALID_ARG, "replica out of range");
    return pt_get_kernel_stats(ms->sc[replica], out, max_entries, n_out);
}

}  // extern "C"
