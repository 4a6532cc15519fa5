// capi.hip -- host driver behind include/mi355pt.h: scene upload, wavefront scheduling on one HIP stream,
// film hand-off, counters and per-kernel HIP-event timing. One process drives one GPU (pt_init).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "kern_decl.h"   // kernel declarations; the definitions are instantiated by the tu_*.hip translation units
#include "host_bvh.h"

extern "C" const unsigned char pt_sobol_blob[];   // tables_blob.cpp (.incbin of data/sobol_tables.bin)
extern "C" const unsigned int pt_sobol_blob_size;

namespace {

// Device binding is per host thread (hipSetDevice is): every thread that enters the library is bound to one device, whose Sobol' /
// Halton tables and CU count it sees through these thread-local views of the per-device contexts below. pt_init selects the
// process-wide default; a scene remembers the device it was created on and re-binds the calling thread when needed, so one
// process can drive several GPUs (pt_multi_*: one host thread + stream per device).
thread_local std::string g_error;
thread_local int g_device = -1;
thread_local int g_num_cus = 256;
std::atomic<int> g_default_device{-1};
// traversal scheduling knobs (env PT_TRACE_REFILL_MIN / PT_TRACE_LEAF_QUORUM override)
uint32_t g_refill_min[4] = {16, 16, 16, 48};     // per launch kind: extend, extend_mis, shadow, extend_camera (round 3, 256-spp passes: {24, 24, 24, 32} -> these: camera launch 42.7 -> 41.1 ms on C2, 95.9 -> 85.4 on C3)
uint32_t g_leaf_quorum[4] = {8, 8, 8, 8};         // lanes at a leaf wait until this many of them do, then all of them run until none is left (sticky). History: round 1
                                                  // shipped {24, 24, 24, 32}, but its ballot ran under the leaf lanes' exec mask and never held a lane back; with the ballot
                                                  // fixed, a quorum that has to form again for every packet of a leaf lost (C2 at 64 spp: 1 -> 1132, 8 -> 1084, 24 -> 1038);
                                                  // the sticky form is a small gain (C2 trace 166.4 -> 163.4 ms per step at 8, flat to 24; C3 349 -> 345; C4 unchanged). That a
                                                  // step which runs the 149-instruction triangle test seven times less often gains 2 % says the loop is bound by the latency
                                                  // of its dependent gathers (~2 us under load, six waves per SIMD), not by instruction issue.
bool g_refill_from_env = false;
bool g_trace_split = false;                        // env PT_TRACE_SPLIT=1: one traversal launch per ray kind (extend / extend_mis / shadow) instead of the mixed launch
bool g_trace_exact = false;                        // pt_set_trace_exact / env PT_TRACE_EXACT=1: walk the two-wide records, PtCounters.bvh_nodes_visited is then the reference's count
uint32_t g_inst_quorum = 16;                      // lanes waiting for the instance transform step (env PT_TRACE_INST_QUORUM)
uint32_t g_trace_waves_per_cu = 28;               // persistent trace waves per CU = 7 per SIMD: k_trace<*, 0> needs 71 VGPRs and 5 KB of LDS per wave (env PT_TRACE_WAVES_PER_CU; 20 -> 24: +1 %, 24 -> 28: +3 %)
thread_local SobolTables g_tabs = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
constexpr int kMaxDevices = kMaxReplicas;
struct DevCtx { bool ready = false; int num_cus = 256; SobolTables tabs = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; };
DevCtx g_ctx[kMaxDevices];
std::mutex g_ctx_mutex;

int fail(int code, const std::string &msg) { g_error = msg; return code; }
#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return fail(PT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));     \
    } while (0)

// Scratch device allocations of one C-ABI call: freed on every exit path (HIP_TRY returns early on errors).
struct DevTmp {
    std::vector<void *> p;
    template <class T> hipError_t alloc(T **out, size_t bytes) { void *q = nullptr; hipError_t e = hipMalloc(&q, bytes ? bytes : 1); if (e == hipSuccess) { p.push_back(q); *out = (T *)q; } return e; }
    ~DevTmp() { for (void *q : p) hipFree(q); }
};

int bind_device(int device);
int ensure_device() {
    if (g_device >= 0) return PT_OK;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(PT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    const int def = g_default_device.load();
    return def >= 0 ? bind_device(def) : pt_init(0);
}

int upload_tables(SobolTables &g_tabs) {   // (fills the per-device context passed in; the name shadows the thread-local view on purpose)
    if (g_tabs.m32) return PT_OK;
    if (pt_sobol_blob_size != 8 + 1024 * 52 * 4 + (25 + 26) * 52 * 8 || std::memcmp(pt_sobol_blob, "PTSOBOL1", 8) != 0)
        return fail(PT_ERR_INVALID_ARG, "embedded Sobol table blob is corrupt");
    const unsigned char *p = pt_sobol_blob + 8;
    void *d = nullptr;
    size_t bytes = pt_sobol_blob_size - 8;
    HIP_TRY(hipMalloc(&d, bytes));
    HIP_TRY(hipMemcpy(d, p, bytes, hipMemcpyHostToDevice));
    g_tabs.m32 = (const uint32_t *)d;
    g_tabs.vdc = (const uint64_t *)((const char *)d + 1024 * 52 * 4);
    g_tabs.vdc_inv = g_tabs.vdc + 25 * 52;
    {   // the generator matrices folded by index nibble (dev_sampler.h: SobolTables::nib)
        std::vector<uint32_t> m32(1024 * 52), nib((size_t)1024 * kSobolNibWords);
        std::memcpy(m32.data(), p, m32.size() * 4);
        for (uint32_t d = 0; d < 1024; ++d) for (uint32_t j = 0; j < kSobolNibbles; ++j) for (uint32_t n = 0; n < 16; ++n) {
            uint32_t v = 0;
            for (uint32_t b = 0; b < 4; ++b) if (n >> b & 1u) v ^= m32[d * 52 + 4 * j + b];
            nib[((size_t)j * 1024 + d) * 16 + n] = v;
        }
        void *dn = nullptr;
        HIP_TRY(hipMalloc(&dn, nib.size() * 4));
        HIP_TRY(hipMemcpy(dn, nib.data(), nib.size() * 4, hipMemcpyHostToDevice));
        g_tabs.nib = (const uint32_t *)dn;
    }
    // Halton: the first 1000 primes (PRIMES / PRIME_SUMS, lowdiscrepancy.rs:9-192: here sieved, not tabulated) and the digit
    // permutations of compute_radical_inverse_permutations(&mut RNG::default()) (lowdiscrepancy.rs:359-378): per base the
    // identity permutation shuffled by `shuffle` (sampling.rs:178-186) with PCG32 (rng.rs:17-58), one RNG for all bases.
    {
        std::vector<uint32_t> primes, sums;
        for (uint32_t c = 2; primes.size() < kHaltonMaxDims; ++c) { bool pr = true; for (uint32_t q : primes) { if (q * q > c) break; if (c % q == 0) { pr = false; break; } } if (pr) primes.push_back(c); }
        uint32_t total = 0;
        for (uint32_t q : primes) { sums.push_back(total); total += q; }
        std::vector<uint16_t> perm(total);
        uint64_t state = 0x853c49e6748fea9bull; const uint64_t inc = 0xda3e39cb94b95bdbull;
        auto uniform_u32 = [&]() {
            const uint64_t old = state;
            state = old * 0x5851f42d4c957f2dull + inc;
            const uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27), rot = (uint32_t)(old >> 59);
            return (xs >> rot) | (xs << ((~rot + 1u) & 31u));
        };
        auto uniform_below = [&](uint32_t b) { const uint32_t threshold = (~b + 1u) % b; for (;;) { const uint32_t r = uniform_u32(); if (r >= threshold) return r % b; } };
        for (size_t i = 0; i < primes.size(); ++i) {
            uint16_t *pp = perm.data() + sums[i];
            for (uint32_t j = 0; j < primes[i]; ++j) pp[j] = (uint16_t)j;
            for (uint32_t j = 0; j < primes[i]; ++j) { const uint32_t other = j + uniform_below(primes[i] - j); std::swap(pp[j], pp[other]); }
        }
        void *dp = nullptr, *ds = nullptr, *dm = nullptr;
        HIP_TRY(hipMalloc(&dp, primes.size() * 4)); HIP_TRY(hipMalloc(&ds, sums.size() * 4)); HIP_TRY(hipMalloc(&dm, perm.size() * 2));
        HIP_TRY(hipMemcpy(dp, primes.data(), primes.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ds, sums.data(), sums.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dm, perm.data(), perm.size() * 2, hipMemcpyHostToDevice));
        g_tabs.prime = (const uint32_t *)dp; g_tabs.prime_sum = (const uint32_t *)ds; g_tabs.perm = (const uint16_t *)dm;
    }
    return PT_OK;
}

// Bind the calling thread to `device`: hipSetDevice + the device's context (created on first use: CU count, Sobol' / Halton tables).
int bind_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(PT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n || device >= kMaxDevices) return fail(PT_ERR_INVALID_ARG, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    DevCtx &c = g_ctx[device];
    if (!c.ready) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        int st = upload_tables(c.tabs);
        if (st) return st;
        c.ready = true;
    }
    g_device = device; g_num_cus = c.num_cus; g_tabs = c.tabs;
    return PT_OK;
}

struct Stat { std::string name, kernel; uint64_t launches = 0; double ms = 0; uint64_t items = 0, nodes = 0, tris = 0; };
struct TimedLaunch { int stat; hipEvent_t a, b; bool closed; };

}  // namespace

struct pt_scene {
    int device = 0;                    // the HIP device every allocation of this scene lives on
    std::vector<void *> allocs;
    DeviceScene ds{};
    std::vector<PtBVHNode> nodes;
    std::vector<uint32_t> ordered;
    bool class_used[kNumClasses] = {true, false, false, false, true, false, false};
    bool has_null_material = false;   // a primitive without a material: a medium-interface shell (api.rs:597). The path integrator steps over it (path.rs:124-129);
                                      // the volumetric one also walks its shadow / MIS rays through it, segment by segment (kern_shade_common.h: vol_chain_step)
    void *ext_slab = nullptr; size_t ext_capacity = 0;   // PathSoA::ext, allocated for volpath renders of scenes with shells
    bool has_bssrdf = false;           // any subsurface material: probe queues + BssSoA are allocated
    void *bss_slab = nullptr; BssSoA bs{};
    uint4 *probe_ring = nullptr;       // k_trace<.., PROBE>: kProbeRing x 3 x uint4 per persistent lane
    uint32_t n_lights = 0;
    std::vector<PtLight> host_lights; uint32_t env_w = 0, env_h = 0; float env_texel0[3] = {0, 0, 0};
    // light grids (lazy, per effective strategy)
    LightGrid grid[5]{}; bool grid_ready[5] = {false, false, false, false, false};   // by PtLightStrategy; PT_LS_SPATIAL itself resolves to _EAGER or _LAZY
    // PT_LS_SPATIAL_LAZY: voxels are filled when a vertex first needs them (lightdistrib.rs:233-337), once per wavefront iteration
    struct LazyGrid { unsigned long long *cell_ptr = nullptr; float *zero_block = nullptr; uint32_t *req_flag = nullptr, *req_list = nullptr, *req_count = nullptr, *missing = nullptr;
                      size_t ncell = 0, stride = 0; uint64_t filled = 0; } lazy;
    // render workspace
    hipStream_t stream = nullptr;
    void *slab = nullptr; size_t capacity = 0; PathSoA ps{};
    uint32_t *qbuf = nullptr; QueueSet q{};
    QCounters *qc = nullptr; DevCounters *dc = nullptr;
    uint32_t *spill = nullptr; uint32_t spill_waves = 0;
    float *film_rgbw = nullptr; size_t film_px = 0;
    float *d_filter = nullptr;
    PtCounters counters{};
    std::vector<Stat> stats;
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> event_pool;
    bool profile = false;
    int last_stat = -1;

    template <class T> int dalloc(T **out, size_t count) {
        void *p = nullptr;
        if (count == 0) count = 1;
        hipError_t e = hipMalloc(&p, count * sizeof(T));
        if (e != hipSuccess) return fail(PT_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
        allocs.push_back(p);
        *out = (T *)p;
        return PT_OK;
    }
    template <class T> int upload(const T **out, const T *src, size_t count) {
        T *d = nullptr;
        int st = dalloc(&d, count);
        if (st) return st;
        if (count && src) HIP_TRY(hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice));
        *out = d;
        return PT_OK;
    }
    int stat_id(const char *name) {
        for (size_t i = 0; i < stats.size(); ++i) if (stats[i].name == name) return (int)i;
        stats.push_back(Stat{name}); return (int)stats.size() - 1;
    }
    hipEvent_t get_event() {
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e; hipEventCreate(&e); return e;
    }
    // bracket a launch with HIP events on the render stream when profiling
    void begin(const char *name, uint64_t items) {
        int id = stat_id(name);
        last_stat = id;
        stats[id].launches++; stats[id].items += items;
        if (profile) { TimedLaunch t{id, get_event(), get_event(), false}; hipEventRecord(t.a, stream); timed.push_back(t); }
    }
    void end() { if (profile && !timed.empty()) { hipEventRecord(timed.back().b, stream); timed.back().closed = true; } }
    // the kernel symbol behind the launch kind opened by the last begin(), as rocprofv3 prints it
    void set_kernel(const std::string &symbol) { if (last_stat >= 0 && last_stat < (int)stats.size()) stats[last_stat].kernel = symbol; }
    void resolve_timings() {
        for (auto &t : timed) {
            float ms = 0;
            if (t.closed && t.stat >= 0 && t.stat < (int)stats.size() && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) stats[t.stat].ms += ms;
            event_pool.push_back(t.a); event_pool.push_back(t.b);
        }
        timed.clear();
    }
    // A call that failed half way leaves event pairs behind whose stat ids belong to the statistics of THAT call: hand the
    // events back without touching `stats` (entry of every C-ABI call that clears `stats`, and pt_scene_destroy).
    void drop_timings() {
        for (auto &t : timed) { event_pool.push_back(t.a); event_pool.push_back(t.b); }
        timed.clear(); last_stat = -1;
    }
};

// One process, several devices: a replica of the scene per device, one host thread per replica inside pt_multi_render.
struct pt_multi_scene {
    std::vector<pt_scene *> sc;       // replica i lives on dev[i] (a device may appear more than once: replicas then share it)
    std::vector<int> dev;
    std::vector<float *> film;        // per replica: XYZ + weight sums of its tiles, on its device
    std::vector<size_t> film_cap;     // pixels film[i] holds (0: not allocated); a failed or smaller render never leaves a stale size behind
    std::vector<float *> stage;       // on dev[0], one landing buffer per replica that lives on ANOTHER device: the peer copies of all
    std::vector<size_t> stage_cap;    // sources are in flight together (one xGMI link each), issued by the replicas' own host threads
    std::vector<double> render_ms, copy_ms;   // last pt_multi_render, per replica: wall time of its pt_render / of its peer copy
    std::vector<int> peer;            // per replica: PT_PEER_* -- how its film reaches the first device
    double merge_ms = 0;              // last pt_multi_render: from the last replica's render end to the summed film (copy tails + the sum kernel)
    PtCounters counters{};
};

namespace {

// Shade class of a material = the kernel its vertices are shaded by (kernels.h: kNumClasses): by the number of BxDFs the material
// can produce, decided from its constant parameters (a textured parameter can take any value).
uint8_t material_class(const PtMaterial &m) {
    auto textured = [&](int slot) { return m.tex[slot] >= 0; };
    auto black = [](const float c[3]) { return !(c[0] > 0.0f) && !(c[1] > 0.0f) && !(c[2] > 0.0f); };   // .clamps(0, inf).is_black()
    switch (m.type) {
    case PT_MAT_MATTE: return 0;
    case PT_MAT_MIRROR: return (uint8_t)kSpecClass;
    case PT_MAT_METAL: case PT_MAT_SUBSTRATE: return 1;
    case PT_MAT_GLASS:   // glass.rs:57-92: one FresnelSpecular lobe when both roughnesses are 0, else up to two microfacet lobes
        if (textured(PT_MP_U_ROUGHNESS) || textured(PT_MP_V_ROUGHNESS)) return 2;
        return (m.u_roughness == 0.0f && m.v_roughness == 0.0f) ? (uint8_t)kSpecClass : 2;
    case PT_MAT_PLASTIC: return 2;
    case PT_MAT_UBER: {   // uber.rs:40-106: without specular reflection / transmission and fully opaque it is Lambertian + microfacet
        const bool opaque = !textured(PT_MP_OPACITY) && m.opacity[0] >= 1.0f && m.opacity[1] >= 1.0f && m.opacity[2] >= 1.0f;
        const bool no_spec = !textured(PT_MP_KR) && !textured(PT_MP_KT) && black(m.kr) && black(m.kt);
        return (opaque && no_spec) ? 2 : 3;
    }
    default: return 3;
    }
}

// Distribution1D::new on the host (sampling.rs:12-34) for the uniform / power strategies and the env map.
void dist1d(const std::vector<float> &func, std::vector<float> &cdf, float &func_int) {
    size_t n = func.size();
    cdf.assign(n + 1, 0.0f);
    for (size_t i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / (float)n;
    func_int = cdf[n];
    if (func_int == 0.0f) { for (size_t i = 1; i < n + 1; ++i) cdf[i] = (float)i / (float)n; }
    else { for (size_t i = 1; i < n + 1; ++i) cdf[i] /= func_int; }
}

#ifdef PT_TRACE_UTIL
__global__ void k_trace_util_fold(DevCounters *dc, uint32_t kind, uint32_t waves, int reset) {   // per launch: span x waves, then re-arm min / max
    if (!reset) dc->tail[5 + 2 * kind] += (dc->tail[1] - dc->tail[0]) * waves;
    dc->tail[0] = ~0ull; dc->tail[1] = 0ull;
}
#endif

// any: 0 closest hit, 1 any hit (rays of job.sub[0]); 2 mixed: the queues of job.sub[0..2] in one launch (n_upper covers all three)
int launch_trace(pt_scene *sc, int any, TraceJob job, uint32_t n_upper, bool probe = false) {
    if (n_upper == 0) return PT_OK;
    const uint32_t knob = job.sub[0].kind == 4 ? 0 : (job.sub[0].kind & 3);
    job.refill_min = (probe && !g_refill_from_env) ? 24u : g_refill_min[knob]; job.leaf_quorum = g_leaf_quorum[knob];   // (probe chains: 24 measured best on C5, 16: +1.6 %)
    if (sc->ds.n_instances > 0 && !g_refill_from_env) job.refill_min = 8;   // rays through instanced scenes are long (S4: 200 node visits): idle lanes are refilled early (measured 24 -> 8: +9 %)
    uint32_t waves = (n_upper + 63) / 64;
    uint32_t blocks = std::min<uint32_t>((waves + 3) / 4, sc->spill_waves / 4);
    const int mode = (sc->ds.tri_alpha || sc->ds.tri_shadow_alpha) ? 2 : sc->ds.n_spheres > 0 ? 1 : sc->ds.n_instances > 0 ? 3 : 0;  // kern_trace.h: k_trace MODE
    job.inst_quorum = g_inst_quorum;
#ifdef PT_TRACE_UTIL
    hipLaunchKernelGGL(k_trace_util_fold, dim3(1), dim3(1), 0, sc->stream, sc->dc, job.sub[0].kind & 3u, 0u, 1);
#endif
    const bool quad = !g_trace_exact;   // production: the four-wide records; pt_set_trace_exact(1): the two-wide walk with the reference's node-visit counter
    #define PT_LAUNCH_TRACE(A, M, P) do { if (quad) hipLaunchKernelGGL((k_trace<A, M, P, true>), dim3(blocks), dim3(kTraceBlock), 0, sc->stream, sc->ds, job); \
                                          else hipLaunchKernelGGL((k_trace<A, M, P, false>), dim3(blocks), dim3(kTraceBlock), 0, sc->stream, sc->ds, job); } while (0)
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
    sc->set_kernel(std::string("k_trace<") + std::to_string(any) + ", " + std::to_string(mode) + ", " + (probe ? "true" : "false") + ", " + (quad ? "true" : "false") + ">");
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
        if ((st = sc->dalloc(&sc->spill, (size_t)sc->spill_waves * 64 * 2 * kSpillEntries))) return st;
        if ((st = sc->dalloc(&sc->d_filter, 256))) return st;
        if (sc->has_bssrdf && (st = sc->dalloc(&sc->probe_ring, (size_t)sc->spill_waves * 64 * kProbeRing * 3))) return st;
    }
    if (capacity > sc->capacity) {
        if (sc->slab) { hipFree(sc->slab); hipFree(sc->qbuf); sc->slab = nullptr; sc->qbuf = nullptr; }
        if (sc->ext_slab) { hipFree(sc->ext_slab); sc->ext_slab = nullptr; sc->ext_capacity = 0; }
        if (sc->bss_slab) { hipFree(sc->bss_slab); sc->bss_slab = nullptr; }
        size_t bytes = capacity * (size_t)kPathBytes + 4096;
        hipError_t e = hipMalloc(&sc->slab, bytes);
        if (e != hipSuccess) return fail(PT_ERR_OUT_OF_MEMORY, "path-state slab: " + std::string(hipGetErrorString(e)));
        char *p = (char *)sc->slab;   // hipMalloc returns 256-byte aligned memory; every record array starts on a 64-byte line
        PathSoA &ps = sc->ps;
        ps.core = (float *)p; p += capacity * (size_t)PathSoA::kCoreWords * 4;
        ps.nee = (float *)p; p += capacity * (size_t)PathSoA::kNeeWords * 4;
        ps.mis = (float *)p; p += capacity * (size_t)PathSoA::kMisWords * 4;
        ps.ray = (float *)p; p += capacity * (size_t)PathSoA::kRayWords * 4;
        ps.hit = (float *)p; p += capacity * (size_t)PathSoA::kHitWords * 4;
        if (sc->has_bssrdf) {
            e = hipMalloc(&sc->bss_slab, capacity * (size_t)kBssSoAArrays * 4);
            if (e != hipSuccess) return fail(PT_ERR_OUT_OF_MEMORY, "BSSRDF probe state: " + std::string(hipGetErrorString(e)));
            BssSoA &bs = sc->bs;
            float *bp = (float *)sc->bss_slab;
            float **ba[] = {&bs.start_x, &bs.start_y, &bs.start_z, &bs.target_x, &bs.target_y, &bs.target_z, &bs.po_x, &bs.po_y, &bs.po_z,
                            &bs.ns_x, &bs.ns_y, &bs.ns_z, &bs.ss_x, &bs.ss_y, &bs.ss_z, &bs.u1n, &bs.sa_r, &bs.sa_g, &bs.sa_b, &bs.sc_r, &bs.sc_g, &bs.sc_b};
            for (float **f : ba) { *f = bp; bp += capacity; }
            bs.mat = (uint32_t *)bp; bp += capacity; bs.cnt = (uint32_t *)bp; bp += capacity; bs.iface = (uint32_t *)bp;
            static_assert(sizeof(ba) / sizeof(ba[0]) + 3 == kBssSoAArrays, "BssSoA layout");
        }
        // queues: ext[2] + shade[2][classes] + shadow + mis (+ probe[2])
        size_t nq = 2 + 2 * kNumClasses + 2 + (sc->has_bssrdf ? 2 : 0);
        e = hipMalloc((void **)&sc->qbuf, nq * capacity * 4);
        if (e != hipSuccess) return fail(PT_ERR_OUT_OF_MEMORY, "queues: " + std::string(hipGetErrorString(e)));
        uint32_t *qp = sc->qbuf;
        for (int i = 0; i < 2; ++i) { sc->q.ext[i] = qp; qp += capacity; }
        for (int i = 0; i < 2; ++i) for (int c = 0; c < kNumClasses; ++c) { sc->q.shade[i][c] = qp; qp += capacity; }
        sc->q.shadow = qp; qp += capacity; sc->q.mis = qp; qp += capacity;
        sc->q.probe[0] = sc->q.probe[1] = nullptr;
        if (sc->has_bssrdf) { sc->q.probe[0] = qp; qp += capacity; sc->q.probe[1] = qp; }
        sc->capacity = capacity;
    }
    if (film_px > sc->film_px) {
        if (sc->film_rgbw) hipFree(sc->film_rgbw);
        HIP_TRY(hipMalloc((void **)&sc->film_rgbw, film_px * 16));
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
    sc->set_kernel("k_shade<" + std::to_string(MAXL) + ", " + std::to_string(mode) + ", " + std::to_string(DIFF) + ">");
    if (rc.volpath) hipLaunchKernelGGL((k_shade<MAXL, 3, DIFF == 2 ? 0 : DIFF>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, rc, g_tabs, grid, sc->ps, job);
    else if (sc->ds.n_textures > 0) hipLaunchKernelGGL((k_shade<MAXL, 2, DIFF>), dim3(blocks), dim3(256), 0, sc->stream, sc->ds, rc, g_tabs, grid, sc->ps, job);
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
    const size_t per_path = (size_t)kPathBytes + 4u * (2 + 2 * kNumClasses + 2 + (sc->has_bssrdf ? 2 : 0)) + (sc->has_bssrdf ? 4u * kBssSoAArrays : 0u)
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
        float *blocks = nullptr; int st;
        if ((st = sc->dalloc(&blocks, n * z.stride))) return st;
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
    int cur = 0;
    static const char *shade_names[kNumClasses] = {"shade_matte", "shade_1lobe", "shade_2lobe", "shade_uber", "shade_miss", "shade_medium", "shade_specular"};
    const int kMaxIterations = 1 << 20;   // a path needs <= max_depth + null-surface skips + probe segments iterations
    for (int iter = 0; iter <= kMaxIterations; ++iter) {
        QCounters h;
        HIP_TRY(hipMemcpyAsync(&h, qc, sizeof h, hipMemcpyDeviceToHost, sc->stream));
        HIP_TRY(hipStreamSynchronize(sc->stream));
        if (h.error) return fail((int)h.error, h.error == PT_ERR_STACK_OVERFLOW ? "BVH traversal stack overflow (> 64 entries)" : h.error == PT_ERR_PROBE_CHAIN ? "BSSRDF probe chain with more than 2^32 - 1 intersections" : "Sobol dimension overflow (>= 1024)");
        if (iter == kMaxIterations) return fail(PT_ERR_PROBE_CHAIN, "pass did not finish within 2^20 wavefront iterations");
        if (iter > 0 && iter % 2048 == 0 && getenv("PT_DEBUG_ITER")) {
            fprintf(stderr, "[iter %d] ext %u shadow %u mis %u probe %u shade:", iter, h.ext[cur], h.shadow, h.mis, h.probe[cur]);
            for (int c = 0; c < kNumClasses; ++c) fprintf(stderr, " %u", h.shade[cur][c]);
            fprintf(stderr, "\n");
        }
        const uint32_t n_ext = h.ext[cur], n_resolve = h.shade[cur][kMissClass], n_shadow = h.shadow, n_mis = h.mis, n_probe = h.probe[cur];
        // volpath with grid media: vertices that did their NEE set-up last iteration wait in their own shade class for stage B
        uint32_t n_stage_b = 0;
        if (rc.volpath && (sc->ds.has_grid || sc->ds.has_shells)) for (int c = 0; c < kNumClasses; ++c) if (c != kMissClass) n_stage_b += h.shade[cur][c];
        if (n_ext == 0 && n_resolve == 0 && n_probe == 0 && n_stage_b == 0) break;
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
            sc->begin("route", n_ext); sc->set_kernel("k_route");
            hipLaunchKernelGGL(k_route, dim3(std::min<uint32_t>((n_ext + 255) / 256, (uint32_t)g_num_cus * 8u)), dim3(256), 0, sc->stream, sc->ds,
                               (const uint32_t *)sc->q.ext[cur], (const uint32_t *)&qc->ext[cur], sc->ps, &qc->shade[cur][0],
                               sc->q.shade[cur][0], sc->q.shade[cur][1], sc->q.shade[cur][2], sc->q.shade[cur][3], sc->q.shade[cur][4], sc->q.shade[cur][kSpecClass]);
            sc->end();
        }
        hipLaunchKernelGGL(k_reset, dim3(1), dim3(64), 0, sc->stream, qc, 2u, cur);
        if (grid.cell_ptr) {   // first-touch voxels: the vertices of every shade class name theirs, then the new ones are computed
            const uint32_t upper0 = n_ext + n_resolve;
            for (int c = 0; c < kNumClasses; ++c) {
                if (c == kMissClass) continue;
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
            bj.shade_next0 = sc->q.shade[1 - cur][kMissClass]; bj.shade_next0_count = &qc->shade[1 - cur][kMissClass];
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
            const bool used = sc->class_used[c] || (c == 1 && rc.volpath && sc->class_used[kSpecClass]);   // (the volumetric router folds class 6 into class 1)
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
            if (!used || class_n[c] == 0) continue;
            ShadeJob sj{};
            sj.queue = sc->q.shade[cur][c]; sj.count = &qc->shade[cur][c];
            sj.ext_next = sc->q.ext[1 - cur]; sj.ext_next_count = &qc->ext[1 - cur];
            sj.shade_next0 = sc->q.shade[1 - cur][kMissClass]; sj.shade_next0_count = &qc->shade[1 - cur][kMissClass];
            sj.shadow = sc->q.shadow; sj.shadow_count = &qc->shadow; sj.mis = sc->q.mis; sj.mis_count = &qc->mis;
            sj.error = &qc->error; sj.counters = sc->dc; sj.cls = (uint32_t)c;
            sj.self_next = sc->q.shade[1 - cur][c]; sj.self_next_count = &qc->shade[1 - cur][c];
            if (c == 3 && sc->has_bssrdf) { sj.probe_next = sc->q.probe[1 - cur]; sj.probe_next_count = &qc->probe[1 - cur]; sj.bs = sc->bs; }
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
            else launch_shade<5>(sc, rc, grid, sj, class_n[c]);
            sc->end();
        }
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_reset, dim3(1), dim3(64), 0, sc->stream, qc, 8u, cur);
        cur = 1 - cur;
    }
    sc->begin("film", total);
        sc->set_kernel("k_film");
    hipLaunchKernelGGL(k_film, dim3((rc.n_pix_slots + 255) / 256), dim3(256), 0, sc->stream, rc, sc->ps, sc->d_filter, sc->film_rgbw, sc->dc);
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
    static const char *sn[kNumClasses] = {"shade_matte", "shade_1lobe", "shade_2lobe", "shade_uber", "shade_miss", "shade_medium", "shade_specular"};
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

}  // namespace

extern "C" {

int pt_init(int device_ordinal) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(PT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(PT_ERR_INVALID_ARG, "device ordinal out of range");
    if (const char *e = getenv("PT_TRACE_REFILL_MIN")) { int a = 0, b = 0, c = 0, d = 0; int n = sscanf(e, "%d,%d,%d,%d", &a, &b, &c, &d); if (n == 1) b = c = d = a; if (n == 3) d = a; if (n >= 1) { g_refill_min[0] = a; g_refill_min[1] = b; g_refill_min[2] = c; g_refill_min[3] = d; g_refill_from_env = true; } }
    if (const char *e = getenv("PT_TRACE_SPLIT")) g_trace_split = atoi(e) != 0;
    if (const char *e = getenv("PT_TRACE_EXACT")) g_trace_exact = atoi(e) != 0;
    if (const char *e = getenv("PT_TRACE_INST_QUORUM")) { int v = atoi(e); if (v >= 1 && v <= 64) g_inst_quorum = (uint32_t)v; }
    if (const char *e = getenv("PT_TRACE_WAVES_PER_CU")) { int v = atoi(e); if (v >= 4 && v <= 32) g_trace_waves_per_cu = (uint32_t)(v & ~3); }
    if (const char *e = getenv("PT_TRACE_LEAF_QUORUM")) { int a = 0, b = 0, c = 0, d = 0; int n = sscanf(e, "%d,%d,%d,%d", &a, &b, &c, &d); if (n == 1) b = c = d = a; if (n == 3) d = a; if (n >= 1) { g_leaf_quorum[0] = a; g_leaf_quorum[1] = b; g_leaf_quorum[2] = c; g_leaf_quorum[3] = d; } }
    g_default_device.store(device_ordinal);
    return bind_device(device_ordinal);
}

int pt_set_trace_exact(int exact) { const int prev = g_trace_exact ? 1 : 0; g_trace_exact = exact != 0; return prev; }

int pt_device_count(int *n_devices) {
    if (!n_devices) return fail(PT_ERR_INVALID_ARG, "null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *n_devices = n;
    return PT_OK;
}

const char *pt_last_error(void) { return g_error.c_str(); }

int pt_scene_create(const PtSceneDesc *d, pt_scene **out) {
    if (!d || !out) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (d->n_prims == 0 || !d->prim_shape || !d->prim_material || !d->prim_light) return fail(PT_ERR_INVALID_ARG, "scene has no primitives");
    if (d->n_triangles && (!d->P || !d->indices)) return fail(PT_ERR_INVALID_ARG, "triangle arrays missing");
    if (d->n_spheres && !d->spheres) return fail(PT_ERR_INVALID_ARG, "sphere array missing");
    for (uint32_t i = 0; i < 3 * d->n_triangles; ++i) if (d->indices[i] >= d->n_vertices) return fail(PT_ERR_INVALID_ARG, "vertex index out of range");
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        uint32_t s = d->prim_shape[i];
        const uint32_t kind = s >> 30, idx = s & 0x3fffffffu;
        if (!((kind == PT_SHAPE_TRIANGLE && idx < d->n_triangles) || (kind == PT_SHAPE_SPHERE && idx < d->n_spheres))) return fail(PT_ERR_INVALID_ARG, "primitive shape reference out of range");
        if (d->prim_material[i] != PT_NONE && d->prim_material[i] >= d->n_materials) return fail(PT_ERR_INVALID_ARG, "material index out of range");
        if (d->prim_light[i] != PT_NONE && d->prim_light[i] >= d->n_lights) return fail(PT_ERR_INVALID_ARG, "light index out of range");
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const PtMaterial &m = d->materials[i];
        if (m.type != PT_MAT_SUBSURFACE) continue;
        if (!d->bssrdf_tables || m.bssrdf_table >= d->n_bssrdf_tables) return fail(PT_ERR_INVALID_ARG, "subsurface material without a BSSRDF table");
        const PtBSSRDFTable &t = d->bssrdf_tables[m.bssrdf_table];
        if (t.n_rho < 2 || t.n_radius < 2 || !t.rho_samples || !t.radius_samples || !t.profile || !t.rhoeff || !t.profile_cdf) return fail(PT_ERR_INVALID_ARG, "incomplete BSSRDF table");
    }
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const PtTexture &t = d->textures[i];
        if (t.type > PT_TEX_DOTS) return fail(PT_ERR_UNSUPPORTED, "texture type not implemented");
        for (int k = 0; k < 3; ++k) if (t.child[k] >= (int32_t)d->n_textures) return fail(PT_ERR_INVALID_ARG, "texture child index out of range");
        if (t.type == PT_TEX_IMAGEMAP) {
            if (!d->images || t.image >= d->n_images) return fail(PT_ERR_INVALID_ARG, "image texture without an image");
            const PtImage &im = d->images[t.image];
            auto pow2 = [](uint32_t v) { return v && !(v & (v - 1)); };
            if (!pow2(im.width) || !pow2(im.height) || !im.texels || (im.channels != 1 && im.channels != 3) || im.n_levels == 0 || im.n_levels > 16)
                return fail(PT_ERR_INVALID_ARG, "PtImage must be a power-of-two MIPMap pyramid with 1 or 3 channels");
            if (t.wrap > PT_WRAP_BLACK) return fail(PT_ERR_UNSUPPORTED, "ImageWrap::Clamp is not implemented");
            if (!t.trilinear && !d->ewa_weight_lut) return fail(PT_ERR_INVALID_ARG, "EWA image texture without ewa_weight_lut");
        }
        if ((t.type == PT_TEX_SCALE || t.type == PT_TEX_CHECKERBOARD2D || t.type == PT_TEX_CHECKERBOARD3D || t.type == PT_TEX_DOTS) && (t.child[0] < 0 || t.child[1] < 0)) return fail(PT_ERR_INVALID_ARG, "texture node needs two children");
        if (t.type == PT_TEX_MIX && (t.child[0] < 0 || t.child[1] < 0 || t.child[2] < 0)) return fail(PT_ERR_INVALID_ARG, "mix texture needs three children");
    }
    for (const int32_t *arr : {d->tri_alpha, d->tri_shadow_alpha})
        if (arr) for (uint32_t i = 0; i < d->n_triangles; ++i) if (arr[i] >= (int32_t)d->n_textures) return fail(PT_ERR_INVALID_ARG, "alpha-mask texture index out of range");
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const PtMaterial &m = d->materials[i];
        if (m.type > PT_MAT_DISNEY) return fail(PT_ERR_INVALID_ARG, "unknown material type");
        if (m.type == PT_MAT_DISNEY) {   // disney.rs:741-836: BxDFs the parameters can produce; the shade class holds five
            const bool thin = m.disney_thin != 0;
            const float dw = (1.0f - m.disney[PT_DS_METALLIC]) * (1.0f - m.disney[PT_DS_SPECTRANS]);
            int n = 1 + (m.disney[PT_DS_CLEARCOAT] > 0.0f) + (m.disney[PT_DS_SPECTRANS] > 0.0f) + (thin ? 1 : 0);
            if (dw > 0.0f) n += (thin ? 2 : 1) + 1 + (m.disney[PT_DS_SHEEN] > 0.0f);
            if (n > 5) return fail(PT_ERR_UNSUPPORTED, "disney material with more than 5 BxDFs");
            if (disney_has_bssrdf(m) && d->n_textures && m.tex[PT_MP_KD] >= 0) return fail(PT_ERR_UNSUPPORTED, "disney: a textured color together with scatterdistance");
            if (disney_has_bssrdf(m) && (m.disney_scatter[0] <= 0.0f || m.disney_scatter[1] <= 0.0f || m.disney_scatter[2] <= 0.0f)) return fail(PT_ERR_INVALID_ARG, "disney: scatterdistance must be positive in every channel");
        }
        if (m.type == PT_MAT_MIX) {   // mix.rs:25-50: two plain materials whose lobes fit the five-lobe shade class together
            auto lobes = [](const PtMaterial &q) { switch (q.type) { case PT_MAT_GLASS: return 2; case PT_MAT_PLASTIC: return 2; case PT_MAT_UBER: return 5; case PT_MAT_TRANSLUCENT: return 4; case PT_MAT_DISNEY: return 5; default: return 1; } };
            int total = 0;
            for (int k = 0; k < 2; ++k) {
                if (m.mix[k] >= d->n_materials) return fail(PT_ERR_INVALID_ARG, "mix material index out of range");
                const PtMaterial &q = d->materials[m.mix[k]];
                if (q.type == PT_MAT_MIX || q.type == PT_MAT_SUBSURFACE || disney_has_bssrdf(q)) return fail(PT_ERR_UNSUPPORTED, "mix of mix / subsurface materials");
                total += lobes(q);
            }
            if (total > 5) return fail(PT_ERR_UNSUPPORTED, "mix material with more than 5 BxDFs");
        }
        for (int k = 0; k < 16; ++k) {
            if (d->n_textures == 0 && m.tex[k] > 0) return fail(PT_ERR_INVALID_ARG, "material references a texture but the scene has none");
            if (d->n_textures && m.tex[k] >= (int32_t)d->n_textures) return fail(PT_ERR_INVALID_ARG, "material texture index out of range");
        }
    }
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        const PtLight &L = d->lights[i];
        if (L.type == PT_LIGHT_DIFFUSE_AREA && L.prim >= d->n_prims) return fail(PT_ERR_INVALID_ARG, "area light primitive out of range");
        if (L.type == PT_LIGHT_INFINITE && !d->env_texels) return fail(PT_ERR_INVALID_ARG, "infinite light without env_texels");
    }
    int st = ensure_device();
    if (st) return st;
    pt_scene *sc = new pt_scene();
    sc->device = g_device;
    auto bail = [&](int code) { pt_scene_destroy(sc); return code; };
    DeviceScene &ds = sc->ds;
    // ---- accelerators: one BVH per multi-primitive object (api.rs:1692-1700) + the top-level BVH (adopted or built)
    const bool instanced = d->n_instances > 0 && d->top_refs && d->n_top > 0;
    if (d->n_instances && !instanced) return bail(fail(PT_ERR_INVALID_ARG, "instances given without top_refs"));
    const uint32_t n_top = instanced ? d->n_top : d->n_prims;
    auto prim_bound = [&](uint32_t i, pth::PrimBound &out) {
        if ((d->prim_shape[i] >> 30) == PT_SHAPE_SPHERE) {  // Shape::world_bound = transform_bounds(object_bound) (shape.rs:23-25, sphere.rs:53-57, transform.rs:592-605)
            const PtSphere &S = d->spheres[d->prim_shape[i] & 0x3fffffffu];
            M4 o2w; std::memcpy(o2w.m, S.object_to_world, 64);
            const float lo[3] = {-S.radius, -S.radius, S.z_min}, hi[3] = {S.radius, S.radius, S.z_max};
            const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 1, 1}, {1, 1, 0}, {1, 0, 1}, {1, 1, 1}};
            for (int c = 0; c < 8; ++c) {
                V3 p = xf_point(o2w, V3(corner[c][0] ? hi[0] : lo[0], corner[c][1] ? hi[1] : lo[1], corner[c][2] ? hi[2] : lo[2]));
                const float pc[3] = {p.x, p.y, p.z};
                for (int k = 0; k < 3; ++k) { out.lo[k] = c ? std::fmin(out.lo[k], pc[k]) : pc[k]; out.hi[k] = c ? std::fmax(out.hi[k], pc[k]) : pc[k]; }
            }
            return;
        }
        uint32_t tri = d->prim_shape[i] & 0x3fffffffu;  // Triangle::world_bound (triangle.rs:130-134)
        const float *a = d->P + 3 * (size_t)d->indices[3 * tri], *b = d->P + 3 * (size_t)d->indices[3 * tri + 1], *c = d->P + 3 * (size_t)d->indices[3 * tri + 2];
        for (int k = 0; k < 3; ++k) { out.lo[k] = std::fmin(std::fmin(a[k], b[k]), c[k]); out.hi[k] = std::fmax(std::fmax(a[k], b[k]), c[k]); }
    };
    const uint32_t maxp = d->max_node_prims ? d->max_node_prims : 4;
    if (d->split_method > PT_SPLIT_HLBVH) return bail(fail(PT_ERR_INVALID_ARG, "unknown split_method"));
    // BVHAccel::new (bvh.rs:145-198): SAH on the host (the reference's tree) or HLBVH on the device (gpu_bvh.hip)
    auto build_accel = [&](const std::vector<pth::PrimBound> &pb, std::vector<PtBVHNode> &nodes, std::vector<uint32_t> &ordered) -> int {
        if (d->split_method != PT_SPLIT_HLBVH) { pth::build_sah_bvh(pb, maxp, nodes, ordered); return PT_OK; }
        const char *msg = "HLBVH build failed";
        if (pth::build_hlbvh_gpu(pb, maxp, nodes, ordered, &msg)) return fail(PT_ERR_HIP, msg);
        return PT_OK;
    };
    struct ObjAccel { std::vector<PtBVHNode> nodes; std::vector<uint32_t> ordered; };
    std::vector<ObjAccel> obj(instanced ? d->n_objects : 0);
    if (instanced) {
        for (uint32_t o = 0; o < d->n_objects; ++o) {
            const PtObject &O = d->objects[o];
            if (O.n_prims == 0 || (uint64_t)O.first_prim + O.n_prims > d->n_prims) return bail(fail(PT_ERR_INVALID_ARG, "object primitive range out of bounds"));
            if (O.n_prims == 1) continue;
            std::vector<pth::PrimBound> pb(O.n_prims);
            for (uint32_t i = 0; i < O.n_prims; ++i) prim_bound(O.first_prim + i, pb[i]);
            if ((st = build_accel(pb, obj[o].nodes, obj[o].ordered))) return bail(st);
            for (auto &e : obj[o].ordered) e += O.first_prim;
        }
        for (uint32_t i = 0; i < d->n_instances; ++i) if (d->instances[i].object >= d->n_objects) return bail(fail(PT_ERR_INVALID_ARG, "instance object index out of range"));
        for (uint32_t i = 0; i < n_top; ++i) {
            uint32_t r = d->top_refs[i];
            if ((r & PT_TOP_INSTANCE) ? ((r & ~PT_TOP_INSTANCE) >= d->n_instances) : (r >= d->n_prims)) return bail(fail(PT_ERR_INVALID_ARG, "top_refs entry out of range"));
        }
    }
    auto top_ref = [&](uint32_t pos) { return instanced ? d->top_refs[pos] : pos; };
    if (d->nodes && d->n_nodes && d->ordered_prims) {
        sc->nodes.assign(d->nodes, d->nodes + d->n_nodes);
        sc->ordered.assign(d->ordered_prims, d->ordered_prims + n_top);
        for (uint32_t i = 0; i < n_top; ++i) if (sc->ordered[i] >= n_top) return bail(fail(PT_ERR_INVALID_ARG, "ordered_prims entry out of range"));
        for (uint32_t i = 0; i < d->n_nodes; ++i) {
            const PtBVHNode &n = sc->nodes[i];
            bool ok = n.n_prims ? ((uint64_t)n.offset + n.n_prims <= n_top) : (n.offset < d->n_nodes && i + 1 < d->n_nodes && n.axis < 3);
            if (!ok) return bail(fail(PT_ERR_INVALID_ARG, "malformed BVH node"));
        }
    } else {
        std::vector<pth::PrimBound> pb(n_top);
        for (uint32_t i = 0; i < n_top; ++i) {
            const uint32_t r = top_ref(i);
            if (!(r & PT_TOP_INSTANCE)) { prim_bound(r, pb[i]); continue; }
            // TransformedPrimitive::world_bound = prim_to_world.motion_bounds(inner bound) (primitive.rs:53-55, transform.rs:1564-1567,592-605)
            const PtInstance &I = d->instances[r & ~PT_TOP_INSTANCE];
            const PtObject &O = d->objects[I.object];
            pth::PrimBound inner;
            if (O.n_prims == 1) prim_bound(O.first_prim, inner);
            else for (int k = 0; k < 3; ++k) { inner.lo[k] = obj[I.object].nodes[0].bmin[k]; inner.hi[k] = obj[I.object].nodes[0].bmax[k]; }
            M4 i2w; std::memcpy(i2w.m, I.instance_to_world, 64);
            const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 1, 1}, {1, 1, 0}, {1, 0, 1}, {1, 1, 1}};
            for (int c = 0; c < 8; ++c) {
                V3 p = xf_point(i2w, V3(corner[c][0] ? inner.hi[0] : inner.lo[0], corner[c][1] ? inner.hi[1] : inner.lo[1], corner[c][2] ? inner.hi[2] : inner.lo[2]));
                const float pc[3] = {p.x, p.y, p.z};
                for (int k = 0; k < 3; ++k) { pb[i].lo[k] = c ? std::fmin(pb[i].lo[k], pc[k]) : pc[k]; pb[i].hi[k] = c ? std::fmax(pb[i].hi[k], pc[k]) : pc[k]; }
            }
        }
        if ((st = build_accel(pb, sc->nodes, sc->ordered))) return bail(st);
    }
    // uploads
#define UP(field, src, count) if ((st = sc->upload(&ds.field, src, (size_t)(count)))) return bail(st)
    // Two-wide traversal records (dev_scene.h: WideNode) and the packet order of every accelerator, concatenated:
    // [top level][object 0][object 1]... ; references inside an accelerator are offset by its bases.
    std::vector<uint32_t> leaf_last, packet_refs;
    std::vector<WideNode> wide;
    std::vector<DevInstance> dinst(instanced ? d->n_instances : 0);
    std::vector<QuadNode> quad;
    auto append_accel = [&](const std::vector<PtBVHNode> &nn, const std::vector<uint32_t> &refs, uint32_t &root_ref, uint32_t &root_ref4) {
        const uint32_t wbase = (uint32_t)wide.size(), pbase = (uint32_t)packet_refs.size();
        std::vector<uint32_t> wide_id(nn.size(), 0);
        uint32_t n_int = 0;
        for (size_t i = 0; i < nn.size(); ++i) { if (nn[i].n_prims == 0) wide_id[i] = wbase + n_int++; else leaf_last.push_back(pbase + nn[i].offset + nn[i].n_prims - 1); }
        auto ref_of = [&](uint32_t i) { return nn[i].n_prims ? (kLeafBit | (pbase + nn[i].offset)) : wide_id[i]; };
        wide.resize(wbase + n_int);
        for (size_t i = 0; i < nn.size(); ++i) {
            if (nn[i].n_prims) continue;
            WideNode &w = wide[wide_id[i]];
            const PtBVHNode &l = nn[i + 1], &r = nn[nn[i].offset];
            w.lmin[0] = l.bmin[0]; w.lmin[1] = l.bmin[1]; w.lmin[2] = l.bmin[2]; w.lmax0 = l.bmax[0];
            w.lmax12[0] = l.bmax[1]; w.lmax12[1] = l.bmax[2]; w.rmin01[0] = r.bmin[0]; w.rmin01[1] = r.bmin[1];
            w.rmin2 = r.bmin[2]; w.rmax[0] = r.bmax[0]; w.rmax[1] = r.bmax[1]; w.rmax[2] = r.bmax[2];
            w.left_ref = ref_of((uint32_t)i + 1); w.right_ref = ref_of(nn[i].offset);
            w.meta = nn[i].axis; w.pad = 0;
        }
        // four-wide records of the same tree (dev_scene.h: QuadNode): two binary levels per record, emitted depth first
        {
            const float inf = std::numeric_limits<float>::infinity();
            std::vector<uint32_t> todo;   // binary interior nodes that root a record, in emission order (their record = quad[qbase + position])
            std::vector<uint32_t> quad_id(nn.size(), PT_NONE);
            auto qref_of = [&](uint32_t i) { return nn[i].n_prims ? (kLeafBit | (pbase + nn[i].offset)) : quad_id[i]; };
            const uint32_t qbase = (uint32_t)quad.size();
            if (!nn.empty() && nn[0].n_prims == 0) {
                // pre-order numbering: a record's interior grandchildren root the next records, left to right
                std::vector<uint32_t> stack{0};
                while (!stack.empty()) {
                    const uint32_t i = stack.back(); stack.pop_back();
                    quad_id[i] = qbase + (uint32_t)todo.size(); todo.push_back(i);
                    uint32_t kids[4]; int nk = 0;
                    for (uint32_t c : {(uint32_t)i + 1u, (uint32_t)nn[i].offset}) {
                        if (nn[c].n_prims) continue;
                        kids[nk++] = c + 1u; kids[nk++] = nn[c].offset;
                    }
                    for (int k = nk - 1; k >= 0; --k) if (nn[kids[k]].n_prims == 0) stack.push_back(kids[k]);
                }
            }
            quad.resize(qbase + todo.size());
            for (size_t t = 0; t < todo.size(); ++t) {
                const uint32_t i = todo[t];
                QuadNode &q = quad[qbase + t];
                for (int a = 0; a < 3; ++a) for (int k = 0; k < 4; ++k) { q.lo[a][k] = inf; q.hi[a][k] = -inf; }
                for (int k = 0; k < 4; ++k) q.ref[k] = PT_NONE;
                q.pad[0] = q.pad[1] = q.pad[2] = 0;
                const uint32_t c2[2] = {i + 1u, (uint32_t)nn[i].offset};
                uint32_t axes[3] = {nn[i].axis, 0u, 0u};
                auto put = [&](int slot, uint32_t n) {
                    for (int a = 0; a < 3; ++a) { q.lo[a][slot] = nn[n].bmin[a]; q.hi[a][slot] = nn[n].bmax[a]; }
                    q.ref[slot] = qref_of(n);
                };
                for (int side = 0; side < 2; ++side) {
                    const uint32_t c = c2[side];
                    if (nn[c].n_prims) put(2 * side, c);
                    else { axes[1 + side] = nn[c].axis; put(2 * side, c + 1u); put(2 * side + 1, nn[c].offset); }
                }
                // order word: for each of the eight sign octants o = nx | ny << 1 | nz << 2, three bits at 3 o: bit 0 = the ray is negative along N's
                // axis (the right pair comes first), bit 1 = along L's (slot 1 before slot 0), bit 2 = along R's (slot 3 before slot 2)
                q.meta = 0;
                for (uint32_t o = 0; o < 8; ++o) q.meta |= (((o >> axes[0]) & 1u) | (((o >> axes[1]) & 1u) << 1) | (((o >> axes[2]) & 1u) << 2)) << (3u * o);
            }
            root_ref4 = nn.empty() ? 0u : qref_of(0);
        }
        packet_refs.insert(packet_refs.end(), refs.begin(), refs.end());
        root_ref = ref_of(0);
    };
    {
        std::vector<uint32_t> top_order(n_top);
        for (uint32_t i = 0; i < n_top; ++i) top_order[i] = top_ref(sc->ordered[i]);
        append_accel(sc->nodes, top_order, ds.root_ref, ds.root_ref4);
        ds.n_nodes = (uint32_t)sc->nodes.size();
        for (int k = 0; k < 3; ++k) { ds.root_min[k] = sc->nodes[0].bmin[k]; ds.root_max[k] = sc->nodes[0].bmax[k]; }
        std::vector<uint32_t> obj_root(obj.size(), 0), obj_root4(obj.size(), 0);
        for (size_t o = 0; o < obj.size(); ++o) {
            if (d->objects[o].n_prims == 1) {  // single primitive: a one-packet "leaf" without a BVH
                obj_root[o] = obj_root4[o] = kLeafBit | (uint32_t)packet_refs.size();
                leaf_last.push_back((uint32_t)packet_refs.size());
                packet_refs.push_back(d->objects[o].first_prim);
            } else append_accel(obj[o].nodes, obj[o].ordered, obj_root[o], obj_root4[o]);
        }
        for (size_t i = 0; i < dinst.size(); ++i) {
            const PtInstance &I = d->instances[i]; DevInstance &D = dinst[i];
            std::memcpy(D.world_to_instance, I.world_to_instance, 64); std::memcpy(D.instance_to_world, I.instance_to_world, 64);
            D.single = d->objects[I.object].n_prims == 1; D.root_ref = obj_root[I.object]; D.root_ref4 = obj_root4[I.object];
            for (int k = 0; k < 3; ++k) { D.root_min[k] = D.single ? 0.0f : obj[I.object].nodes[0].bmin[k]; D.root_max[k] = D.single ? 0.0f : obj[I.object].nodes[0].bmax[k]; }
            bool ident = true;
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) ident = ident && I.instance_to_world[4 * r + c] == ((r == c) ? 1.0f : 0.0f);
            D.identity = ident;
        }
        if (wide.size() > (size_t)kRefMask || packet_refs.size() > (size_t)kRefMask) return bail(fail(PT_ERR_UNSUPPORTED, "scene exceeds 2^25 BVH records / packets"));
        if (wide.empty()) wide.resize(1);
        UP(wide, wide.data(), wide.size());
        UP(instances, dinst.data(), dinst.size()); ds.n_instances = (uint32_t)dinst.size();
    }
    UP(P, d->P, 3 * (size_t)d->n_vertices);
    if (d->N) UP(N, d->N, 3 * (size_t)d->n_vertices);
    if (d->S) UP(S, d->S, 3 * (size_t)d->n_vertices);
    if (d->UV) UP(UV, d->UV, 2 * (size_t)d->n_vertices);
    UP(indices, d->indices, 3 * (size_t)d->n_triangles); ds.n_triangles = d->n_triangles;
    {
        std::vector<uint8_t> fl(d->n_triangles, 0);
        if (d->tri_flags) fl.assign(d->tri_flags, d->tri_flags + d->n_triangles);
        for (auto &f : fl) { if (!d->N) f &= ~PT_TRI_HAS_N; if (!d->S) f &= ~PT_TRI_HAS_S; if (!d->UV) f &= ~PT_TRI_HAS_UV; }
        UP(tri_flags, fl.data(), fl.size());
    }
    UP(prim_shape, d->prim_shape, d->n_prims); UP(prim_material, d->prim_material, d->n_prims); UP(prim_light, d->prim_light, d->n_prims);
    ds.n_prims = d->n_prims;
    UP(materials, d->materials, d->n_materials); ds.n_materials = d->n_materials;
    if (d->n_media && d->media) {   // participating media (volpath only)
        UP(media, d->media, d->n_media); ds.n_media = d->n_media;
        std::vector<DevGridAux> aux(d->n_media, DevGridAux{nullptr, 0.0f, 0.0f});
        for (uint32_t i = 0; i < d->n_media; ++i) {   // GridDensityMedium::new (grid.rs:40-72)
            const PtMedium &m = d->media[i];
            if (m.type > PT_MEDIUM_GRID) return bail(fail(PT_ERR_INVALID_ARG, "unknown medium type"));
            if (m.type != PT_MEDIUM_GRID) continue;
            const size_t nvox = (size_t)m.nx * m.ny * m.nz;
            if (!m.density || nvox == 0 || nvox > ((size_t)1 << 31)) return bail(fail(PT_ERR_INVALID_ARG, "grid medium without a density grid"));
            // grid.rs:46-52: `sigma_t = (sigma_a + sigma_s)[0]`; a spectrally varying coefficient is reported with error!() and rendering goes on with the first channel
            for (int k = 1; k < 3; ++k) if (m.sigma_a[k] + m.sigma_s[k] != m.sigma_a[0] + m.sigma_s[0]) { fprintf(stderr, "mi355pt: GridDensityMedium requires spectrally uniform attenuation coefficient (medium %u: using channel 0, as grid.rs:46-52 does)\n", i); break; }
            float maxd = 0.0f;
            for (size_t k = 0; k < nvox; ++k) maxd = std::fmax(maxd, m.density[k]);
            if (!(maxd > 0.0f)) return bail(fail(PT_ERR_INVALID_ARG, "grid medium with no positive density"));
            if ((st = sc->upload(&aux[i].density, m.density, nvox))) return bail(st);
            aux[i].sigma_t = m.sigma_a[0] + m.sigma_s[0]; aux[i].inv_max_density = 1.0f / maxd;
            ds.has_grid = 1u;
        }
        UP(grid_aux, aux.data(), aux.size());
        if (d->prim_medium_inside && d->prim_medium_outside) {
            for (uint32_t i = 0; i < d->n_prims; ++i)
                if ((d->prim_medium_inside[i] != PT_NONE && d->prim_medium_inside[i] >= d->n_media) || (d->prim_medium_outside[i] != PT_NONE && d->prim_medium_outside[i] >= d->n_media))
                    return bail(fail(PT_ERR_INVALID_ARG, "primitive medium index out of range"));
            UP(prim_med_in, d->prim_medium_inside, d->n_prims); UP(prim_med_out, d->prim_medium_outside, d->n_prims);
        }
        sc->class_used[kMediumClass] = true;
    }
    for (uint32_t i = 0; i < d->n_prims; ++i) if (d->prim_material[i] == PT_NONE) sc->has_null_material = true;
    ds.has_shells = sc->has_null_material ? 1u : 0u;
    UP(spheres, d->spheres, d->n_spheres); ds.n_spheres = d->n_spheres;
    UP(lights, d->lights, d->n_lights); ds.n_lights = d->n_lights; sc->n_lights = d->n_lights;
    if (d->n_lights) sc->host_lights.assign(d->lights, d->lights + d->n_lights);
    if (d->env_texels) { sc->env_w = d->env_width; sc->env_h = d->env_height; for (int k = 0; k < 3; ++k) sc->env_texel0[k] = d->env_power_lookup[k]; }
    {
        std::vector<uint8_t> mc(std::max<uint32_t>(1, d->n_materials), 0);
        for (uint32_t i = 0; i < d->n_materials; ++i) { mc[i] = material_class(d->materials[i]); sc->class_used[mc[i]] = true; }
        UP(mat_class, mc.data(), mc.size());
        std::vector<DevBssTable> bt(d->n_bssrdf_tables);
        for (uint32_t i = 0; i < d->n_bssrdf_tables; ++i) {
            const PtBSSRDFTable &t = d->bssrdf_tables[i];
            bt[i].n_rho = (int)t.n_rho; bt[i].n_radius = (int)t.n_radius;
            if (!t.rho_samples || !t.radius_samples || !t.profile || !t.rhoeff || !t.profile_cdf) continue;   // unreferenced slot
            if ((st = sc->upload(&bt[i].rho_samples, t.rho_samples, t.n_rho))) return bail(st);
            if ((st = sc->upload(&bt[i].radius_samples, t.radius_samples, t.n_radius))) return bail(st);
            if ((st = sc->upload(&bt[i].profile, t.profile, (size_t)t.n_rho * t.n_radius))) return bail(st);
            if ((st = sc->upload(&bt[i].rhoeff, t.rhoeff, t.n_rho))) return bail(st);
            if ((st = sc->upload(&bt[i].profile_cdf, t.profile_cdf, (size_t)t.n_rho * t.n_radius))) return bail(st);
        }
        UP(bss_tables, bt.data(), bt.size()); ds.n_bss_tables = d->n_bssrdf_tables;
        if (d->n_textures) {   // textures: nodes as given + one postfix program per node (children before parent)
            std::vector<PtMaterial> mats(d->materials, d->materials + d->n_materials);
            UP(textures, d->textures, d->n_textures); ds.n_textures = d->n_textures;
            std::vector<uint32_t> off(d->n_textures + 1, 0), prog;
            for (uint32_t r = 0; r < d->n_textures; ++r) {
                off[r] = (uint32_t)prog.size();
                // iterative post-order; the value-stack depth is tracked to validate kTexStack
                struct Fr { int node; int next; };
                std::vector<Fr> st{{(int)r, 0}};
                int depth = 0, max_depth = 0; size_t guard = 0;
                while (!st.empty()) {
                    Fr &f = st.back();
                    const PtTexture &t = d->textures[f.node];
                    const int nchild = (t.type == PT_TEX_MIX) ? 3 : (t.type == PT_TEX_SCALE || t.type == PT_TEX_CHECKERBOARD2D || t.type == PT_TEX_CHECKERBOARD3D || t.type == PT_TEX_DOTS) ? 2 : 0;
                    if (f.next < nchild) { const int c = t.child[f.next++]; st.push_back({c, 0}); if (++guard > 4096 || st.size() > 64) return bail(fail(PT_ERR_INVALID_ARG, "texture graph too deep or cyclic")); continue; }
                    prog.push_back((uint32_t)f.node);
                    depth += 1 - nchild; max_depth = std::max(max_depth, depth + nchild);
                    st.pop_back();
                }
                if (max_depth > kTexStack) return bail(fail(PT_ERR_UNSUPPORTED, "texture expression needs a deeper value stack than kTexStack"));
            }
            off[d->n_textures] = (uint32_t)prog.size();
            UP(tex_prog_offset, off.data(), off.size()); UP(tex_prog, prog.data(), prog.size());
            std::vector<DevImage> imgs(d->n_images);
            for (uint32_t i = 0; i < d->n_images; ++i) {
                const PtImage &im = d->images[i];
                imgs[i].width = im.width; imgs[i].height = im.height; imgs[i].n_levels = im.n_levels; imgs[i].channels = im.channels;
                if (!im.texels || im.n_levels == 0 || im.n_levels > 16) continue;   // unreferenced slot
                size_t o = 0;
                for (uint32_t l = 0; l < im.n_levels; ++l) { imgs[i].level_offset[l] = (uint32_t)o; o += (size_t)std::max(1u, im.width >> l) * std::max(1u, im.height >> l) * im.channels; }
                if ((st = sc->upload(&imgs[i].texels, im.texels, o))) return bail(st);
            }
            UP(images, imgs.data(), imgs.size());
            if (d->ewa_weight_lut) UP(ewa_lut, d->ewa_weight_lut, 128);
            auto any_mask = [&](const int32_t *a) { if (!a) return false; for (uint32_t i = 0; i < d->n_triangles; ++i) if (a[i] >= 0) return true; return false; };
            if (any_mask(d->tri_alpha)) UP(tri_alpha, d->tri_alpha, d->n_triangles);
            if (any_mask(d->tri_shadow_alpha)) UP(tri_shadow_alpha, d->tri_shadow_alpha, d->n_triangles);
        }
        for (uint32_t i = 0; i < d->n_materials; ++i) if (d->materials[i].type == PT_MAT_SUBSURFACE || disney_has_bssrdf(d->materials[i])) sc->has_bssrdf = true;
        std::vector<uint32_t> inf;
        for (uint32_t i = 0; i < d->n_lights; ++i) if (d->lights[i].type == PT_LIGHT_INFINITE) inf.push_back(i);
        UP(infinite_lights, inf.data(), inf.size()); ds.n_infinite = (uint32_t)inf.size();
    }
    if (d->env_texels) {  // Distribution2D::new (sampling.rs:100-117) over the importance image
        ds.env_w = d->env_width; ds.env_h = d->env_height;
        UP(env_texels, d->env_texels, 3 * (size_t)ds.env_w * ds.env_h);
        size_t nu = 2 * (size_t)ds.env_w, nv = 2 * (size_t)ds.env_h;
        std::vector<float> func(d->env_importance, d->env_importance + nu * nv), cdf(nv * (nu + 1)), fint(nv), mcdf; float mint;
        for (size_t v = 0; v < nv; ++v) {
            std::vector<float> row(func.begin() + v * nu, func.begin() + (v + 1) * nu), c; float fi;
            dist1d(row, c, fi);
            std::copy(c.begin(), c.end(), cdf.begin() + v * (nu + 1)); fint[v] = fi;
        }
        dist1d(fint, mcdf, mint);
        UP(env_func, func.data(), func.size()); UP(env_cdf, cdf.data(), cdf.size()); UP(env_func_int, fint.data(), fint.size());
        UP(env_marg_func, fint.data(), fint.size()); UP(env_marg_cdf, mcdf.data(), mcdf.size()); ds.env_marg_int = mint;
    }
#undef UP
    // world bound = root node bounds (bvh.rs:697-703); Light::preprocess -> bounding sphere (bounds.rs:516-524)
    for (int k = 0; k < 3; ++k) { ds.wb_min[k] = sc->nodes[0].bmin[k]; ds.wb_max[k] = sc->nodes[0].bmax[k]; }
    {
        float c[3]; bool inside = true;
        for (int k = 0; k < 3; ++k) { c[k] = (ds.wb_min[k] + ds.wb_max[k]) * (1.0f / 2.0f); inside = inside && c[k] >= ds.wb_min[k] && c[k] <= ds.wb_max[k]; }
        float dx = ds.wb_max[0] - c[0], dy = ds.wb_max[1] - c[1], dz = ds.wb_max[2] - c[2];
        ds.world_radius = inside ? std::sqrt(dx * dx + dy * dy + dz * dz) : 0.0f;
        for (int k = 0; k < 3; ++k) ds.world_center[k] = c[k];
    }
    // leaf triangle packets + light areas (device)
    {
        const uint32_t *d_ordered = nullptr;
        const uint32_t n_packets = (uint32_t)packet_refs.size();
        if ((st = sc->upload(&d_ordered, packet_refs.data(), packet_refs.size()))) return bail(st);
        TriPacket *leaf = nullptr; float *area = nullptr; float4 *lrec = nullptr;
        // the four-wide records and the packets share ONE allocation, so that the production traversal addresses both with 32-bit byte offsets from
        // one base (at most 2^24 records x 128 B + 2^25 packets x 48 B < 4 GB): [records][packets + 2] (+2: a packet's fourth quad is loaded with it)
        if (quad.empty()) quad.resize(1);
        const size_t quad_bytes = quad.size() * sizeof(QuadNode), pool_bytes = quad_bytes + ((size_t)n_packets + 2) * sizeof(TriPacket);
        if (pool_bytes >= (size_t)0xE0000000u) return bail(fail(PT_ERR_UNSUPPORTED, "scene exceeds 4 GB of traversal records + packets"));
        uint8_t *pool = nullptr;
        if ((st = sc->dalloc(&pool, pool_bytes))) return bail(st);
        if (hipMemcpy(pool, quad.data(), quad_bytes, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(PT_ERR_HIP, "upload of the four-wide records"));
        leaf = reinterpret_cast<TriPacket *>(pool + quad_bytes);
        ds.quad = reinterpret_cast<const QuadNode *>(pool); ds.leaf_off = (uint32_t)quad_bytes; ds.pool_bytes = (uint32_t)pool_bytes;
        if (hipMemset(leaf, 0, ((size_t)n_packets + 2) * sizeof(TriPacket)) != hipSuccess) return bail(fail(PT_ERR_HIP, "memset"));
        if ((st = sc->dalloc(&area, std::max<uint32_t>(1, d->n_lights)))) return bail(st);
        if ((st = sc->dalloc(&lrec, 6 * (size_t)std::max<uint32_t>(1, d->n_lights)))) return bail(st);
        hipLaunchKernelGGL(k_build_packets, dim3((n_packets + 255) / 256), dim3(256), 0, 0, ds, d_ordered, n_packets, leaf);
        ds.leaf = leaf;
        {
            const uint32_t *d_last = nullptr;
            if ((st = sc->upload(&d_last, leaf_last.data(), leaf_last.size()))) return bail(st);
            hipLaunchKernelGGL(k_mark_leaf_ends, dim3(((uint32_t)leaf_last.size() + 255) / 256), dim3(256), 0, 0, leaf, d_last, (uint32_t)leaf_last.size());
        }
        if (d->n_lights) hipLaunchKernelGGL(k_light_area, dim3((d->n_lights + 255) / 256), dim3(256), 0, 0, ds, area, lrec);
        ds.light_area = area; ds.light_rec = lrec;
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return bail(fail(PT_ERR_HIP, "scene preparation kernels failed"));
    }
    *out = sc;
    return PT_OK;
}

void pt_scene_destroy(pt_scene *sc) {
    if (!sc) return;
    if (sc->device != g_device) bind_device(sc->device);
    for (void *p : sc->allocs) hipFree(p);
    if (sc->slab) hipFree(sc->slab);
    if (sc->qbuf) hipFree(sc->qbuf);
    if (sc->bss_slab) hipFree(sc->bss_slab);
    if (sc->ext_slab) hipFree(sc->ext_slab);
    if (sc->film_rgbw) hipFree(sc->film_rgbw);
    sc->drop_timings();
    for (auto e : sc->event_pool) hipEventDestroy(e);
    if (sc->stream) hipStreamDestroy(sc->stream);
    delete sc;
}

int pt_scene_bvh_info(const pt_scene *sc, uint32_t *n_nodes, uint32_t *n_prims) {
    if (!sc || !n_nodes || !n_prims) return fail(PT_ERR_INVALID_ARG, "null argument");
    *n_nodes = (uint32_t)sc->nodes.size(); *n_prims = (uint32_t)sc->ordered.size();
    return PT_OK;
}
int pt_scene_bvh_read(const pt_scene *sc, PtBVHNode *nodes, uint32_t *ordered) {
    if (!sc || !nodes || !ordered) return fail(PT_ERR_INVALID_ARG, "null argument");
    std::memcpy(nodes, sc->nodes.data(), sc->nodes.size() * sizeof(PtBVHNode));
    std::memcpy(ordered, sc->ordered.data(), sc->ordered.size() * 4);
    return PT_OK;
}

int pt_render(pt_scene *sc, const PtRenderParams *rp, float *film_xyzw, int film_is_device) {
    if (!sc || !rp || !film_xyzw) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (rp->spp == 0) return fail(PT_ERR_INVALID_ARG, "spp must be > 0");
    if (!(rp->filter_radius[0] > 0.0f) || !(rp->filter_radius[1] > 0.0f)) return fail(PT_ERR_INVALID_ARG, "filter radius must be > 0");
    if (rp->tile_world > 1 && rp->tile_rank >= rp->tile_world) return fail(PT_ERR_INVALID_ARG, "tile_rank >= tile_world");
    if (sc->device != g_device) { int bst = bind_device(sc->device); if (bst) return bst; }
    RenderConst rc;
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
    const size_t film_px = (size_t)rc.film_w * rc.film_h;
    sc->profile = rp->profile != 0;
    sc->drop_timings();
    sc->stats.clear();
    int st = PT_OK;
    if (rc.n_pix_slots > 0) {
        uint32_t S = rp->spp_per_pass;
        if (S == 0) S = choose_pass_size(sc, rc.n_pix_slots, rp->spp, 1, rc.volpath != 0);
        S = std::min(S, rp->spp);
        if ((size_t)rc.n_pix_slots * S > ((size_t)1 << 31)) return fail(PT_ERR_INVALID_ARG, "pass too large");
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
    if (h.error) return fail((int)h.error, "traversal error raised on device");
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
    for (uint32_t i = 0; i < n_devices; ++i) {
        if ((st = bind_device(device_ordinals[i]))) return bail(st);
        pt_scene *sc = nullptr;
        if ((st = pt_scene_create(&d, &sc))) return bail(st);
        ms->sc.push_back(sc); ms->dev.push_back(device_ordinals[i]);
        ms->film.push_back(nullptr); ms->film_cap.push_back(0); ms->stage.push_back(nullptr); ms->stage_cap.push_back(0);
        ms->render_ms.push_back(0); ms->copy_ms.push_back(0);
        if (i == 0) {   // the replicas adopt the first replica's top-level tree instead of building it again
            d.nodes = sc->nodes.data(); d.n_nodes = (uint32_t)sc->nodes.size(); d.ordered_prims = sc->ordered.data();
        }
    }
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
        pt_scene_destroy(ms->sc[i]);
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
