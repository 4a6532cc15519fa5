// host_common.h -- what the host driver's translation units share: the library's process / thread state, the scene objects behind the C ABI's opaque
// handles, and the helpers that cross files. The driver used to be one file (capi.hip, 1 700 lines); it is split by what a reader looks for:
//   host_device.hip   device binding per host thread, Sobol' / Halton tables, pt_init and the process-wide knobs
//   scene_create.hip  pt_scene_create: validation, accelerators (host SAH / GPU HLBVH), two- and four-wide records, uploads, light records
//   render_loop.hip   the wavefront scheduler: workspace, light grids, launch_trace / launch_shade, run_pass, pt_render, counters and kernel stats
//   parity_api.hip    the entry points tests use to compare single stages with the oracle (rays, Sobol' / Halton samples, camera rays)
//   multi_device.hip  pt_multi_*: one process driving several devices (one host thread + stream per replica, peer-copy film merge)
#pragma once
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

namespace pth {

// Device binding is per host thread (hipSetDevice is): every thread that enters the library is bound to one device, whose Sobol' /
// Halton tables and CU count it sees through these thread-local views of the per-device contexts below. pt_init selects the
// process-wide default; a scene remembers the device it was created on and re-binds the calling thread when needed, so one
// process can drive several GPUs (pt_multi_*: one host thread + stream per device).
extern thread_local std::string g_error;
extern thread_local int g_device;
extern thread_local int g_num_cus;
extern std::atomic<int> g_default_device;
// traversal scheduling knobs (env PT_TRACE_REFILL_MIN / PT_TRACE_LEAF_QUORUM override)
extern uint32_t g_refill_min[4], g_leaf_quorum[4];
extern bool g_refill_from_env;
extern bool g_trace_split;
extern bool g_trace_exact;
extern uint32_t g_inst_quorum;
extern bool g_shade_specialise;
extern bool g_film_final;
extern int g_test_max_iterations;
extern uint32_t g_trace_waves_per_cu;
extern uint32_t g_test_pool_pad_records;
extern thread_local SobolTables g_tabs;
constexpr int kMaxDevices = kMaxReplicas;
struct DevCtx { bool ready = false; int num_cus = 256; SobolTables tabs = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; };
extern DevCtx g_ctx[kMaxDevices];
extern std::mutex g_ctx_mutex;

int fail(int code, const std::string &msg);
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
int ensure_device();

struct Stat { std::string name, kernel; uint64_t launches = 0; double ms = 0; uint64_t items = 0, nodes = 0, tris = 0; };
struct TimedLaunch { int stat; hipEvent_t a, b; bool closed; };

}  // namespace pth
using namespace pth;

struct pt_scene {
    int device = 0;                    // the HIP device every allocation of this scene lives on
    std::vector<void *> allocs;
    DeviceScene ds{};
    std::vector<PtBVHNode> nodes;
    std::vector<uint32_t> ordered;
    bool pool_big = false;          // four-wide records + packets beyond 4 GB: k_trace<.., 2> (64-bit addresses)
    bool quad_walk_only = false;    // more than 2^25 two-wide records or packets: the scene has no exact (two-wide) walk, pt_set_trace_exact(1) renders are refused
    bool exact_walk_only = false;   // an adopted top-level tree whose child boxes do not nest: the two-wide walk tests every box like the reference (scene_create.hip)
    bool class_used[kNumClasses] = {true, false, false, false, true, false, false, false, false, false, false};   // shade classes the scene's materials map to (kernels.h: kNumClasses; scene_create.hip: material_class)
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
                      size_t ncell = 0, stride = 0; uint64_t filled = 0;
                      float *arena = nullptr; size_t arena_left = 0; } lazy;   // (arena: the blocks of newly touched voxels are carved from 64 MB slabs, not allocated one fill at a time)
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
    std::vector<double> create_ms; double create_wall_ms = 0;   // pt_multi_scene_create: per replica its pt_scene_create, and the whole call (replicas 1.. are created concurrently)
    double merge_ms = 0;              // last pt_multi_render: from the last replica's render end to the summed film (copy tails + the sum kernel)
    PtCounters counters{};
};

namespace pth {
// scene_create.hip
uint8_t material_class(const PtMaterial &m, bool specialise, bool untextured);
void dist1d(const std::vector<float> &func, std::vector<float> &cdf, float &func_int);
// render_loop.hip
int launch_trace(pt_scene *sc, int any, TraceJob job, uint32_t n_upper, bool probe = false);
const char *device_error_text(uint32_t code);
int ensure_workspace(pt_scene *sc, size_t capacity, size_t film_px);
int ensure_light_grid(pt_scene *sc, int requested, int &effective);
void fill_render_const(const PtRenderParams *rp, RenderConst &rc);
uint32_t choose_pass_size(const pt_scene *sc, uint32_t n_pix_slots, uint32_t spp, uint32_t share, bool volpath);
int run_pass(pt_scene *sc, RenderConst &rc, const LightGrid &grid, bool rp_profile_exact);
void read_counters(pt_scene *sc);
}  // namespace pth
