// host_device.hip -- the library's process / thread state: device binding, Sobol' / Halton tables per device, pt_init and the knobs (host_common.h has the map).
#include "host_common.h"

extern "C" const unsigned char pt_sobol_blob[];   // tables_blob.cpp (.incbin of data/sobol_tables.bin)
extern "C" const unsigned int pt_sobol_blob_size;

namespace pth {
thread_local std::string g_error;
thread_local int g_device = -1;
thread_local int g_num_cus = 256;
std::atomic<int> g_default_device{-1};
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
uint32_t g_test_pool_pad_records = 0;              // TEST HOOK (env PT_TEST_POOL_PAD_RECORDS): that many unused 128-byte records in front of every scene's record / packet pool, so that a small
                                                   // scene's records and packets lie beyond the 4 GB a 32-bit byte offset reaches (tests/test_gpu_parity.py: the production walk addresses 16-byte quads)
int g_test_max_iterations = 0;                     // TEST HOOK (env PT_TEST_MAX_ITERATIONS, read at every pt_render): the cap on wavefront iterations per pass (2^20 otherwise), so that the PT_ERR_PROBE_CHAIN return of a pass that does not end can be tested
bool g_film_final = true;                          // env PT_FILM_FINAL=0: the per-iteration k_shade_miss pass instead of ending the paths in the film kernel (plain path integrator without subsurface materials)
bool g_shade_specialise = true;                        // env PT_SHADE_SPECIALISE=0: every shade class runs its general kernel (no per-scene lobe-set forms)
uint32_t g_inst_quorum = 16;                      // lanes waiting for the instance transform step (env PT_TRACE_INST_QUORUM)
uint32_t g_trace_waves_per_cu = 28;               // persistent trace waves per CU the grid (and the per-wave HBM stack slab) is sized for: 7 per SIMD, what the exact walk of triangle-only scenes fits; the production walk fits 5 (4 with instances) and its surplus blocks start as the first ones drain (env PT_TRACE_WAVES_PER_CU; 20 / 24 / 28: the same, profiles/r4/NOTES.md)
thread_local SobolTables g_tabs = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
DevCtx g_ctx[kMaxDevices];
std::mutex g_ctx_mutex;
int fail(int code, const std::string &msg) { g_error = msg; return code; }

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

}  // namespace pth

extern "C" {

int pt_init(int device_ordinal) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(PT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(PT_ERR_INVALID_ARG, "device ordinal out of range");
    if (const char *e = getenv("PT_TRACE_REFILL_MIN")) { int a = 0, b = 0, c = 0, d = 0; int n = sscanf(e, "%d,%d,%d,%d", &a, &b, &c, &d); if (n == 1) b = c = d = a; if (n == 3) d = a; if (n >= 1) { g_refill_min[0] = a; g_refill_min[1] = b; g_refill_min[2] = c; g_refill_min[3] = d; g_refill_from_env = true; } }
    if (const char *e = getenv("PT_TRACE_SPLIT")) g_trace_split = atoi(e) != 0;
    if (const char *e = getenv("PT_TRACE_EXACT")) g_trace_exact = atoi(e) != 0;
    if (const char *e = getenv("PT_SHADE_SPECIALISE")) g_shade_specialise = atoi(e) != 0;
    if (const char *e = getenv("PT_FILM_FINAL")) g_film_final = atoi(e) != 0;
    if (const char *e = getenv("PT_TEST_POOL_PAD_RECORDS")) { const long long v = atoll(e); if (v >= 0 && v < (1ll << 27)) g_test_pool_pad_records = (uint32_t)v; }
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

}  // extern "C"
