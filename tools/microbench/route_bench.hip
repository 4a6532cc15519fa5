// route_bench.hip -- where is the floor of k_route (csrc/kern_misc.h)? The production kernel reads a queue entry (path id), gathers ONE word of the path's 32-byte hit
// record (its shade class) and appends the path id to the class's queue through block-level LDS staging. This bench runs the same loop over a synthetic queue of sorted,
// half-dense path ids with the class taken (a) from the 32-byte records by path id (production), (b) from a byte array by path id (round 3's experiment), (c) from a byte
// array by QUEUE POSITION (streaming: what a class byte written by k_trace per queue entry would give), (d) from the path id's low bit (no second read at all).
//   ./route_bench [entries in millions = 256] [blocks per CU = 6]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int CAP> struct LdsQueue { uint32_t count; uint32_t base; uint32_t buf[CAP]; };
template <int CAP> __device__ void lq_init(LdsQueue<CAP> &q) { if (threadIdx.x == 0) { q.count = 0; q.base = 0; } }
template <int CAP> __device__ void lq_push(LdsQueue<CAP> &q, uint32_t value, bool pred) {
    unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    uint32_t lane = __lane_id();
    uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&q.count, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    if (pred) q.buf[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = value;
}
template <int CAP> __device__ void lq_flush_nosync(LdsQueue<CAP> &q, uint32_t *gcount, uint32_t *gbuf, uint32_t reserve, bool force) {
    const uint32_t n = q.count;
    if (n != 0 && (force || n + reserve > (uint32_t)CAP)) {
        if (threadIdx.x == 0) q.base = atomicAdd(gcount, n);
        __syncthreads();
        const uint32_t b = q.base;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) gbuf[b + i] = q.buf[i];
        __syncthreads();
        if (threadIdx.x == 0) q.count = 0;
    }
}

// MODE 0: class from hit[pid * 8 + 7] (32-byte records); 1: from cls_pid[pid]; 2: from cls_pos[qi]; 3: pid & 1
template <int MODE, int CAP> __global__ __launch_bounds__(256) void k_route_bench(const uint32_t *queue, uint32_t count, const uint32_t *hit, const uint8_t *cls_pid, const uint8_t *cls_pos,
                                                                               uint32_t *class_count, uint32_t *c0, uint32_t *c1, uint32_t *c2, uint32_t *c3, uint32_t *c4, uint32_t *c6) {
    __shared__ LdsQueue<CAP> q0, q1, q2, q3, q4, q6;
    lq_init(q0); lq_init(q1); lq_init(q2); lq_init(q3); lq_init(q4); lq_init(q6);
    __syncthreads();
    const uint32_t rounded = (count + 255u) & ~255u;
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < rounded; qi += gridDim.x * blockDim.x) {
        const bool valid = qi < count;
        uint32_t pid = 0, cls = 4u;
        if (valid) {
            pid = queue[qi];
            if (MODE == 0) cls = (hit[(size_t)pid * 8 + 7] >> 13) & 7u;
            else if (MODE == 1) cls = cls_pid[pid];
            else if (MODE == 2) cls = cls_pos[qi];
            else cls = (pid & 1u) ? 0u : 4u;
        }
        lq_push(q0, pid, valid && cls == 0u); lq_push(q1, pid, valid && cls == 1u);
        lq_push(q2, pid, valid && cls == 2u); lq_push(q3, pid, valid && cls == 3u); lq_push(q4, pid, valid && cls == 4u);
        lq_push(q6, pid, valid && cls == 6u);
        __syncthreads();
        lq_flush_nosync(q0, class_count + 0, c0, 256u, false); lq_flush_nosync(q1, class_count + 1, c1, 256u, false);
        lq_flush_nosync(q2, class_count + 2, c2, 256u, false); lq_flush_nosync(q3, class_count + 3, c3, 256u, false);
        lq_flush_nosync(q4, class_count + 4, c4, 256u, false); lq_flush_nosync(q6, class_count + 6, c6, 256u, false);
        __syncthreads();
    }
    lq_flush_nosync(q0, class_count + 0, c0, 0u, true); lq_flush_nosync(q1, class_count + 1, c1, 0u, true);
    lq_flush_nosync(q2, class_count + 2, c2, 0u, true); lq_flush_nosync(q3, class_count + 3, c3, 0u, true);
    lq_flush_nosync(q4, class_count + 4, c4, 0u, true); lq_flush_nosync(q6, class_count + 6, c6, 0u, true);
}

// three staging queues of CAP entries (scenes that use at most three shade classes): classes 0, 4, 6
template <int MODE, int CAP> __global__ __launch_bounds__(256) void k_route_bench3(const uint32_t *queue, uint32_t count, const uint32_t *hit, const uint8_t *cls_pid, const uint8_t *cls_pos,
                                                                                uint32_t *class_count, uint32_t *c0, uint32_t *c4, uint32_t *c6) {
    __shared__ LdsQueue<CAP> q0, q4, q6;
    lq_init(q0); lq_init(q4); lq_init(q6);
    __syncthreads();
    const uint32_t rounded = (count + 255u) & ~255u;
    for (uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x; qi < rounded; qi += gridDim.x * blockDim.x) {
        const bool valid = qi < count;
        uint32_t pid = 0, cls = 4u;
        if (valid) {
            pid = queue[qi];
            if (MODE == 0) cls = (hit[(size_t)pid * 8 + 7] >> 13) & 7u;
            else if (MODE == 2) cls = cls_pos[qi];
            else cls = (pid & 1u) ? 0u : 4u;
        }
        lq_push(q0, pid, valid && cls == 0u); lq_push(q4, pid, valid && cls == 4u); lq_push(q6, pid, valid && cls == 6u);
        __syncthreads();
        lq_flush_nosync(q0, class_count + 0, c0, 256u, false); lq_flush_nosync(q4, class_count + 4, c4, 256u, false); lq_flush_nosync(q6, class_count + 6, c6, 256u, false);
        __syncthreads();
    }
    lq_flush_nosync(q0, class_count + 0, c0, 0u, true); lq_flush_nosync(q4, class_count + 4, c4, 0u, true); lq_flush_nosync(q6, class_count + 6, c6, 0u, true);
}

__global__ void k_fill(uint32_t *queue, uint32_t n, uint32_t *hit, uint8_t *cls_pid, uint8_t *cls_pos) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15;
        const uint32_t pid = 2u * (uint32_t)i + (h & 1u);   // sorted, half dense
        const uint32_t cls = (h & 2u) ? 0u : 4u;           // matte / miss, about half each
        queue[i] = pid; hit[(size_t)pid * 8 + 7] = cls << 13; cls_pid[pid] = (uint8_t)cls; cls_pos[i] = (uint8_t)cls;
    }
}

int main(int argc, char **argv) {
    const uint32_t n = (uint32_t)((argc > 1 ? atol(argv[1]) : 256) * 1000000L);
    const uint32_t bpc = argc > 2 ? (uint32_t)atoi(argv[2]) : 6;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t cus = (uint32_t)prop.multiProcessorCount;
    uint32_t *queue, *hit, *cc, *out[6]; uint8_t *cls_pid, *cls_pos;
    CHECK(hipMalloc(&queue, (size_t)n * 4)); CHECK(hipMalloc(&hit, (size_t)n * 2 * 32)); CHECK(hipMalloc(&cls_pid, (size_t)n * 2)); CHECK(hipMalloc(&cls_pos, n));
    CHECK(hipMalloc(&cc, 64)); for (auto &o : out) CHECK(hipMalloc(&o, (size_t)n * 4));
    CHECK(hipMemset(hit, 0, (size_t)n * 2 * 32));
    hipLaunchKernelGGL(k_fill, dim3(cus * 8), dim3(256), 0, 0, queue, n, hit, cls_pid, cls_pos);
    CHECK(hipDeviceSynchronize());
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    auto run = [&](const char *name, int mode, int cap, uint32_t blocks_per_cu) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(cc, 0, 64));
            const dim3 g(cus * blocks_per_cu), t(256);
            CHECK(hipEventRecord(a));
#define L(M, C) hipLaunchKernelGGL((k_route_bench<M, C>), g, t, 0, 0, queue, n, hit, cls_pid, cls_pos, cc, out[0], out[1], out[2], out[3], out[4], out[5])
            if (cap == 2048) { if (mode == 0) L(0, 2048); else if (mode == 2) L(2, 2048); else L(3, 2048); }
            else if (cap == 1024) { if (mode == 0) L(0, 1024); else if (mode == 1) L(1, 1024); else if (mode == 2) L(2, 1024); else L(3, 1024); }
            else { if (mode == 0) L(0, 512); else if (mode == 1) L(1, 512); else if (mode == 2) L(2, 512); else L(3, 512); }
#undef L
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        uint32_t h[16]; CHECK(hipMemcpy(h, cc, 64, hipMemcpyDeviceToHost));
        printf("{\"class_from\": \"%s\", \"entries\": %u, \"queue_cap\": %d, \"blocks_per_cu\": %u, \"ms\": %.3f, \"ps_per_entry\": %.2f, \"matte\": %u, \"miss\": %u}\n", name, n, cap, blocks_per_cu, best, best * 1e9 / n, h[0], h[4]);
        fflush(stdout);
    };
    for (uint32_t b2 : {bpc, 2 * bpc}) {
        run("hit record by path id (production)", 0, 1024, b2);
        run("byte array by path id", 1, 1024, b2);
        run("byte array by queue position", 2, 1024, b2);
        run("path id bit (no second read)", 3, 1024, b2);
    }
    auto run3 = [&](const char *name, int mode, int cap, uint32_t blocks_per_cu) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(cc, 0, 64));
            const dim3 g(cus * blocks_per_cu), t(256);
            CHECK(hipEventRecord(a));
#define L3(M, C) hipLaunchKernelGGL((k_route_bench3<M, C>), g, t, 0, 0, queue, n, hit, cls_pid, cls_pos, cc, out[0], out[4], out[5])
            if (cap == 2048) { if (mode == 0) L3(0, 2048); else L3(3, 2048); }
            else if (cap == 4096) { if (mode == 0) L3(0, 4096); else L3(3, 4096); }
            else { if (mode == 0) L3(0, 1024); else L3(3, 1024); }
#undef L3
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
        }
        uint32_t h[16]; CHECK(hipMemcpy(h, cc, 64, hipMemcpyDeviceToHost));
        printf("{\"class_from\": \"%s\", \"queues\": 3, \"entries\": %u, \"queue_cap\": %d, \"blocks_per_cu\": %u, \"ms\": %.3f, \"ps_per_entry\": %.2f, \"matte\": %u, \"miss\": %u}\n", name, n, cap, blocks_per_cu, best, best * 1e9 / n, h[0], h[4]);
        fflush(stdout);
    };
    run("hit record by path id (production)", 0, 2048, 3); run("byte array by queue position", 2, 2048, 3); run("path id bit (no second read)", 3, 2048, 3);
    run("hit record by path id (production)", 0, 2048, 6);
    run3("hit record by path id (production)", 0, 1024, 12); run3("hit record by path id (production)", 0, 2048, 6); run3("hit record by path id (production)", 0, 4096, 3);
    run3("path id bit (no second read)", 3, 1024, 12); run3("path id bit (no second read)", 3, 2048, 6); run3("path id bit (no second read)", 3, 4096, 3);
    run("byte array by queue position", 2, 512, 12);
    run("path id bit (no second read)", 3, 512, 12);
    return 0;
}
