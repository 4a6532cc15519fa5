// gather_bench.hip -- calibration of rocprofv3's FETCH_SIZE for the access pattern of the path tracer's traversal: every lane reads
// ONE random, 64-byte-aligned record of a large table with dwordx4 loads (16, 32, 48 or 64 bytes of it), nothing coalesces across lanes.
// Known request bytes (lanes x iterations x bytes) are printed next to the kernel name so that FETCH_SIZE / TCC_EA0_RDREQ of the same
// dispatch can be compared with them:   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./gather_bench <table MB> <iterations>
// (is the x2 the guide gives for wide streaming reads also right for 64-byte gathers?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int QUADS> __global__ void k_gather(const uint4 *table, uint32_t n_records, uint32_t iters, uint32_t *out) {
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t rec = (uint32_t)(((uint64_t)(x >> 4) * n_records) >> 28);   // uniform in [0, n_records)
        const uint4 *p = table + 4 * (size_t)rec;
#pragma unroll
        for (int q = 0; q < QUADS; ++q) { const uint4 v = p[q]; acc += v.x ^ v.y ^ v.z ^ v.w; }
        x ^= acc & 1u;   // the next address depends on the data: one dependent gather per iteration, like a BVH step
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_stream(const uint4 *table, size_t n_quads, uint32_t *out) {   // reference: wide coalesced streaming read (the guide's x2 case)
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_quads; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = table[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 200;
    const uint32_t iters = argc > 2 ? (uint32_t)atoi(argv[2]) : 256;
    const size_t bytes = mb << 20; const uint32_t n_records = (uint32_t)(bytes / 64);
    uint4 *table; uint32_t *out;
    const uint32_t blocks = 256 * 8, threads = 256;
    CHECK(hipMalloc(&table, bytes)); CHECK(hipMalloc(&out, (size_t)blocks * threads * 4));
    CHECK(hipMemset(table, 1, bytes));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    auto run = [&](const char *name, int quads) {
        CHECK(hipEventRecord(a));
        if (quads == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(threads), 0, 0, table, n_records, iters, out);
        else if (quads == 2) hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(threads), 0, 0, table, n_records, iters, out);
        else if (quads == 3) hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(threads), 0, 0, table, n_records, iters, out);
        else if (quads == 4) hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(threads), 0, 0, table, n_records, iters, out);
        else hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(threads), 0, 0, table, bytes / 16, out);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
        const double req = quads ? (double)blocks * threads * iters * 16.0 * quads : (double)bytes;
        const double lines = quads ? (double)blocks * threads * iters * 64.0 : (double)bytes;
        printf("{\"kernel\": \"%s\", \"table_MB\": %zu, \"requested_bytes\": %.0f, \"touched_64B_lines_bytes\": %.0f, \"ms\": %.3f, \"requested_GBs\": %.1f}\n", name, mb, req, lines, ms, req / ms / 1e6);
    };
    run("k_stream", 0); run("k_stream", 0);
    run("k_gather<1>", 1); run("k_gather<2>", 2); run("k_gather<3>", 3); run("k_gather<4>", 4);
    return 0;
}
