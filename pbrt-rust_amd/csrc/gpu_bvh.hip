// gpu_bvh.hip -- HLBVH construction on the GPU: Morton codes, LSD radix sort, one Karras pass for
// every LBVH treelet at once, bottom-up bounds, and SAH upper levels over the (<= 4096) treelet roots.
// Selected with PtSceneDesc.split_method = PT_SPLIT_HLBVH (accelerator "bvh" "string splitmethod" "hlbvh").
//
// What it restates (behaviour, not code): BVHAccel::hlbvh_build / emit_lbvh / build_upper_sah
// (accelerators/bvh.rs:377-660), encode_morton3 / left_shift3 (:832-857), radix_sort (:859-912).
// The reference's own HLBVH path cannot serve as a parity target: emit_lbvh's leaf loop walks the whole remaining
// slice instead of n_primitives entries (bvh.rs:488-493) and ordered_prims is filled through an atomic offset, so
// the primitive order depends on thread timing (bvh.rs:424-455). This builder produces the tree the algorithm
// describes, deterministically; oracle/ref_hlbvh.h restates it on the CPU and the tests require the two trees to be
// bit-identical, and rendered hits to equal those of the SAH tree. Deliberate choices where the reference leaves
// room:
//   * ordered_prims is the Morton order (stable: equal codes keep creation order);
//   * a range becomes a leaf when it holds <= max_node_prims primitives (emit_lbvh tests `<`), so leaves are as
//     large as the SAH builder's;
//   * primitives with identical 30-bit codes are split by position (highest differing bit of their sorted
//     positions, axis 0) instead of forming one leaf of unbounded size (LinearBVHNode.n_primitives is a u16);
//   * build_upper_sah partitions stably and falls back to the middle when the bucket split leaves a side empty
//     (the reference asserts centroid_bounds.max != min, bvh.rs:588).
// Output is the same LinearBVHNode array (depth-first, first child at i+1) the SAH builder returns, so the rest
// of scene creation, pt_scene_bvh_read and the oracle's "adopted accelerator" path are unchanged.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include "host_bvh.h"

namespace pth {
namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kLeafTag = 0x80000000u;   // child / treelet-root reference: leaf at sorted position (ref & ~tag)
constexpr int kTreeletBits = 12, kCodeBits = 30, kBins = 1 << kTreeletBits;
constexpr int kSortTile = 2048;              // keys per one-wave block in the radix passes

__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
inline float ord2f(uint32_t u) { u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; std::memcpy(&f, &u, 4); return f; }

// ---- 1. centroid bounds: Bounds3f over `.5 * pMin + .5 * pMax` of every primitive (bvh.rs:386-390)
__global__ void k_centroid_bounds(const PrimBound *pb, uint32_t n, uint32_t *cb /* [6] ordered-uint min xyz, max xyz */) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int k = 0; k < 3; ++k) { const float c = 0.5f * pb[i].lo[k] + 0.5f * pb[i].hi[k]; lo[k] = fminf(lo[k], c); hi[k] = fmaxf(hi[k], c); }
    for (int k = 0; k < 3; ++k) {
        for (int o = 32; o > 0; o >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], o)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&cb[k], f2ord(lo[k])); atomicMax(&cb[3 + k], f2ord(hi[k])); }
    }
}

__device__ __forceinline__ uint32_t left_shift3(uint32_t x) {  // bvh.rs:832-848
    if (x == (1u << 10)) x -= 1;
    x = (x | (x << 16)) & 0x30000ffu;
    x = (x | (x << 8)) & 0x300f00fu;
    x = (x | (x << 4)) & 0x30c30c3u;
    x = (x | (x << 2)) & 0x9249249u;
    return x;
}

// ---- 2. Morton codes: encode_morton3(bounds.offset(centroid) * 1024) (bvh.rs:392-401,850-857)
__global__ void k_morton(const PrimBound *pb, uint32_t n, const float *cbf /* min xyz, max xyz */, uint32_t *code, uint32_t *index) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float c = 0.5f * pb[i].lo[k] + 0.5f * pb[i].hi[k];
        float o = c - cbf[k];
        if (cbf[3 + k] > cbf[k]) o /= cbf[3 + k] - cbf[k];
        const float v = o * 1024.0f;
        q[k] = (v > 0.0f) ? (v >= 1024.0f ? 1024u : (uint32_t)v) : 0u;   // `as u32` saturates, NaN -> 0
    }
    code[i] = (left_shift3(q[2]) << 2) | (left_shift3(q[1]) << 1) | left_shift3(q[0]);
    index[i] = i;
}

// ---- 3. LSD radix sort, 8 bits per pass, stable (bvh.rs:859-912 sorts 6 bits per pass; the result is the same
//         permutation). One wave per tile, so ranks inside a tile need no block barriers.
__global__ __launch_bounds__(64) void k_sort_hist(const uint32_t *key, uint32_t n, int shift, uint32_t n_tiles, uint32_t *hist /* [256][n_tiles] */) {
    __shared__ uint32_t h[256];
    for (int d = threadIdx.x; d < 256; d += 64) h[d] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
    for (int r = 0; r < kSortTile / 64; ++r) {
        const uint32_t i = base + r * 64 + threadIdx.x;
        if (i < n) atomicAdd(&h[(key[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < 256; d += 64) hist[(size_t)d * n_tiles + blockIdx.x] = h[d];
}

// exclusive scan of `m` counters in place (digit-major, so tile t's digit-d run starts at scan[d][t]); one block
__global__ __launch_bounds__(1024) void k_sort_scan(uint32_t *hist, uint32_t m) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (m + 1023) / 1024, lo = threadIdx.x * per, hi = min(m, lo + per);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += hist[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan of the 1024 partial sums
        const uint32_t v = (threadIdx.x >= (uint32_t)o) ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t c = hist[i]; hist[i] = run; run += c; }
}

__global__ __launch_bounds__(64) void k_sort_scatter(const uint32_t *key_in, const uint32_t *val_in, uint32_t n, int shift, uint32_t n_tiles,
                                                     const uint32_t *scan, uint32_t *key_out, uint32_t *val_out) {
    __shared__ uint32_t next[256];   // where the tile's next key with digit d goes
    for (int d = threadIdx.x; d < 256; d += 64) next[d] = scan[(size_t)d * n_tiles + blockIdx.x];
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile, lane = threadIdx.x;
    for (int r = 0; r < kSortTile / 64; ++r) {
        const uint32_t i = base + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t k = valid ? key_in[i] : 0u, v = valid ? val_in[i] : 0u, d = (k >> shift) & 255u;
        unsigned long long peers = __ballot(valid);   // lanes of this round holding the same digit
        for (int b = 0; b < 8; ++b) { const unsigned long long bal = __ballot(valid && ((d >> b) & 1u)); peers &= ((d >> b) & 1u) ? bal : ~bal; }
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull)), cnt = (uint32_t)__popcll(peers);
        uint32_t pos = 0;
        if (valid) pos = next[d] + rank;
        __syncthreads();
        if (valid && rank + 1 == cnt) next[d] += cnt;
        __syncthreads();
        if (valid) { key_out[pos] = k; val_out[pos] = v; }
    }
}

// ---- 4. Karras 2012: every internal node of the binary radix tree over the keys (code << 32 | position), in parallel.
//         Restricted to one treelet the tree is exactly emit_lbvh's recursion: split at the highest differing bit.
struct Tree {
    uint32_t n;
    const uint32_t *code;     // sorted
    uint32_t *left, *right;   // [n-1] child references (internal index, or kLeafTag | position)
    uint32_t *parent;         // [2n-1]: internal nodes then leaves; kNone for the root
    uint32_t *first, *last;   // [n-1] covered positions
    float *bounds;            // [2n-1][6]
    uint32_t *esize;          // [2n-1] number of LinearBVHNodes the subtree emits
    uint32_t *visits;         // [n-1] bottom-up arrival counters
};

__device__ __forceinline__ int delta(const uint32_t *code, uint32_t n, int i, int j) {
    if (j < 0 || j >= (int)n) return -1;
    const unsigned long long a = ((unsigned long long)code[i] << 32) | (uint32_t)i, b = ((unsigned long long)code[j] << 32) | (uint32_t)j;
    return __clzll((long long)(a ^ b));
}

__global__ void k_karras(Tree t, uint32_t *root_of_bin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = (int)t.n;
    if (i >= n - 1) return;
    const int d = (delta(t.code, n, i, i + 1) - delta(t.code, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(t.code, n, i, i - d);
    int lmax = 2;
    while (delta(t.code, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int s = lmax >> 1; s >= 1; s >>= 1) if (delta(t.code, n, i, i + (l + s) * d) > dmin) l += s;
    const int j = i + l * d;
    const int dnode = delta(t.code, n, i, j);
    int s = 0, step = l;
    do { step = (step + 1) >> 1; if (delta(t.code, n, i, i + (s + step) * d) > dnode) s += step; } while (step > 1);
    const int gamma = i + s * d + min(d, 0);
    const int f = min(i, j), e = max(i, j);
    const uint32_t lref = (f == gamma) ? (kLeafTag | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t rref = (e == gamma + 1) ? (kLeafTag | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    t.left[i] = lref; t.right[i] = rref; t.first[i] = (uint32_t)f; t.last[i] = (uint32_t)e;
    t.parent[(lref & kLeafTag) ? (uint32_t)(n - 1) + (lref & ~kLeafTag) : lref] = (uint32_t)i;
    t.parent[(rref & kLeafTag) ? (uint32_t)(n - 1) + (rref & ~kLeafTag) : rref] = (uint32_t)i;
    if (i == 0) t.parent[0] = kNone;
    // treelet root: covers exactly one run of equal top-12-bit codes (bvh.rs:405-421)
    const uint32_t bf = t.code[f] >> (kCodeBits - kTreeletBits), be = t.code[e] >> (kCodeBits - kTreeletBits);
    if (bf == be && (f == 0 || (t.code[f - 1] >> (kCodeBits - kTreeletBits)) != bf) && (e == n - 1 || (t.code[e + 1] >> (kCodeBits - kTreeletBits)) != bf))
        root_of_bin[bf] = (uint32_t)i;
}

template <class T> __device__ __forceinline__ T ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // not served from a stale L1 line

// ---- 5. bottom-up: bounds and emitted sizes. The second thread to arrive at a node owns it.
__global__ void k_fit(Tree t, const PrimBound *pb, const uint32_t *index, uint32_t max_prims, uint32_t *root_of_bin) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, n = t.n;
    if (p >= n) return;
    const uint32_t leaf = n - 1 + p;
    {
        const PrimBound b = pb[index[p]];
        float *o = t.bounds + 6 * (size_t)leaf;
        o[0] = b.lo[0]; o[1] = b.lo[1]; o[2] = b.lo[2]; o[3] = b.hi[0]; o[4] = b.hi[1]; o[5] = b.hi[2];
        t.esize[leaf] = 1;
        const uint32_t bin = t.code[p] >> (kCodeBits - kTreeletBits);   // single-primitive treelet
        if ((p == 0 || (t.code[p - 1] >> (kCodeBits - kTreeletBits)) != bin) && (p == n - 1 || (t.code[p + 1] >> (kCodeBits - kTreeletBits)) != bin)) root_of_bin[bin] = kLeafTag | p;
    }
    if (n == 1) return;
    uint32_t cur = t.parent[leaf];
    while (cur != kNone) {
        __threadfence();
        if (atomicAdd(&t.visits[cur], 1u) == 0u) return;
        __threadfence();
        const uint32_t l = t.left[cur], r = t.right[cur];
        const uint32_t li = (l & kLeafTag) ? n - 1 + (l & ~kLeafTag) : l, ri = (r & kLeafTag) ? n - 1 + (r & ~kLeafTag) : r;
        const float *a = t.bounds + 6 * (size_t)li, *b = t.bounds + 6 * (size_t)ri;
        float *o = t.bounds + 6 * (size_t)cur;
        for (int k = 0; k < 3; ++k) { o[k] = fminf(ld_agent(a + k), ld_agent(b + k)); o[3 + k] = fmaxf(ld_agent(a + 3 + k), ld_agent(b + 3 + k)); }
        const uint32_t count = t.last[cur] - t.first[cur] + 1;
        t.esize[cur] = (count <= max_prims) ? 1u : 1u + ld_agent(t.esize + li) + ld_agent(t.esize + ri);
        cur = t.parent[cur];
    }
}

struct TreeletInfo { float bounds[6]; uint32_t esize, root; };
__global__ void k_treelet_info(Tree t, const uint32_t *root_of_bin, TreeletInfo *info) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (uint32_t)kBins) return;
    const uint32_t r = root_of_bin[b];
    info[b].root = r;
    if (r == kNone) { info[b].esize = 0; return; }
    const uint32_t id = (r & kLeafTag) ? t.n - 1 + (r & ~kLeafTag) : r;
    for (int k = 0; k < 6; ++k) info[b].bounds[k] = t.bounds[6 * (size_t)id + k];
    info[b].esize = t.esize[id];
}

// ---- 6. emit LinearBVHNodes: every radix-tree node that survives the leaf collapse finds its depth-first index by
//         walking up to its treelet root (flatten_bvhtree order, bvh.rs:662-693: node, first subtree, second subtree)
__global__ void k_emit(Tree t, uint32_t max_prims, const uint32_t *base_of_bin, PtBVHNode *out) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x, n = t.n;
    if (id >= 2 * n - 1) return;
    const bool is_leaf = id >= n - 1;
    const uint32_t f = is_leaf ? id - (n - 1) : t.first[id], e = is_leaf ? f : t.last[id];
    const uint32_t bin = t.code[f] >> (kCodeBits - kTreeletBits);
    if ((t.code[e] >> (kCodeBits - kTreeletBits)) != bin) return;          // upper-level node: replaced by the SAH levels
    uint32_t acc = 0, cur = id;
    for (;;) {
        const uint32_t p = t.parent[cur];
        const bool troot = p == kNone || (t.code[t.first[p]] >> (kCodeBits - kTreeletBits)) != (t.code[t.last[p]] >> (kCodeBits - kTreeletBits));
        if (troot) break;
        if (cur == id && t.last[p] - t.first[p] + 1 <= max_prims) return;   // absorbed into a collapsed leaf
        const uint32_t l = t.left[p], li = (l & kLeafTag) ? n - 1 + (l & ~kLeafTag) : l;
        acc += 1u + ((li == cur) ? 0u : t.esize[li]);
        cur = p;
    }
    const uint32_t dfs = base_of_bin[bin] + acc, count = e - f + 1;
    PtBVHNode o;
    for (int k = 0; k < 3; ++k) { o.bmin[k] = t.bounds[6 * (size_t)id + k]; o.bmax[k] = t.bounds[6 * (size_t)id + 3 + k]; }
    o.pad = 0;
    if (count <= max_prims) { o.offset = f; o.n_prims = (uint16_t)count; o.axis = 0; }
    else {
        const uint32_t l = t.left[id], li = (l & kLeafTag) ? n - 1 + (l & ~kLeafTag) : l;
        const uint32_t r = t.right[id], split = (r & kLeafTag) ? (r & ~kLeafTag) : t.first[r];   // first position of the second child
        const uint32_t x = t.code[split - 1] ^ t.code[split];
        o.offset = dfs + 1u + t.esize[li]; o.n_prims = 0;
        o.axis = x ? (uint8_t)((31 - __clz((int)x)) % 3) : 0;   // emit_lbvh: axis = bit_index % 3 (bvh.rs:571)
    }
    out[dfs] = o;
}

// ---- SAH over the treelet roots (build_upper_sah, bvh.rs:578-658), host side: at most 4096 items
struct Upper {
    const std::vector<TreeletInfo> &tr;      // non-empty treelets in Morton order
    std::vector<uint32_t> base;              // depth-first index of every treelet's root
    std::vector<std::pair<uint32_t, PtBVHNode>> nodes;   // the upper interior nodes with their indices
    uint32_t next = 0;
    explicit Upper(const std::vector<TreeletInfo> &t) : tr(t), base(t.size(), 0) {}
    static float area(const float b[6]) { const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2]; return (dx * dy + dx * dz + dy * dz) * 2.0f; }
    static void grow(float b[6], const float o[6]) { for (int k = 0; k < 3; ++k) { b[k] = std::fmin(b[k], o[k]); b[3 + k] = std::fmax(b[3 + k], o[3 + k]); } }
    static void empty(float b[6]) { for (int k = 0; k < 3; ++k) { b[k] = std::numeric_limits<float>::max(); b[3 + k] = std::numeric_limits<float>::lowest(); } }
    void build(std::vector<uint32_t> &items, size_t start, size_t end, float out_bounds[6]) {
        if (end - start == 1) {
            const uint32_t t = items[start];
            base[t] = next; next += tr[t].esize;
            std::memcpy(out_bounds, tr[t].bounds, 24);
            return;
        }
        const uint32_t me = next++;
        const size_t slot = nodes.size();
        nodes.emplace_back(me, PtBVHNode{});
        float bounds[6], cb[6]; empty(bounds); empty(cb);
        auto centroid = [&](uint32_t t, int k) { return (tr[t].bounds[k] + tr[t].bounds[3 + k]) * 0.5f; };
        for (size_t i = start; i < end; ++i) {
            grow(bounds, tr[items[i]].bounds);
            for (int k = 0; k < 3; ++k) { const float c = centroid(items[i], k); cb[k] = std::fmin(cb[k], c); cb[3 + k] = std::fmax(cb[3 + k], c); }
        }
        const float ex = cb[3] - cb[0], ey = cb[4] - cb[1], ez = cb[5] - cb[2];
        const int dim = (ex > ey && ex > ez) ? 0 : ((ey > ez) ? 1 : 2);
        size_t mid = (start + end) / 2;
        if (cb[3 + dim] != cb[dim]) {
            constexpr int NB = 12;
            auto bucket = [&](uint32_t t) {
                const float v = NB * ((centroid(t, dim) - cb[dim]) / (cb[3 + dim] - cb[dim]));
                int b = (v > 0.0f) ? (v >= (float)NB ? NB : (int)v) : 0;
                return b == NB ? NB - 1 : b;
            };
            uint32_t cnt[NB] = {}; float bb[NB][6];
            for (int b = 0; b < NB; ++b) empty(bb[b]);
            for (size_t i = start; i < end; ++i) { const int b = bucket(items[i]); cnt[b]++; grow(bb[b], tr[items[i]].bounds); }
            float best = 0.0f; int best_b = -1;
            for (int i = 0; i < NB - 1; ++i) {
                float b0[6], b1[6]; empty(b0); empty(b1); uint32_t c0 = 0, c1 = 0;
                for (int j = 0; j <= i; ++j) { if (cnt[j]) grow(b0, bb[j]); c0 += cnt[j]; }
                for (int j = i + 1; j < NB; ++j) { if (cnt[j]) grow(b1, bb[j]); c1 += cnt[j]; }
                const float cost = 0.125f + ((c0 ? (float)c0 * area(b0) : 0.0f) + (c1 ? (float)c1 * area(b1) : 0.0f)) / area(bounds);
                if (best_b < 0 || cost < best) { best = cost; best_b = i; }
            }
            auto it = std::stable_partition(items.begin() + start, items.begin() + end, [&](uint32_t t) { return bucket(t) <= best_b; });
            const size_t m = (size_t)(it - items.begin());
            if (m != start && m != end) mid = m;
        }
        float lb[6], rb[6];
        build(items, start, mid, lb);
        const uint32_t second = next;
        build(items, mid, end, rb);
        PtBVHNode &nd = nodes[slot].second;
        for (int k = 0; k < 3; ++k) { nd.bmin[k] = bounds[k]; nd.bmax[k] = bounds[3 + k]; }
        nd.offset = second; nd.n_prims = 0; nd.axis = (uint8_t)dim; nd.pad = 0;
        std::memcpy(out_bounds, bounds, 24);
    }
};

struct DevBuf {   // frees its allocations on scope exit
    std::vector<void *> p;
    template <class T> bool alloc(T **out, size_t count) {
        void *q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return false;
        p.push_back(q); *out = (T *)q; return true;
    }
    ~DevBuf() { for (void *q : p) hipFree(q); }
};
}  // namespace

int build_hlbvh_gpu(const std::vector<PrimBound> &prims, uint32_t max_node_prims, std::vector<PtBVHNode> &nodes, std::vector<uint32_t> &ordered, const char **err) {
    static const char *e_hip = "HLBVH build: HIP call failed", *e_mem = "HLBVH build: out of device memory";
    nodes.clear(); ordered.clear();
    const uint32_t n = (uint32_t)prims.size();
    if (n == 0) return 0;
    if (max_node_prims == 0 || max_node_prims > 255) max_node_prims = std::min<uint32_t>(255, std::max<uint32_t>(1, max_node_prims));
    const bool timing = std::getenv("PT_BVH_TIMING") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    DevBuf dev;
    PrimBound *d_pb; uint32_t *d_cb, *d_code[2], *d_idx[2], *d_hist, *d_root_of_bin, *d_base_of_bin; float *d_cbf; TreeletInfo *d_info;
    Tree t{}; t.n = n;
    const uint32_t n_tiles = (n + kSortTile - 1) / kSortTile, n_int = n > 1 ? n - 1 : 0;
    bool ok = dev.alloc(&d_pb, n) && dev.alloc(&d_cb, 6) && dev.alloc(&d_cbf, 6) && dev.alloc(&d_code[0], n) && dev.alloc(&d_code[1], n) && dev.alloc(&d_idx[0], n) &&
              dev.alloc(&d_idx[1], n) && dev.alloc(&d_hist, 256 * (size_t)n_tiles) && dev.alloc(&d_root_of_bin, kBins) && dev.alloc(&d_base_of_bin, kBins) && dev.alloc(&d_info, kBins) &&
              dev.alloc(&t.left, n_int) && dev.alloc(&t.right, n_int) && dev.alloc(&t.parent, 2 * (size_t)n) && dev.alloc(&t.first, n_int) && dev.alloc(&t.last, n_int) &&
              dev.alloc(&t.bounds, 6 * 2 * (size_t)n) && dev.alloc(&t.esize, 2 * (size_t)n) && dev.alloc(&t.visits, n_int);
    if (!ok) { if (err) *err = e_mem; return 1; }
#define HCHK(x) do { if ((x) != hipSuccess) { if (err) *err = e_hip; return 2; } } while (0)
    HCHK(hipMemcpy(d_pb, prims.data(), (size_t)n * sizeof(PrimBound), hipMemcpyHostToDevice));
    const uint32_t cb_init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    HCHK(hipMemcpy(d_cb, cb_init, sizeof cb_init, hipMemcpyHostToDevice));
    HCHK(hipMemset(d_root_of_bin, 0xFF, kBins * 4));
    HCHK(hipMemset(t.visits, 0, std::max<size_t>(n_int, 1) * 4));
    HCHK(hipMemset(t.parent, 0xFF, 4));   // the root has no parent: k_karras writes it too, but does not run for a single primitive (k_emit reads it)
    hipLaunchKernelGGL(k_centroid_bounds, dim3(std::min<uint32_t>(1024, (n + 255) / 256)), dim3(256), 0, 0, d_pb, n, d_cb);
    uint32_t cb_ord[6]; float cbf[6];
    HCHK(hipMemcpy(cb_ord, d_cb, sizeof cb_ord, hipMemcpyDeviceToHost));
    for (int k = 0; k < 6; ++k) cbf[k] = ord2f(cb_ord[k]);
    HCHK(hipMemcpy(d_cbf, cbf, sizeof cbf, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_morton, dim3((n + 255) / 256), dim3(256), 0, 0, d_pb, n, d_cbf, d_code[0], d_idx[0]);
    int cur = 0;
    for (int shift = 0; shift < kCodeBits; shift += 8) {
        hipLaunchKernelGGL(k_sort_hist, dim3(n_tiles), dim3(64), 0, 0, d_code[cur], n, shift, n_tiles, d_hist);
        hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(1024), 0, 0, d_hist, 256u * n_tiles);
        hipLaunchKernelGGL(k_sort_scatter, dim3(n_tiles), dim3(64), 0, 0, d_code[cur], d_idx[cur], n, shift, n_tiles, d_hist, d_code[cur ^ 1], d_idx[cur ^ 1]);
        cur ^= 1;
    }
    t.code = d_code[cur];
    if (n > 1) hipLaunchKernelGGL(k_karras, dim3((n - 1 + 255) / 256), dim3(256), 0, 0, t, d_root_of_bin);
    hipLaunchKernelGGL(k_fit, dim3((n + 255) / 256), dim3(256), 0, 0, t, d_pb, d_idx[cur], max_node_prims, d_root_of_bin);
    hipLaunchKernelGGL(k_treelet_info, dim3(kBins / 256), dim3(256), 0, 0, t, d_root_of_bin, d_info);
    std::vector<TreeletInfo> info(kBins), tr;
    HCHK(hipMemcpy(info.data(), d_info, kBins * sizeof(TreeletInfo), hipMemcpyDeviceToHost));
    std::vector<uint32_t> bin_of;
    for (uint32_t b = 0; b < (uint32_t)kBins; ++b) if (info[b].root != kNone) { tr.push_back(info[b]); bin_of.push_back(b); }
    auto t1 = std::chrono::steady_clock::now();
    Upper up(tr);
    std::vector<uint32_t> items(tr.size());
    for (size_t i = 0; i < items.size(); ++i) items[i] = (uint32_t)i;
    float root_bounds[6];
    up.build(items, 0, items.size(), root_bounds);
    std::vector<uint32_t> base_of_bin(kBins, 0);
    for (size_t i = 0; i < tr.size(); ++i) base_of_bin[bin_of[i]] = up.base[i];
    const uint32_t total = up.next;
    auto t2 = std::chrono::steady_clock::now();
    PtBVHNode *d_out;
    if (!dev.alloc(&d_out, total)) { if (err) *err = e_mem; return 1; }
    HCHK(hipMemcpy(d_base_of_bin, base_of_bin.data(), kBins * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_emit, dim3((2 * n - 1 + 255) / 256), dim3(256), 0, 0, t, max_node_prims, d_base_of_bin, d_out);
    nodes.resize(total); ordered.resize(n);
    HCHK(hipMemcpy(nodes.data(), d_out, (size_t)total * sizeof(PtBVHNode), hipMemcpyDeviceToHost));
    HCHK(hipMemcpy(ordered.data(), d_idx[cur], (size_t)n * 4, hipMemcpyDeviceToHost));
    HCHK(hipGetLastError());
#undef HCHK
    for (auto &un : up.nodes) nodes[un.first] = un.second;
    if (timing) {
        auto t3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[hlbvh] %u prims -> %u nodes, %zu treelets: device sort+tree %.1f ms, upper SAH %.1f ms, emit+readback %.1f ms\n", n, total, tr.size(), ms(t0, t1), ms(t1, t2), ms(t2, t3));
    }
    return 0;
}

}  // namespace pth
