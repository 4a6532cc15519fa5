// ref_hlbvh.h -- ORACLE (test infrastructure, never shipped or linked by the product): CPU restatement of the
// HLBVH construction the library runs on the GPU for splitmethod "hlbvh" (pbrt-rust_amd/csrc/gpu_bvh.hip).
//
// Follows the algorithm of BVHAccel::hlbvh_build / emit_lbvh / build_upper_sah (accelerators/bvh.rs:377-660),
// encode_morton3 / left_shift3 (:832-857) and radix_sort (:859-912, here std::stable_sort: same permutation).
// PARITY UNPINNED against the reference for this path, and it would stay so even with a Rust toolchain: the
// reference's emit_lbvh leaf loop and its channel-ordered primitive list are broken / timing dependent
// (bvh.rs:424-455,488-493). The pinned statement is therefore: (1) GPU tree == this tree, bit for bit;
// (2) images traced through it equal the SAH tree's images. The deliberate choices (Morton-ordered primitives, leaf
// when count <= max_node_prims, equal codes split by position, stable upper partition with a middle fallback) are
// listed in gpu_bvh.hip's header and DESIGN.md.
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>
#include "ref_scene.h"

namespace ref {
namespace hlbvh {

inline uint32_t left_shift3(uint32_t x) {  // bvh.rs:832-848
    if (x == (1u << 10)) x -= 1;
    x = (x | (x << 16)) & 0x30000ffu;
    x = (x | (x << 8)) & 0x300f00fu;
    x = (x | (x << 4)) & 0x30c30c3u;
    x = (x | (x << 2)) & 0x9249249u;
    return x;
}
inline uint32_t quantize(float v) { return (v > 0.0f) ? (v >= 1024.0f ? 1024u : (uint32_t)v) : 0u; }  // Rust `as u32`

struct BuildNode { float b[6]; int child[2] = {-1, -1}; uint32_t first = 0, count = 0; uint8_t axis = 0; };

struct Builder {
    uint32_t max_prims;
    std::vector<uint32_t> code;       // sorted Morton codes
    std::vector<const Bounds3 *> pb;  // primitive bounds in sorted order
    std::vector<BuildNode> arena;

    static void empty(float b[6]) { for (int k = 0; k < 3; ++k) { b[k] = std::numeric_limits<float>::max(); b[3 + k] = std::numeric_limits<float>::lowest(); } }
    static void grow(float b[6], const float o[6]) { for (int k = 0; k < 3; ++k) { b[k] = fmin_(b[k], o[k]); b[3 + k] = fmax_(b[3 + k], o[3 + k]); } }
    static float area(const float b[6]) { float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2]; return (dx * dy + dx * dz + dy * dz) * 2.0f; }

    // emit_lbvh (bvh.rs:460-576) over sorted positions [s, e): split at the highest bit in which the first and the last
    // key differ; the key is the code followed by the position, so equal codes split by position.
    int emit(uint32_t s, uint32_t e) {
        int me = (int)arena.size();
        arena.emplace_back();
        if (e - s <= max_prims) {
            BuildNode n; empty(n.b);
            for (uint32_t i = s; i < e; ++i) { const float o[6] = {pb[i]->pmin.x, pb[i]->pmin.y, pb[i]->pmin.z, pb[i]->pmax.x, pb[i]->pmax.y, pb[i]->pmax.z}; grow(n.b, o); }
            n.first = s; n.count = e - s;
            arena[me] = n;
            return me;
        }
        auto key = [&](uint32_t i) { return ((uint64_t)code[i] << 32) | i; };
        const uint64_t x = key(s) ^ key(e - 1);
        const int bit = 63 - __builtin_clzll(x);
        uint32_t lo = s, hi = e - 1;   // invariant: bit clear at lo, set at hi
        while (lo + 1 != hi) { uint32_t mid = (lo + hi) / 2; if ((key(mid) >> bit) & 1) hi = mid; else lo = mid; }
        const int l = emit(s, hi), r = emit(hi, e);
        BuildNode n;
        std::memcpy(n.b, arena[l].b, 24); grow(n.b, arena[r].b);
        n.child[0] = l; n.child[1] = r; n.axis = bit >= 32 ? (uint8_t)((bit - 32) % 3) : 0;   // bvh.rs:571
        arena[me] = n;
        return me;
    }

    // build_upper_sah (bvh.rs:578-658) over treelet roots
    int upper(std::vector<int> &roots, size_t start, size_t end) {
        if (end - start == 1) return roots[start];
        int me = (int)arena.size();
        arena.emplace_back();
        float bounds[6], cb[6]; empty(bounds); empty(cb);
        auto centroid = [&](int r, int k) { return (arena[r].b[k] + arena[r].b[3 + k]) * 0.5f; };
        for (size_t i = start; i < end; ++i) {
            grow(bounds, arena[roots[i]].b);
            for (int k = 0; k < 3; ++k) { float c = centroid(roots[i], k); cb[k] = fmin_(cb[k], c); cb[3 + k] = fmax_(cb[3 + k], c); }
        }
        const float ex = cb[3] - cb[0], ey = cb[4] - cb[1], ez = cb[5] - cb[2];
        const int dim = (ex > ey && ex > ez) ? 0 : ((ey > ez) ? 1 : 2);
        size_t mid = (start + end) / 2;
        if (cb[3 + dim] != cb[dim]) {
            const int NB = 12;
            auto bucket = [&](int r) {
                float v = NB * ((centroid(r, dim) - cb[dim]) / (cb[3 + dim] - cb[dim]));
                int b = (v > 0.0f) ? (v >= (float)NB ? NB : (int)v) : 0;
                return b == NB ? NB - 1 : b;
            };
            uint32_t cnt[12] = {}; float bb[12][6];
            for (int b = 0; b < NB; ++b) empty(bb[b]);
            for (size_t i = start; i < end; ++i) { int b = bucket(roots[i]); cnt[b]++; grow(bb[b], arena[roots[i]].b); }
            float best = 0.0f; int best_b = -1;
            for (int i = 0; i < NB - 1; ++i) {
                float b0[6], b1[6]; empty(b0); empty(b1); uint32_t c0 = 0, c1 = 0;
                for (int j = 0; j <= i; ++j) { if (cnt[j]) grow(b0, bb[j]); c0 += cnt[j]; }
                for (int j = i + 1; j < NB; ++j) { if (cnt[j]) grow(b1, bb[j]); c1 += cnt[j]; }
                float cost = 0.125f + ((c0 ? (float)c0 * area(b0) : 0.0f) + (c1 ? (float)c1 * area(b1) : 0.0f)) / area(bounds);
                if (best_b < 0 || cost < best) { best = cost; best_b = i; }
            }
            auto it = std::stable_partition(roots.begin() + start, roots.begin() + end, [&](int r) { return bucket(r) <= best_b; });
            size_t m = (size_t)(it - roots.begin());
            if (m != start && m != end) mid = m;
        }
        const int l = upper(roots, start, mid), r = upper(roots, mid, end);
        BuildNode n; std::memcpy(n.b, bounds, 24);
        n.child[0] = l; n.child[1] = r; n.axis = (uint8_t)dim;
        arena[me] = n;
        return me;
    }

    uint32_t flatten(std::vector<PtBVHNode> &out, int node) {  // flatten_bvhtree (bvh.rs:662-693)
        const BuildNode &n = arena[node];
        uint32_t my = (uint32_t)out.size();
        out.emplace_back();
        PtBVHNode o;
        for (int k = 0; k < 3; ++k) { o.bmin[k] = n.b[k]; o.bmax[k] = n.b[3 + k]; }
        o.pad = 0;
        if (n.child[0] < 0) { o.offset = n.first; o.n_prims = (uint16_t)n.count; o.axis = 0; out[my] = o; return my; }
        flatten(out, n.child[0]);
        o.offset = flatten(out, n.child[1]); o.n_prims = 0; o.axis = n.axis;
        out[my] = o;
        return my;
    }
};

// item i keeps number i: fills nodes + ordered item numbers (same contract as build_accel)
inline void build(const std::vector<Bounds3> &bounds, uint32_t max_node_prims, std::vector<PtBVHNode> &nodes, std::vector<uint32_t> &ordered) {
    nodes.clear(); ordered.clear();
    const size_t n = bounds.size();
    if (n == 0) return;
    float cb[6]; Builder::empty(cb);
    std::vector<V3> cen(n);
    for (size_t i = 0; i < n; ++i) {
        cen[i] = bounds[i].pmin * 0.5f + bounds[i].pmax * 0.5f;
        cb[0] = fmin_(cb[0], cen[i].x); cb[1] = fmin_(cb[1], cen[i].y); cb[2] = fmin_(cb[2], cen[i].z);
        cb[3] = fmax_(cb[3], cen[i].x); cb[4] = fmax_(cb[4], cen[i].y); cb[5] = fmax_(cb[5], cen[i].z);
    }
    std::vector<uint32_t> code(n);
    for (size_t i = 0; i < n; ++i) {
        const float c[3] = {cen[i].x, cen[i].y, cen[i].z};
        uint32_t q[3];
        for (int k = 0; k < 3; ++k) { float o = c[k] - cb[k]; if (cb[3 + k] > cb[k]) o /= cb[3 + k] - cb[k]; q[k] = quantize(o * 1024.0f); }
        code[i] = (left_shift3(q[2]) << 2) | (left_shift3(q[1]) << 1) | left_shift3(q[0]);
    }
    ordered.resize(n);
    for (size_t i = 0; i < n; ++i) ordered[i] = (uint32_t)i;
    std::stable_sort(ordered.begin(), ordered.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });
    Builder B; B.max_prims = std::min<uint32_t>(255, std::max<uint32_t>(1, max_node_prims));
    B.code.resize(n); B.pb.resize(n);
    for (size_t i = 0; i < n; ++i) { B.code[i] = code[ordered[i]]; B.pb[i] = &bounds[ordered[i]]; }
    B.arena.reserve(2 * n + 8192);
    std::vector<int> roots;
    for (uint32_t s = 0, e = 1; e <= n; ++e)
        if (e == n || (B.code[s] & 0x3ffc0000u) != (B.code[e] & 0x3ffc0000u)) { roots.push_back(B.emit(s, e)); s = e; }   // bvh.rs:405-421
    const int root = B.upper(roots, 0, roots.size());
    nodes.reserve(B.arena.size());
    B.flatten(nodes, root);
}

}  // namespace hlbvh
}  // namespace ref
