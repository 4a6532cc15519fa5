// host_bvh.cpp -- host-side SAH BVH construction (used when the caller does not pass a prebuilt
// accelerator). The Rust host normally owns this step (BASELINE north_star); the tree must be the SAME tree
// the reference builds, because equal-t ties are resolved by traversal order (SURVEY "Hard parts"):
//   BVHAccel::new / recursive_build / split_sah / flatten_bvhtree   accelerators/bvh.rs:145-375,662-693
//   Bounds3f default/union/offset/surface_area/maximum_extent        core/geometry/bounds.rs:336-410,459-515
// Implementation: explicit work stack (no recursion), SoA primitive records. Emission order of
// `ordered_prims` follows the reference (right subtree before left, bvh.rs:275-276); node numbering is the
// depth-first left-first order of flatten_bvhtree.
#include "host_bvh.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace pth {

namespace {
struct B3 {
    float lo[3], hi[3];
    B3() { for (int i = 0; i < 3; ++i) { lo[i] = std::numeric_limits<float>::max(); hi[i] = std::numeric_limits<float>::lowest(); } }
    void grow(const B3 &o) { for (int i = 0; i < 3; ++i) { lo[i] = std::fmin(lo[i], o.lo[i]); hi[i] = std::fmax(hi[i], o.hi[i]); } }
    void grow_pt(const float p[3]) { for (int i = 0; i < 3; ++i) { lo[i] = std::fmin(lo[i], p[i]); hi[i] = std::fmax(hi[i], p[i]); } }
    float area() const { float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return (dx * dy + dx * dz + dy * dz) * 2.0f; }
    int max_extent() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx > dy && dx > dz) return 0;
        return (dy > dz) ? 1 : 2;
    }
    float offset(const float p[3], int d) const {  // Bounds3::offset, one component
        float o = p[d] - lo[d];
        if (hi[d] > lo[d]) o /= hi[d] - lo[d];
        return o;
    }
};
struct Rec { uint32_t prim; B3 b; float c[3]; };
struct TNode { B3 b; int32_t left = -1, right = -1; uint32_t first = 0, count = 0; uint8_t axis = 0; };

inline size_t bucket_of(const B3 &cb, const Rec &r, int dim) {
    float v = 12.0f * cb.offset(r.c, dim);
    size_t b = (v > 0.0f) ? (v >= 1.8e19f ? SIZE_MAX : (size_t)v) : 0;  // Rust `as usize` saturates, NaN -> 0
    return b == 12 ? 11 : b;
}
}  // namespace

namespace {
// Builds the subtree over recs[start, end) into `tree` (node 0 = its root; children always after their parent).
// `obase` = position in `ordered` where this subtree's primitives go: the reference processes the right subtree
// before the left one (bvh.rs:275-276), so a node's right child inherits its base and the left child starts after
// the right subtree's primitives. That makes every subtree independent of the rest -> subtrees are built in parallel.
// Top-level pass: subtrees that have shrunk to `defer_max` primitives or fewer (but are still worth a task) are not
// expanded but returned in `deferred`.
struct Work { int32_t node; size_t start, end, obase; };
// Chunked parallel reduction for the O(np) scans of the big top-level nodes (bounds / centroid bounds / bucket statistics
// are unions and counts: the result does not depend on the order). `threads` <= 1 runs inline.
template <class Acc, class Body, class Merge>
void scan_range(size_t start, size_t end, unsigned threads, Acc &acc, Body body, Merge merge) {
    const size_t np = end - start;
    if (threads <= 1 || np < (size_t)262144) { for (size_t i = start; i < end; ++i) body(acc, i); return; }
    std::vector<Acc> part(threads);
    std::vector<std::thread> pool;
    const size_t chunk = (np + threads - 1) / threads;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&, t]() { const size_t a = start + t * chunk, b = std::min(end, a + chunk); for (size_t i = a; i < b; ++i) body(part[t], i); });
    for (auto &th : pool) th.join();
    for (unsigned t = 0; t < threads; ++t) merge(acc, part[t]);
}
struct BucketAcc { size_t cnt[12] = {0}; B3 bb[12]; };
void build_range(std::vector<Rec> &recs, size_t max_prims, std::vector<uint32_t> &ordered, std::vector<TNode> &tree, Work root,
                 size_t defer_max, std::vector<Work> *deferred, unsigned scan_threads = 1) {
    std::vector<Work> work;
    work.push_back(root);
    while (!work.empty()) {
        Work w = work.back(); work.pop_back();
        const size_t start = w.start, end = w.end, np = end - start;
        if (deferred && np <= defer_max && np >= 4096) { deferred->push_back(w); continue; }
        B3 bounds;
        scan_range(start, end, scan_threads, bounds, [&](B3 &a, size_t i) { a.grow(recs[i].b); }, [](B3 &a, const B3 &b) { a.grow(b); });
        auto leaf = [&]() {
            TNode &t = tree[w.node];
            t.first = (uint32_t)w.obase; t.count = (uint32_t)np; t.b = bounds;
            for (size_t i = start; i < end; ++i) ordered[w.obase + (i - start)] = recs[i].prim;
        };
        if (np == 1) { leaf(); continue; }
        B3 cb;
        scan_range(start, end, scan_threads, cb, [&](B3 &a, size_t i) { a.grow_pt(recs[i].c); }, [](B3 &a, const B3 &b) { a.grow(b); });
        const int dim = cb.max_extent();
        if (cb.hi[dim] == cb.lo[dim]) { leaf(); continue; }
        size_t mid = (start + end) / 2;
        if (np <= 2) {  // bvh.rs:304-311
            if (start != end - 1 && recs[end - 1].c[dim] < recs[start].c[dim]) std::swap(recs[start], recs[end - 1]);
        } else {
            BucketAcc ba;
            scan_range(start, end, scan_threads, ba, [&](BucketAcc &a, size_t i) { size_t b = bucket_of(cb, recs[i], dim); a.cnt[b]++; a.bb[b].grow(recs[i].b); },
                       [](BucketAcc &a, const BucketAcc &b) { for (int k = 0; k < 12; ++k) { a.cnt[k] += b.cnt[k]; a.bb[k].grow(b.bb[k]); } });
            size_t (&cnt)[12] = ba.cnt; B3 (&bb)[12] = ba.bb;
            float cost[11];
            const float inv_total = bounds.area();
            for (int i = 0; i < 11; ++i) {
                B3 b0, b1; size_t c0 = 0, c1 = 0;
                for (int j = 0; j <= i; ++j) { b0.grow(bb[j]); c0 += cnt[j]; }
                for (int j = i + 1; j < 12; ++j) { b1.grow(bb[j]); c1 += cnt[j]; }
                cost[i] = 1.0f + ((float)c0 * b0.area() + (float)c1 * b1.area()) / inv_total;
            }
            float best = cost[0]; int split = 0;
            for (int i = 1; i < 11; ++i) if (cost[i] < best) { best = cost[i]; split = i; }
            if (np > max_prims || best < (float)np) {
                // Iterator::partition_in_place: swap the first `false` from the front with the last `true` from the back
                size_t i = start, j = end, trues = 0;
                for (;;) {
                    while (i < j && (int)bucket_of(cb, recs[i], dim) <= split) { ++i; ++trues; }
                    if (i >= j) break;
                    size_t head = i++;
                    while (j > i && !((int)bucket_of(cb, recs[j - 1], dim) <= split)) --j;
                    if (j <= i) break;
                    size_t tail = --j;
                    std::swap(recs[head], recs[tail]);
                    ++trues;
                }
                mid = start + trues;
            } else { leaf(); continue; }
        }
        int32_t l = (int32_t)tree.size(); tree.emplace_back();
        int32_t r = (int32_t)tree.size(); tree.emplace_back();
        tree[w.node].left = l; tree[w.node].right = r; tree[w.node].axis = (uint8_t)dim; tree[w.node].count = 0;
        work.push_back({l, start, mid, w.obase + (end - mid)});
        work.push_back({r, mid, end, w.obase});
    }
}
}  // namespace

void build_sah_bvh(const std::vector<PrimBound> &prims, uint32_t max_node_prims, std::vector<PtBVHNode> &nodes, std::vector<uint32_t> &ordered) {
    nodes.clear(); ordered.clear();
    const size_t n = prims.size();
    if (n == 0) return;
    const bool timing = std::getenv("PT_BVH_TIMING") != nullptr;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto t_start = tnow();
    auto lap = [&](const char *what) { if (timing) { auto t = tnow(); std::fprintf(stderr, "[bvh] %-18s %.3f s\n", what, std::chrono::duration<double>(t - t_start).count()); t_start = t; } };
    const size_t max_prims = std::min<uint32_t>(255u, max_node_prims);
    std::vector<Rec> recs(n);
    for (size_t i = 0; i < n; ++i) {
        Rec &r = recs[i];
        r.prim = (uint32_t)i;
        for (int k = 0; k < 3; ++k) { r.b.lo[k] = prims[i].lo[k]; r.b.hi[k] = prims[i].hi[k]; r.c[k] = prims[i].lo[k] * 0.5f + prims[i].hi[k] * 0.5f; }
    }
    lap("records");
    ordered.assign(n, 0u);
    std::vector<TNode> tree;
    tree.reserve(2 * n);
    tree.emplace_back();
    // the top six levels or so on this thread (their big scans chunked over threads); subtrees of <= n/64 primitives go to
    // worker threads
    unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 1; if (hw > 64) hw = 64;
    if (const char *e = std::getenv("PT_BVH_THREADS")) { int v = std::atoi(e); if (v >= 1 && v <= 256) hw = (unsigned)v; }
    std::vector<Work> deferred;
    const bool parallel = n >= 65536 && hw > 1;
    build_range(recs, max_prims, ordered, tree, Work{0, 0, n, 0}, n / 64, parallel ? &deferred : nullptr, parallel ? std::min(hw, 16u) : 1u);
    lap("top levels");
    if (!deferred.empty()) {
        // largest subtrees first
        std::vector<std::vector<TNode>> local(deferred.size());
        std::atomic<size_t> next{0};
        std::vector<size_t> order(deferred.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return deferred[a].end - deferred[a].start > deferred[b].end - deferred[b].start; });
        auto worker = [&]() {
            for (;;) {
                size_t k = next.fetch_add(1); if (k >= order.size()) break;
                const size_t t = order[k];
                local[t].reserve(2 * (deferred[t].end - deferred[t].start));
                local[t].emplace_back();
                Work w = deferred[t]; w.node = 0;
                build_range(recs, max_prims, ordered, local[t], w, 0, nullptr);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned i = 0; i + 1 < std::min<size_t>(hw, order.size()); ++i) pool.emplace_back(worker);
        worker();
        for (auto &th : pool) th.join();
        lap("subtrees");
        // splice the local trees behind the top tree (children stay after their parents)
        for (size_t t = 0; t < deferred.size(); ++t) {
            const int32_t off = (int32_t)tree.size() - 1;   // local node k (k >= 1) -> global off + k ; local root -> deferred[t].node
            const std::vector<TNode> &lt = local[t];
            auto fix = [&](int32_t c) { return c < 0 ? c : (c == 0 ? deferred[t].node : off + c); };
            TNode root = lt[0]; root.left = fix(root.left); root.right = fix(root.right);
            tree[deferred[t].node] = root;
            for (size_t k = 1; k < lt.size(); ++k) { TNode x = lt[k]; x.left = fix(x.left); x.right = fix(x.right); tree.push_back(x); }
        }
    }
    lap("splice");
    // interior bounds = union(left, right) (init_interior, bvh.rs:115-121): children are always created after
    // their parent, so a reverse sweep sees both children finished.
    for (size_t i = tree.size(); i-- > 0;) {
        TNode &t = tree[i];
        if (t.left >= 0) { B3 b = tree[t.left].b; b.grow(tree[t.right].b); t.b = b; }
    }
    lap("interior bounds");
    // flatten_bvhtree: depth-first, left child first; interior.offset = index of the right child
    nodes.resize(tree.size());
    struct F { int32_t node; int32_t parent_slot; };
    std::vector<F> st;
    st.push_back({0, -1});
    uint32_t next = 0;
    while (!st.empty()) {
        F f = st.back(); st.pop_back();
        const TNode &t = tree[f.node];
        uint32_t my = next++;
        if (f.parent_slot >= 0) nodes[f.parent_slot].offset = my;
        PtBVHNode ln; std::memset(&ln, 0, sizeof ln);
        for (int k = 0; k < 3; ++k) { ln.bmin[k] = t.b.lo[k]; ln.bmax[k] = t.b.hi[k]; }
        if (t.left < 0) { ln.n_prims = (uint16_t)t.count; ln.offset = t.first; ln.axis = 0; nodes[my] = ln; }
        else {
            ln.n_prims = 0; ln.axis = t.axis; nodes[my] = ln;
            st.push_back({t.right, (int32_t)my});  // visited after the whole left subtree
            st.push_back({t.left, -1});
        }
    }
    lap("flatten");
}

}  // namespace pth
