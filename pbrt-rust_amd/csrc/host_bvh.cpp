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
#include <cmath>
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

void build_sah_bvh(const std::vector<PrimBound> &prims, uint32_t max_node_prims, std::vector<PtBVHNode> &nodes, std::vector<uint32_t> &ordered) {
    nodes.clear(); ordered.clear();
    const size_t n = prims.size();
    if (n == 0) return;
    const size_t max_prims = std::min<uint32_t>(255u, max_node_prims);
    std::vector<Rec> recs(n);
    for (size_t i = 0; i < n; ++i) {
        Rec &r = recs[i];
        r.prim = (uint32_t)i;
        for (int k = 0; k < 3; ++k) { r.b.lo[k] = prims[i].lo[k]; r.b.hi[k] = prims[i].hi[k]; r.c[k] = prims[i].lo[k] * 0.5f + prims[i].hi[k] * 0.5f; }
    }
    std::vector<TNode> tree;
    tree.reserve(2 * n);
    ordered.reserve(n);
    struct Work { int32_t node; size_t start, end; };
    std::vector<Work> work;
    tree.emplace_back();
    work.push_back({0, 0, n});
    while (!work.empty()) {
        Work w = work.back(); work.pop_back();
        const size_t start = w.start, end = w.end, np = end - start;
        B3 bounds;
        for (size_t i = start; i < end; ++i) bounds.grow(recs[i].b);
        auto leaf = [&]() {
            TNode &t = tree[w.node];
            t.first = (uint32_t)ordered.size(); t.count = (uint32_t)np; t.b = bounds;
            for (size_t i = start; i < end; ++i) ordered.push_back(recs[i].prim);
        };
        if (np == 1) { leaf(); continue; }
        B3 cb;
        for (size_t i = start; i < end; ++i) cb.grow_pt(recs[i].c);
        const int dim = cb.max_extent();
        if (cb.hi[dim] == cb.lo[dim]) { leaf(); continue; }
        size_t mid = (start + end) / 2;
        if (np <= 2) {  // bvh.rs:304-311
            if (start != end - 1 && recs[end - 1].c[dim] < recs[start].c[dim]) std::swap(recs[start], recs[end - 1]);
        } else {
            size_t cnt[12] = {0}; B3 bb[12];
            for (size_t i = start; i < end; ++i) { size_t b = bucket_of(cb, recs[i], dim); cnt[b]++; bb[b].grow(recs[i].b); }
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
        work.push_back({l, start, mid});   // popped second
        work.push_back({r, mid, end});     // popped first: right subtree emits its primitives first
    }
    // interior bounds = union(left, right) (init_interior, bvh.rs:115-121): children are always created after
    // their parent, so a reverse sweep sees both children finished.
    for (size_t i = tree.size(); i-- > 0;) {
        TNode &t = tree[i];
        if (t.left >= 0) { B3 b = tree[t.left].b; b.grow(tree[t.right].b); t.b = b; }
    }
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
}

}  // namespace pth
