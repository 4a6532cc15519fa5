// tools/study/wide_study.cpp -- CPU study for VERDICT r5 item 2 (a traversal structure built for the machine): what would a SAH-collapsed wide BVH with multi-triangle
// leaves and distance-ordered children cost per ray against the production walk (the reference's binary SAH tree collapsed two levels at a time, reference order)?
// Counts only (records fetched, box tests, triangle tests, leaf visits, pushes): plain float arithmetic, no parity claims. Driven by tools/study/wide_study.py.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

struct Node { float bmin[3], bmax[3]; uint32_t offset; uint16_t n_prims; uint8_t axis, pad; };   // PtBVHNode (include/mi355pt.h)
struct Ray { float o[3], d[3], tmax; uint32_t any; };
struct Stats { double rays, records, box_tests, tri_tests, leaf_visits, pushes, max_stack, hits, near_ties; };

static const Node *N; static const uint32_t *ORD; static const float *P; static const uint32_t *IDX;

static inline bool slab(const float *bmin, const float *bmax, const Ray &r, const float *inv, float tmax, float &tmin_out) {
    float tmin = 0.0f, tmx = tmax;
    for (int a = 0; a < 3; ++a) {
        float t0 = (bmin[a] - r.o[a]) * inv[a], t1 = (bmax[a] - r.o[a]) * inv[a];
        if (inv[a] < 0.0f) std::swap(t0, t1);
        t1 *= 1.0000004f;
        if (t0 > tmin) tmin = t0;
        if (t1 < tmx) tmx = t1;
        if (tmin > tmx) return false;
    }
    tmin_out = tmin; return true;
}
static inline bool tri_hit(uint32_t prim, const Ray &r, float tmax, float &t) {   // Moeller-Trumbore (counts only)
    const float *p0 = P + 3 * IDX[3 * prim], *p1 = P + 3 * IDX[3 * prim + 1], *p2 = P + 3 * IDX[3 * prim + 2];
    float e1[3], e2[3], pv[3], tv[3], qv[3];
    for (int k = 0; k < 3; ++k) { e1[k] = p1[k] - p0[k]; e2[k] = p2[k] - p0[k]; }
    pv[0] = r.d[1] * e2[2] - r.d[2] * e2[1]; pv[1] = r.d[2] * e2[0] - r.d[0] * e2[2]; pv[2] = r.d[0] * e2[1] - r.d[1] * e2[0];
    const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
    if (det == 0.0f) return false;
    const float inv = 1.0f / det;
    for (int k = 0; k < 3; ++k) tv[k] = r.o[k] - p0[k];
    const float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
    if (u < 0.0f || u > 1.0f) return false;
    qv[0] = tv[1] * e1[2] - tv[2] * e1[1]; qv[1] = tv[2] * e1[0] - tv[0] * e1[2]; qv[2] = tv[0] * e1[1] - tv[1] * e1[0];
    const float v = (r.d[0] * qv[0] + r.d[1] * qv[1] + r.d[2] * qv[2]) * inv;
    if (v < 0.0f || u + v > 1.0f) return false;
    t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv;
    return t > 1e-6f * std::fabs(t) && t < tmax;
}

// ---- walker A: the production walk (two binary levels per record, reference order, one-primitive-per-packet leaves as the reference built them) ----
static void walk_ref4(const Ray &r, Stats &s) {
    float inv[3] = {1.0f / r.d[0], 1.0f / r.d[1], 1.0f / r.d[2]};
    const bool neg[3] = {inv[0] < 0, inv[1] < 0, inv[2] < 0};
    float tmax = r.tmax; bool hit = false;
    struct E { uint32_t node; float tmin; };
    E stack[128]; int sp = 0; int maxsp = 0;
    uint32_t cur = 0; bool have = true;
    if (N[0].n_prims) { have = true; }
    while (true) {
        if (have) {
            const Node &n = N[cur];
            if (n.n_prims) {
                s.leaf_visits++;
                for (uint32_t i = 0; i < n.n_prims; ++i) { s.tri_tests++; float t; if (tri_hit(ORD[n.offset + i], r, tmax, t)) { hit = true; if (r.any) { s.hits++; s.max_stack = std::max<double>(s.max_stack, maxsp); return; } tmax = t; } }
                have = false;
            } else {
                s.records++;
                uint32_t slots[4]; int ns = 0;
                const uint32_t L = cur + 1, R = n.offset;
                const uint32_t order[2] = {neg[n.axis] ? R : L, neg[n.axis] ? L : R};
                for (uint32_t x : order) {
                    if (N[x].n_prims) slots[ns++] = x;
                    else { const uint32_t a = x + 1, b = N[x].offset; if (neg[N[x].axis]) { slots[ns++] = b; slots[ns++] = a; } else { slots[ns++] = a; slots[ns++] = b; } }
                }
                float T[4]; bool ok[4];
                for (int k = 0; k < ns; ++k) { s.box_tests++; ok[k] = slab(N[slots[k]].bmin, N[slots[k]].bmax, r, inv, tmax, T[k]); }
                int first = -1;
                for (int k = 0; k < ns; ++k) if (ok[k]) { first = k; break; }
                if (first < 0) have = false;
                else {
                    for (int k = ns - 1; k > first; --k) if (ok[k]) { stack[sp++] = {slots[k], T[k]}; s.pushes++; }
                    maxsp = std::max(maxsp, sp);
                    cur = slots[first];
                }
            }
        }
        if (!have) {
            bool found = false;
            while (sp > 0) { const E e = stack[--sp]; if (e.tmin < tmax) { cur = e.node; found = true; break; } }
            if (!found) break;
            have = true;
        }
    }
    if (hit) s.hits++;
    s.max_stack = std::max<double>(s.max_stack, maxsp);
}

// ---- the wide tree: collapsed from the same binary tree ----
struct WNode { int n; float bmin[8][3], bmax[8][3]; int32_t child[8]; };   // child >= 0: wide node; < 0: ~leaf index
struct WLeaf { uint32_t first, count; };
static std::vector<WNode> WN; static std::vector<WLeaf> WL; static std::vector<uint32_t> WPRIMS;
static int g_width = 4, g_leaf_max = 4;
static float area(const Node &n) { const float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2]; return 2.0f * (dx * dy + dy * dz + dz * dx); }
static uint32_t count_prims(uint32_t i, std::vector<uint32_t> &cnt) { if (cnt[i]) return cnt[i]; return cnt[i] = N[i].n_prims ? N[i].n_prims : count_prims(i + 1, cnt) + count_prims(N[i].offset, cnt); }
static void gather(uint32_t i, std::vector<uint32_t> &out) { if (N[i].n_prims) { for (uint32_t k = 0; k < N[i].n_prims; ++k) out.push_back(ORD[N[i].offset + k]); } else { gather(i + 1, out); gather(N[i].offset, out); } }
// Should the subtree under binary node i become ONE leaf? SAH with traversal cost 1 (a wide node step) against `ci` per triangle test: a leaf of k triangles costs ci * k.
static float g_ci = 0.6f;
static std::vector<char> g_leafify;
static double subtree_cost(uint32_t i, std::vector<uint32_t> &cnt, std::vector<double> &memo) {   // expected cost below i given the ray hits i's box (binary estimate, good enough to pick leaves)
    if (memo[i] >= 0.0) return memo[i];
    const uint32_t k = count_prims(i, cnt);
    const double as_leaf = g_ci * k;
    if (N[i].n_prims) { g_leafify[i] = 1; return memo[i] = as_leaf; }
    const double a = area(N[i]);
    const double cl = subtree_cost(i + 1, cnt, memo), cr = subtree_cost(N[i].offset, cnt, memo);
    const double split = 0.5 + (a > 0 ? (area(N[i + 1]) / a) * cl + (area(N[N[i].offset]) / a) * cr : 0.0);   // 0.5: a binary level is about half a wide step
    g_leafify[i] = (k <= (uint32_t)g_leaf_max && as_leaf <= split) ? 1 : 0;
    return memo[i] = g_leafify[i] ? as_leaf : split;
}
static bool is_leaf(uint32_t i, std::vector<uint32_t> &cnt, std::vector<double> &memo) { subtree_cost(i, cnt, memo); return g_leafify[i] != 0; }
static int32_t build_wide(uint32_t i, std::vector<uint32_t> &cnt, std::vector<double> &memo) {
    if (is_leaf(i, cnt, memo)) {
        WLeaf l; l.first = (uint32_t)WPRIMS.size(); std::vector<uint32_t> pr; gather(i, pr); l.count = (uint32_t)pr.size();
        WPRIMS.insert(WPRIMS.end(), pr.begin(), pr.end()); WL.push_back(l); return ~(int32_t)(WL.size() - 1);
    }
    // open the child with the largest area until `width` children (standard wide-BVH collapse)
    std::vector<uint32_t> kids{i + 1, N[i].offset};
    while ((int)kids.size() < g_width) {
        int best = -1; float ba = -1.0f;
        for (size_t k = 0; k < kids.size(); ++k) if (!is_leaf(kids[k], cnt, memo)) { const float a = area(N[kids[k]]); if (a > ba) { ba = a; best = (int)k; } }
        if (best < 0) break;
        const uint32_t x = kids[best]; kids[best] = x + 1; kids.push_back(N[x].offset);
    }
    const int32_t me = (int32_t)WN.size(); WN.emplace_back();
    WNode w; w.n = (int)kids.size();
    for (int k = 0; k < w.n; ++k) { std::memcpy(w.bmin[k], N[kids[k]].bmin, 12); std::memcpy(w.bmax[k], N[kids[k]].bmax, 12); }
    for (int k = 0; k < w.n; ++k) w.child[k] = build_wide(kids[k], cnt, memo);
    WN[me] = w;
    return me;
}
static void walk_wide(const Ray &r, Stats &s) {
    float inv[3] = {1.0f / r.d[0], 1.0f / r.d[1], 1.0f / r.d[2]};
    float tmax = r.tmax; bool hit = false; float best2 = INFINITY;
    struct E { int32_t c; float tmin; };
    E stack[256]; int sp = 0, maxsp = 0;
    int32_t cur = 0; bool have = true;
    while (true) {
        if (have) {
            if (cur < 0) {
                const WLeaf &l = WL[~cur]; s.leaf_visits++;
                for (uint32_t i = 0; i < l.count; ++i) {
                    s.tri_tests++; float t;
                    if (tri_hit(WPRIMS[l.first + i], r, tmax * 1.00001f, t)) {
                        if (r.any) { s.hits++; s.max_stack = std::max<double>(s.max_stack, maxsp); return; }
                        if (t < tmax) { best2 = tmax; tmax = t; hit = true; } else best2 = std::min(best2, t);
                    }
                }
                have = false;
            } else {
                const WNode &w = WN[cur]; s.records++;
                E e[8]; int ne = 0;
                for (int k = 0; k < w.n; ++k) { s.box_tests++; float T; if (slab(w.bmin[k], w.bmax[k], r, inv, tmax, T)) e[ne++] = {w.child[k], T}; }
                if (ne == 0) have = false;
                else {
                    std::sort(e, e + ne, [](const E &a, const E &b) { return a.tmin < b.tmin; });   // nearest first
                    for (int k = ne - 1; k > 0; --k) { stack[sp++] = e[k]; s.pushes++; }
                    maxsp = std::max(maxsp, sp);
                    cur = e[0].c;
                }
            }
        }
        if (!have) {
            bool found = false;
            while (sp > 0) { const E e = stack[--sp]; if (e.tmin < tmax) { cur = e.c; found = true; break; } }
            if (!found) break;
            have = true;
        }
    }
    if (hit) { s.hits++; if (best2 <= tmax * 1.00001f) s.near_ties++; }
    s.max_stack = std::max<double>(s.max_stack, maxsp);
}

extern "C" {
// which: 0 = production walk over the reference tree; else the wide tree of (width, leaf_max, ci) built on first use
void study_set(const Node *nodes, const uint32_t *ordered, const float *p, const uint32_t *idx) { N = nodes; ORD = ordered; P = p; IDX = idx; }
int study_build_wide(uint32_t n_nodes, int width, int leaf_max, float ci, double *out) {
    g_width = width; g_leaf_max = leaf_max; g_ci = ci; WN.clear(); WL.clear(); WPRIMS.clear();
    std::vector<uint32_t> cnt(n_nodes, 0); std::vector<double> memo(n_nodes, -1.0); g_leafify.assign(n_nodes, 0);
    build_wide(0, cnt, memo);
    double fill = 0; for (auto &w : WN) fill += w.n;
    out[0] = (double)WN.size(); out[1] = (double)WL.size(); out[2] = fill / std::max<size_t>(1, WN.size()); out[3] = (double)WPRIMS.size() / std::max<size_t>(1, WL.size());
    return 0;
}
void study_walk(int which, const Ray *rays, uint32_t n, Stats *s) {
    std::memset(s, 0, sizeof *s); s->rays = n;
    for (uint32_t i = 0; i < n; ++i) { if (which == 0) walk_ref4(rays[i], *s); else walk_wide(rays[i], *s); }
}
// closest hit of the reference walk, for spawning secondary rays: returns t (inf = miss)
float study_closest(const Ray *r) {
    Stats s; std::memset(&s, 0, sizeof s);
    // reuse walker A but capture tmax: re-run cheaply
    float inv[3] = {1.0f / r->d[0], 1.0f / r->d[1], 1.0f / r->d[2]};
    float tmax = r->tmax; uint32_t stack[128]; int sp = 0; uint32_t cur = 0;
    for (;;) {
        const Node &n = N[cur]; float T;
        if (slab(n.bmin, n.bmax, *r, inv, tmax, T)) {
            if (n.n_prims) { for (uint32_t i = 0; i < n.n_prims; ++i) { float t; if (tri_hit(ORD[n.offset + i], *r, tmax, t)) tmax = t; } if (!sp) break; cur = stack[--sp]; }
            else { if (inv[n.axis] < 0) { stack[sp++] = cur + 1; cur = n.offset; } else { stack[sp++] = n.offset; cur = cur + 1; } }
        } else { if (!sp) break; cur = stack[--sp]; }
    }
    return tmax;
}
}
