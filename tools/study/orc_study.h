// orc_study.h -- instrumentation of STUDY builds of the oracle only (tools/study/c4_study.py compiles oracle/*.cpp with -DORC_STUDY into
// /tmp and never into oracle/liboracle.so). Counts what the round-5 traversal work needed to know before it was built:
//   * how many packets a leaf visit holds (top-level tree / object trees) and how often more than one of them is hit under the t_max the
//     leaf was entered with (the case a cooperative leaf step has to redo in order);
//   * how many instance entries the object-space root test turns away, and how many of those a conservative WORLD-space test from the
//     instance packet's spare words would have turned away first (box of the instance alone; + the two xz diagonals; an oriented box
//     with bf16 rows), and that none of them rejects an entry the root test admits.
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
enum { OS_ENTRIES = 0, OS_ROOT_FAIL, OS_AABB_REJ, OS_AABB_FALSE, OS_DIAG_REJ, OS_DIAG_FALSE, OS_OBB_REJ, OS_OBB_FALSE, OS_QUAD_REJ, OS_QUAD_FALSE_ROOT /* rejected although the root test admits: fine as long as no leaf is reached */, OS_QUAD_REJ_ROOTFAIL, OS_TOP_LEAF = 16 /* +n_prims (<=7) */,
       OS_OBJ_LEAF = 24 /* +n_prims */, OS_OBJ_LEAF_MULTIHIT = 32, OS_OBJ_LEAF_VISITS_CLOSEST = 33, OS_TOP_LEAF_INST = 40 /* + instances in the leaf */, OS_N = 64 };
inline std::atomic<uint64_t> g_orc_study[OS_N];
extern "C" __attribute__((used)) inline void orc_study_read(uint64_t *out) { for (int i = 0; i < OS_N; ++i) out[i] = g_orc_study[i].load(); }
extern "C" __attribute__((used)) inline void orc_study_reset() { for (int i = 0; i < OS_N; ++i) g_orc_study[i] = 0; }
static inline float os_bf16(float v) { uint32_t u; std::memcpy(&u, &v, 4); u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u; float r; std::memcpy(&r, &u, 4); return r; }
// exact (double) slab test of the segment o + t d, t in (0, tmax), against [lo, hi] along `n` directions given as value pairs (e = coordinate of o, f = of d)
static inline bool os_slabs(int n, const double *e, const double *f, const double *lo, const double *hi, double tmax) {
    double t0 = 0.0, t1 = tmax;
    for (int i = 0; i < n; ++i) {
        if (f[i] == 0.0) { if (e[i] < lo[i] || e[i] > hi[i]) return false; continue; }
        double a = (lo[i] - e[i]) / f[i], b = (hi[i] - e[i]) / f[i]; if (a > b) { double s = a; a = b; b = s; }
        if (a > t0) t0 = a; if (b < t1) t1 = b;
        if (t0 > t1) return false;
    }
    return true;
}
template <class S, class R> void orc_study_instance(const S &sc, uint32_t ii, const R &r, const R &ray) {
    const auto &I = sc.instances[ii]; const auto &O = sc.objects[I.object];
    if (O.n_prims <= 1) return;
    g_orc_study[OS_ENTRIES]++;
    const auto &root = sc.obj_accel[I.object].nodes[0];
    decltype(ray.d) inv(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    int neg[3] = {inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f};
    const bool pass = bounds_intersect_p2(root, ray, inv, neg);
    if (!pass) g_orc_study[OS_ROOT_FAIL]++;
    // world-space corners of the object's root box, inflated by 1e-4 of its extent (the margin a device test would carry)
    double c[8][3]; const float *m = I.instance_to_world;
    double lo3[3], hi3[3];
    for (int k = 0; k < 3; ++k) { double ext = (double)root.bmax[k] - root.bmin[k]; lo3[k] = root.bmin[k] - 1e-4 * ext - 1e-6; hi3[k] = root.bmax[k] + 1e-4 * ext + 1e-6; }
    for (int q = 0; q < 8; ++q) {
        double p[3] = {(q & 1) ? hi3[0] : lo3[0], (q & 2) ? hi3[1] : lo3[1], (q & 4) ? hi3[2] : lo3[2]};
        for (int k = 0; k < 3; ++k) c[q][k] = m[4 * k] * p[0] + m[4 * k + 1] * p[1] + m[4 * k + 2] * p[2] + m[4 * k + 3];
    }
    const double o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
    // (a) box of the instance alone, (b) + x+z and x-z
    double lo[5], hi[5], e[5], f[5];
    for (int k = 0; k < 5; ++k) { lo[k] = 1e300; hi[k] = -1e300; }
    for (int q = 0; q < 8; ++q) {
        const double v[5] = {c[q][0], c[q][1], c[q][2], c[q][0] + c[q][2], c[q][0] - c[q][2]};
        for (int k = 0; k < 5; ++k) { if (v[k] < lo[k]) lo[k] = v[k]; if (v[k] > hi[k]) hi[k] = v[k]; }
    }
    e[0] = o[0]; e[1] = o[1]; e[2] = o[2]; e[3] = o[0] + o[2]; e[4] = o[0] - o[2];
    f[0] = d[0]; f[1] = d[1]; f[2] = d[2]; f[3] = d[0] + d[2]; f[4] = d[0] - d[2];
    const double tm = (double)r.t_max * (1.0 + 1e-5);
    const bool aabb = os_slabs(3, e, f, lo, hi, tm), diag = os_slabs(5, e, f, lo, hi, tm);
    if (!aabb) { g_orc_study[OS_AABB_REJ]++; if (pass) g_orc_study[OS_AABB_FALSE]++; }
    if (!diag) { g_orc_study[OS_DIAG_REJ]++; if (pass) g_orc_study[OS_DIAG_FALSE]++; }
    // (c) oriented box: rows of world_to_instance scaled to the unit cube, rounded to bf16, extents re-measured over the true corners
    const float *w = I.world_to_instance;
    double cw[3] = {0, 0, 0}; for (int q = 0; q < 8; ++q) for (int k = 0; k < 3; ++k) cw[k] += c[q][k] / 8.0;
    double A[3][3], h[3], eo[3], fo[3], lo1[3], hi1[3];
    for (int i = 0; i < 3; ++i) {
        const double hext = 0.5 * (hi3[i] - lo3[i]);
        for (int j = 0; j < 3; ++j) A[i][j] = os_bf16((float)(w[4 * i + j] / hext));
        h[i] = 0.0;
        for (int q = 0; q < 8; ++q) { double v = 0; for (int j = 0; j < 3; ++j) v += A[i][j] * (c[q][j] - cw[j]); if (std::fabs(v) > h[i]) h[i] = std::fabs(v); }
        h[i] *= 1.0 + 1.0 / 128.0;
        eo[i] = 0; fo[i] = 0; for (int j = 0; j < 3; ++j) { eo[i] += A[i][j] * (o[j] - cw[j]); fo[i] += A[i][j] * d[j]; }
        lo1[i] = -h[i]; hi1[i] = h[i];
    }
    {   // (d) "gate record": world-space boxes (inflated) of the four slots of the object's four-wide root record; rejected when the ray misses all four
        const auto &nn = sc.obj_accel[I.object].nodes;
        int slots[4]; int ns = 0;
        if (nn[0].n_prims > 0) slots[ns++] = 0;
        else for (int side = 0; side < 2; ++side) { const int ci = side == 0 ? 1 : (int)nn[0].offset; if (nn[ci].n_prims > 0) slots[ns++] = ci; else { slots[ns++] = ci + 1; slots[ns++] = (int)nn[ci].offset; } }
        bool any = false;
        for (int k = 0; k < ns && !any; ++k) {
            const auto &b = nn[slots[k]];
            double wl[3] = {1e300, 1e300, 1e300}, wh[3] = {-1e300, -1e300, -1e300};
            for (int q = 0; q < 8; ++q) {
                double ext[3], p[3];
                for (int a = 0; a < 3; ++a) { ext[a] = (double)b.bmax[a] - b.bmin[a]; p[a] = ((q >> a) & 1) ? b.bmax[a] + 1e-4 * ext[a] + 1e-6 : b.bmin[a] - 1e-4 * ext[a] - 1e-6; }
                for (int a = 0; a < 3; ++a) { const double v = m[4 * a] * p[0] + m[4 * a + 1] * p[1] + m[4 * a + 2] * p[2] + m[4 * a + 3]; if (v < wl[a]) wl[a] = v; if (v > wh[a]) wh[a] = v; }
            }
            any = os_slabs(3, e, f, wl, wh, tm);
        }
        if (!any) { g_orc_study[OS_QUAD_REJ]++; if (pass) g_orc_study[OS_QUAD_FALSE_ROOT]++; else g_orc_study[OS_QUAD_REJ_ROOTFAIL]++; }
    }
    const bool obb = os_slabs(3, eo, fo, lo1, hi1, tm);
    if (!obb) { g_orc_study[OS_OBB_REJ]++; if (pass) g_orc_study[OS_OBB_FALSE]++; }
}
template <class S, class O, class N, class R> void orc_study_leaf(const S &sc, const O &ord, bool top, const N &node, const R &r, bool any) {
    const uint32_t n = node.n_prims < 7u ? node.n_prims : 7u;
    if (top && !sc.top_refs.empty()) {
        g_orc_study[OS_TOP_LEAF + n]++;
        uint32_t ni = 0; for (uint32_t i = 0; i < node.n_prims; ++i) if (sc.top_ref(ord[node.offset + i]) & PT_TOP_INSTANCE) ni++;
        g_orc_study[OS_TOP_LEAF_INST + (ni < 7u ? ni : 7u)]++;
        return;
    }
    g_orc_study[OS_OBJ_LEAF + n]++;
    if (any) return;
    g_orc_study[OS_OBJ_LEAF_VISITS_CLOSEST]++;
    uint32_t hits = 0;
    for (uint32_t i = 0; i < node.n_prims; ++i) {
        const uint32_t e = top ? sc.top_ref(ord[node.offset + i]) : ord[node.offset + i];
        const uint32_t s = sc.prim_shape[e];
        if ((s >> 30) != PT_SHAPE_TRIANGLE) continue;
        float t, b[3]; R rc = r;
        if (sc.tri_intersect(s & 0x3fffffffu, rc, t, b)) hits++;
    }
    if (hits > 1) g_orc_study[OS_OBJ_LEAF_MULTIHIT]++;
}
