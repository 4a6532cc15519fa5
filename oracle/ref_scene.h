// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_math.h header).
// ref_scene.h: scene container, SAH BVH build + traversal, triangle and sphere intersection.
//   accelerators/bvh.rs:35-375,662-814; core/geometry/bounds.rs:336-410,459-472,507-580;
//   shapes/triangle.rs:101-584; core/primitive.rs:126-153; core/scene.rs:54-66;
//   core/interaction.rs:186-249; core/shape.rs:40-82.
#pragma once
#include <memory>
#include "ref_math.h"
#include "../include/mi355pt.h"
#include <vector>
#ifdef ORC_STUDY   // study builds only (tools/study/orc_study.h): never defined for oracle/liboracle.so
#include "../tools/study/orc_study.h"
#define ORC_STUDY_INSTANCE(sc, ii, r, ray) ::orc_study_instance(sc, ii, r, ray)
#define ORC_STUDY_COUNT(k) (::g_orc_study[k]++)
#define ORC_STUDY_LEAF(sc, ord, top, node, r, any) ::orc_study_leaf(sc, ord, top, node, r, any)
#else
#define ORC_STUDY_INSTANCE(sc, ii, r, ray) ((void)0)
#define ORC_STUDY_COUNT(k) ((void)0)
#define ORC_STUDY_LEAF(sc, ord, top, node, r, any) ((void)0)
#endif

namespace ref {

struct Bounds3 {
    V3 pmin, pmax;
    Bounds3() : pmin(std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()),
                pmax(std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest()) {}
    Bounds3(V3 a, V3 b)  // from_points
        : pmin(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)), pmax(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)) {}
    V3 diagonal() const { return pmax - pmin; }
    int maximum_extent() const {
        V3 d = diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        else if (d.y > d.z) return 1;
        return 2;
    }
    Float surface_area() const { V3 d = diagonal(); return (d.x * d.y + d.x * d.z + d.y * d.z) * 2.0f; }
    V3 offset(V3 p) const {
        V3 o = p - pmin;
        if (pmax.x > pmin.x) o.x /= pmax.x - pmin.x;
        if (pmax.y > pmin.y) o.y /= pmax.y - pmin.y;
        if (pmax.z > pmin.z) o.z /= pmax.z - pmin.z;
        return o;
    }
    V3 lerp3(V3 t) const { return V3(lerp(t.x, pmin.x, pmax.x), lerp(t.y, pmin.y, pmax.y), lerp(t.z, pmin.z, pmax.z)); }
    bool inside(V3 p) const { return p.x >= pmin.x && p.x <= pmax.x && p.y >= pmin.y && p.y <= pmax.y && p.z >= pmin.z && p.z <= pmax.z; }
};
inline Bounds3 union_b(const Bounds3 &a, const Bounds3 &b) {
    Bounds3 r;
    r.pmin = V3(fmin_(a.pmin.x, b.pmin.x), fmin_(a.pmin.y, b.pmin.y), fmin_(a.pmin.z, b.pmin.z));
    r.pmax = V3(fmax_(a.pmax.x, b.pmax.x), fmax_(a.pmax.y, b.pmax.y), fmax_(a.pmax.z, b.pmax.z));
    return r;
}
inline Bounds3 union_p(const Bounds3 &a, V3 p) {
    Bounds3 r;
    r.pmin = V3(fmin_(a.pmin.x, p.x), fmin_(a.pmin.y, p.y), fmin_(a.pmin.z, p.z));
    r.pmax = V3(fmax_(a.pmax.x, p.x), fmax_(a.pmax.y, p.y), fmax_(a.pmax.z, p.z));
    return r;
}

struct Counters {
    uint64_t camera_rays = 0, intersect_tests = 0, shadow_tests = 0, nodes = 0, tri_tests = 0, sphere_tests = 0;
    uint64_t zero_num = 0, zero_den = 0, path_len[16] = {0};
    uint64_t san_nan = 0, san_neg = 0, san_inf = 0, splats = 0;
    uint64_t ref_asserts = 0;   // assert!()s of PathIntegrator::li / VolPathIntegrator::li that would have fired (the reference panics there)
    void add(const Counters &o) {
        camera_rays += o.camera_rays; intersect_tests += o.intersect_tests; shadow_tests += o.shadow_tests;
        nodes += o.nodes; tri_tests += o.tri_tests; sphere_tests += o.sphere_tests; zero_num += o.zero_num;
        zero_den += o.zero_den; for (int i = 0; i < 16; ++i) path_len[i] += o.path_len[i];
        san_nan += o.san_nan; san_neg += o.san_neg; san_inf += o.san_inf; splats += o.splats; ref_asserts += o.ref_asserts;
    }
};

// What Triangle::intersect leaves in `isect` (shapes/triangle.rs:236-392, interaction.rs:186-249).
struct SurfaceInteraction {
    V3 p, p_error, n, wo;
    P2 uv;
    V3 dpdu, dpdv;
    V3 sh_n, sh_dpdu, sh_dpdv;
    V3 sh_dndu, sh_dndv;       // shading.dndu / dndv (bump mapping only)
    bool has_shape = false;    // SurfaceInteraction.shape is Some (triangles; None for spheres, App. A #6)
    bool shape_flip = false;   // shape.reverse_orientation ^ shape.transform_swapshandedness
    uint32_t prim = PT_NONE;
    uint32_t inst = PT_NONE;   // instance (TransformedPrimitive) the hit went through
    Float t = 0;
    Float b[3] = {0, 0, 0};
};

// core/bssrdf.rs:241-268 BSSRDFTable (built by the host, see pbrt-rust_amd/bssrdf.py)
struct BssrdfTable {
    int n_rho = 0, n_radius = 0;
    std::vector<Float> rho_samples, radius_samples, profile, rhoeff, profile_cdf;
    Float eval_profile(int ri, int di) const { return profile[(size_t)ri * n_radius + di]; }
};

struct Scene {
    std::vector<V3> P, N, S;
    std::vector<P2> UV;
    std::vector<uint32_t> idx;
    std::vector<uint8_t> tri_flags;
    std::vector<PtSphere> spheres;
    std::vector<uint32_t> prim_shape, prim_material, prim_light;
    std::vector<PtMaterial> materials;
    std::vector<BssrdfTable> bssrdf_tables;
    std::shared_ptr<struct TextureSet> textures;   // ref_texture.h (null: no textures)
    std::vector<int32_t> tri_alpha, tri_shadow_alpha;   // per triangle: float texture index or -1 (empty: no masks)
    // Triangle::intersect / intersect_p alpha tests (triangle.rs:275-285,497-545); defined in ref_render.cpp
    bool tri_alpha_rejects(uint32_t tri, const Float b[3], bool shadow) const;
    bool tri_has_alpha(uint32_t tri) const { return (!tri_alpha.empty() && tri_alpha[tri] >= 0) || (!tri_shadow_alpha.empty() && tri_shadow_alpha[tri] >= 0); }
    std::vector<PtLight> lights;
    std::vector<uint32_t> infinite_lights;
    uint32_t env_w = 0, env_h = 0;
    std::vector<RGB> env_texels;
    std::vector<Float> env_importance;
    RGB env_power_lookup;
    uint32_t max_node_prims = 4, split_method = PT_SPLIT_SAH;
    // participating media (volpath only): HomogeneousMedium table + per-primitive MediumInterface (PT_NONE = none)
    std::vector<PtMedium> media; std::vector<uint32_t> prim_med_in, prim_med_out;
    // GridDensityMedium (media/grid.rs): own copy of each grid's densities, sigma_t = (sigma_a + sigma_s)[0], 1 / max density (grid.rs:46-60)
    struct GridAux { std::vector<Float> density; Float sigma_t = 0, inv_max_density = 0; };
    std::vector<GridAux> grid_aux;
    std::vector<PtBVHNode> nodes;       // top-level accelerator
    std::vector<uint32_t> ordered;      // positions in the top-level list (see top_ref)
    struct Accel { std::vector<PtBVHNode> nodes; std::vector<uint32_t> ordered; };
    std::vector<PtObject> objects; std::vector<PtInstance> instances; std::vector<uint32_t> top_refs;
    std::vector<Accel> obj_accel;       // one per object (nodes empty when the object has a single primitive)
    uint32_t top_ref(uint32_t pos) const { return top_refs.empty() ? pos : top_refs[pos]; }
    size_t n_top() const { return top_refs.empty() ? prim_shape.size() : top_refs.size(); }
    Bounds3 wb;

    // ---- triangles ----
    void tri_positions(uint32_t tri, V3 &p0, V3 &p1, V3 &p2) const {
        p0 = P[idx[3 * tri]]; p1 = P[idx[3 * tri + 1]]; p2 = P[idx[3 * tri + 2]];
    }
    void tri_uvs(uint32_t tri, P2 uv[3]) const {  // triangle.rs:109-115
        if (tri_flags[tri] & PT_TRI_HAS_UV) { uv[0] = UV[idx[3 * tri]]; uv[1] = UV[idx[3 * tri + 1]]; uv[2] = UV[idx[3 * tri + 2]]; }
        else { uv[0] = P2(0, 0); uv[1] = P2(1, 0); uv[2] = P2(1, 1); }
    }
    Bounds3 tri_world_bound(uint32_t tri) const {  // triangle.rs:130-134
        V3 p0, p1, p2; tri_positions(tri, p0, p1, p2);
        return union_p(Bounds3(p0, p1), p2);
    }
    Float tri_area(uint32_t tri) const {  // triangle.rs:550-554
        V3 p0, p1, p2; tri_positions(tri, p0, p1, p2);
        return 0.5f * length(cross(p1 - p0, p2 - p0));
    }

    // Watertight test common to intersect / intersect_p (triangle.rs:136-233 == :400-495).
    bool tri_hit_params(uint32_t tri, const Ray &r, Float &t, Float b[3]) const {
        V3 p0, p1, p2; tri_positions(tri, p0, p1, p2);
        V3 p0t = p0 - r.o, p1t = p1 - r.o, p2t = p2 - r.o;
        int kz = max_dimension(vabs(r.d));
        int kx = kz + 1; if (kx == 3) kx = 0;
        int ky = kx + 1; if (ky == 3) ky = 0;
        V3 d = permute(r.d, kx, ky, kz);
        p0t = permute(p0t, kx, ky, kz); p1t = permute(p1t, kx, ky, kz); p2t = permute(p2t, kx, ky, kz);
        Float Sx = -d.x / d.z, Sy = -d.y / d.z, Sz = 1.0f / d.z;
        p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
        p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
        p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
        Float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
        Float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
        Float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
        if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
            double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
            e0 = (float)(p2typ1tx - p2txp1ty);
            double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
            e1 = (float)(p0typ2tx - p0txp2ty);
            double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
            e2 = (float)(p1typ0tx - p1txp0ty);
        }
        if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
        Float det = e0 + e1 + e2;
        if (det == 0.0f) return false;
        p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
        Float tscaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
        if (det < 0.0f && (tscaled >= 0.0f || tscaled < r.t_max * det)) return false;
        else if (det > 0.0f && (tscaled <= 0.0f || tscaled >= r.t_max * det)) return false;
        Float invdet = 1.0f / det;
        b[0] = e0 * invdet; b[1] = e1 * invdet; b[2] = e2 * invdet;
        t = tscaled * invdet;
        Float maxzt = max_component(vabs(V3(p0t.z, p1t.z, p2t.z)));
        Float deltaz = gamma(3) * maxzt;
        Float maxxt = max_component(vabs(V3(p0t.x, p1t.x, p2t.x)));
        Float maxyt = max_component(vabs(V3(p0t.y, p1t.y, p2t.y)));
        Float deltax = gamma(5) * (maxxt + maxzt);
        Float deltay = gamma(5) * (maxyt + maxzt);
        Float deltae = 2.0f * (gamma(2) * maxxt * maxyt + deltay * maxxt + deltax * maxyt);
        Float maxe = max_component(vabs(V3(e0, e1, e2)));
        Float deltat = 3.0f * (gamma(3) * maxe * maxzt + deltae * maxzt + deltaz * maxe) * std::fabs(invdet);
        if (t <= deltat) return false;
        return true;
    }
    // dpdu/dpdv + the "intersection is bogus" rejection (triangle.rs:236-264).
    bool tri_partials(uint32_t tri, V3 &dpdu, V3 &dpdv) const {
        V3 p0, p1, p2; tri_positions(tri, p0, p1, p2);
        P2 uv[3]; tri_uvs(tri, uv);
        Float duv02[2] = {uv[0].x - uv[2].x, uv[0].y - uv[2].y};
        Float duv12[2] = {uv[1].x - uv[2].x, uv[1].y - uv[2].y};
        V3 dp02 = p0 - p2, dp12 = p1 - p2;
        Float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
        bool degenerateuv = std::fabs(determinant) < 1.0e-8f;
        dpdu = V3(); dpdv = V3();
        if (!degenerateuv) {
            Float invdet = 1.0f / determinant;
            dpdu = (dp02 * duv12[1] - dp12 * duv02[1]) * invdet;
            dpdv = (dp02 * -duv12[0] + dp12 * duv02[0]) * invdet;
        }
        if (degenerateuv || length_squared(cross(dpdu, dpdv)) == 0.0f) {
            V3 ng = cross(p2 - p0, p1 - p0);
            if (length_squared(ng) == 0.0f) return false;
            coordinate_system(normalize(ng), dpdu, dpdv);
        }
        return true;
    }
    // Triangle::intersect (triangle.rs:136-398); `with_shape` = the `s: Option<Arc<Shapes>>` argument.
    bool tri_intersect(uint32_t tri, const Ray &r, Float &t, Float b[3]) const {
        if (!tri_hit_params(tri, r, t, b)) return false;
        V3 dpdu, dpdv;
        return tri_partials(tri, dpdu, dpdv);
    }
    void tri_fill_interaction(uint32_t tri, const Ray &r, Float t, const Float b[3], bool with_shape, SurfaceInteraction &si) const {
        V3 p0, p1, p2; tri_positions(tri, p0, p1, p2);
        P2 uv[3]; tri_uvs(tri, uv);
        V3 dpdu, dpdv; tri_partials(tri, dpdu, dpdv);
        V3 dp02 = p0 - p2, dp12 = p1 - p2;
        Float b0 = b[0], b1 = b[1], b2 = b[2];
        Float xabs = std::fabs(b0 * p0.x) + std::fabs(b1 * p1.x) + std::fabs(b2 * p2.x);
        Float yabs = std::fabs(b0 * p0.y) + std::fabs(b1 * p1.y) + std::fabs(b2 * p2.y);
        Float zabs = std::fabs(b0 * p0.z) + std::fabs(b1 * p1.z) + std::fabs(b2 * p2.z);
        si.p_error = V3(xabs, yabs, zabs) * gamma(7);
        si.p = p0 * b0 + p1 * b1 + p2 * b2;
        si.uv = P2(uv[0].x * b0 + uv[1].x * b1 + uv[2].x * b2, uv[0].y * b0 + uv[1].y * b1 + uv[2].y * b2);
        si.dpdu = dpdu; si.dpdv = dpdv;
        si.sh_dpdu = dpdu; si.sh_dpdv = dpdv;
        si.sh_dndu = V3(0, 0, 0); si.sh_dndv = V3(0, 0, 0);
        si.t = t; si.b[0] = b0; si.b[1] = b1; si.b[2] = b2;
        uint8_t fl = tri_flags[tri];
        bool flip = ((fl & PT_TRI_REVERSE_ORIENTATION) != 0) ^ ((fl & PT_TRI_SWAPS_HANDEDNESS) != 0);
        si.has_shape = with_shape; si.shape_flip = flip;
        V3 nn = normalize(cross(dp02, dp12));
        si.n = nn; si.sh_n = nn;
        si.wo = -r.d;  // triangle.rs:296 (not normalised)
        if (flip) { si.n = -nn; si.sh_n = -nn; }
        if (fl & (PT_TRI_HAS_N | PT_TRI_HAS_S)) {
            uint32_t i0 = idx[3 * tri], i1 = idx[3 * tri + 1], i2 = idx[3 * tri + 2];
            V3 ns;
            if (fl & PT_TRI_HAS_N) {
                ns = N[i0] * b0 + N[i1] * b1 + N[i2] * b2;
                if (length_squared(ns) > 0.0f) ns = normalize(ns); else ns = si.n;
            } else ns = si.n;
            V3 ss;
            if (fl & PT_TRI_HAS_S) {
                ss = S[i0] * b0 + S[i1] * b1 + S[i2] * b2;
                if (length_squared(ss) > 0.0f) ss = normalize(ss); else ss = normalize(si.dpdu);
            } else ss = normalize(si.dpdu);
            V3 ts = cross(ss, ns);
            if (length_squared(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
            else coordinate_system(ns, ss, ts);
            if (fl & PT_TRI_HAS_N) {  // dndu / dndv, triangle.rs:349-386
                Float duv02x = uv[0].x - uv[2].x, duv02y = uv[0].y - uv[2].y, duv12x = uv[1].x - uv[2].x, duv12y = uv[1].y - uv[2].y;
                V3 dn1 = N[i0] - N[i2], dn2 = N[i1] - N[i2];
                Float det = duv02x * duv12y - duv02y * duv12x;
                if (std::fabs(det) < 1.0e-8f) {
                    V3 dn = cross(N[i2] - N[i0], N[i1] - N[i0]);
                    if (length_squared(dn) != 0.0f) coordinate_system(dn, si.sh_dndu, si.sh_dndv);
                } else {
                    Float invdet = 1.0f / det;
                    si.sh_dndu = (dn1 * duv12y - dn2 * duv02y) * invdet;
                    si.sh_dndv = (dn1 * -duv12x + dn2 * duv02x) * invdet;
                }
            }
            if (fl & PT_TRI_REVERSE_ORIENTATION) ts = -ts;
            // set_shading_geometry(ss, ts, dndu, dndv, true)  interaction.rs:228-249
            si.sh_n = normalize(cross(ss, ts));
            if (with_shape) {
                if (flip) si.sh_n = -si.sh_n;
                si.n = face_forward(si.n, si.sh_n);
            }
            si.sh_dpdu = ss; si.sh_dpdv = ts;
        }
    }

    // ---- BVH ----
    Bounds3 prim_world_bound(uint32_t prim) const;
    Bounds3 ref_world_bound(uint32_t ref) const;
    void build_bvh();
    void build_object_accels();
    bool accel_intersect(const std::vector<PtBVHNode> &nn, const std::vector<uint32_t> &ord, bool top, Ray &r, SurfaceInteraction &si, Counters &c) const;
    bool accel_intersect_p(const std::vector<PtBVHNode> &nn, const std::vector<uint32_t> &ord, bool top, const Ray &r, Counters &c) const;
    bool ref_intersect(uint32_t ref, Ray &r, SurfaceInteraction &si, Counters &c) const;
    bool ref_intersect_p(uint32_t ref, const Ray &r, Counters &c) const;
    bool intersect(Ray &r, SurfaceInteraction &si, Counters &c) const;  // Scene::intersect
    bool intersect_p(const Ray &r, Counters &c) const;                 // Scene::intersect_p
    bool prim_intersect(uint32_t prim, Ray &r, SurfaceInteraction &si, Counters &c) const;
    bool prim_intersect_p(uint32_t prim, const Ray &r, Counters &c) const;
    // sphere (shapes/sphere.rs) -- ref_sphere.h
    Bounds3 sphere_world_bound(uint32_t s) const;
    bool sphere_intersect(uint32_t s, const Ray &r, Float &thit, SurfaceInteraction &si, bool with_shape) const;
    bool sphere_intersect_p(uint32_t s, const Ray &r) const;
};

// Bounds3f::intersect_p2 (bounds.rs:559-580)
inline bool bounds_intersect_p2(const PtBVHNode &n, const Ray &ray, V3 inv_dir, const int neg[3]) {
    const float *bb[2] = {n.bmin, n.bmax};
    Float tmin = (bb[neg[0]][0] - ray.o.x) * inv_dir.x;
    Float tmax = (bb[1 - neg[0]][0] - ray.o.x) * inv_dir.x;
    Float tymin = (bb[neg[1]][1] - ray.o.y) * inv_dir.y;
    Float tymax = (bb[1 - neg[1]][1] - ray.o.y) * inv_dir.y;
    tmax *= 1.0f + 2.0f * gamma(3);
    tymax *= 1.0f + 2.0f * gamma(3);
    if (tmin > tymax || tymin > tmax) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    Float tzmin = (bb[neg[2]][2] - ray.o.z) * inv_dir.z;
    Float tzmax = (bb[1 - neg[2]][2] - ray.o.z) * inv_dir.z;
    tzmax *= 1.0f + 2.0f * gamma(3);
    if (tmin > tzmax || tzmin > tmax) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    return (tmin < ray.t_max) && (tmax > 0.0f);
}

inline Bounds3 Scene::prim_world_bound(uint32_t prim) const {
    uint32_t s = prim_shape[prim];
    if ((s >> 30) == PT_SHAPE_TRIANGLE) return tri_world_bound(s & 0x3fffffffu);
    return sphere_world_bound(s & 0x3fffffffu);
}

// GeometricPrimitive::intersect (primitive.rs:126-149)
inline bool Scene::prim_intersect(uint32_t prim, Ray &r, SurfaceInteraction &si, Counters &c) const {
    uint32_t s = prim_shape[prim], k = s >> 30, i = s & 0x3fffffffu;
    if (k == PT_SHAPE_TRIANGLE) {
        c.tri_tests++;
        Float t, b[3];
        if (!tri_intersect(i, r, t, b)) return false;
        if (tri_has_alpha(i) && tri_alpha_rejects(i, b, false)) return false;   // triangle.rs:275-285 (test_alpha_texture = true)
        r.t_max = t;
        si.prim = prim; si.inst = PT_NONE; si.t = t; si.b[0] = b[0]; si.b[1] = b[1]; si.b[2] = b[2];
        return true;
    }
    c.sphere_tests++;
    Float t;
    SurfaceInteraction tmp;
    if (!sphere_intersect(i, r, t, tmp, true)) return false;
    r.t_max = t;
    si = tmp; si.prim = prim; si.inst = PT_NONE; si.t = t;
    return true;
}
inline bool Scene::prim_intersect_p(uint32_t prim, const Ray &r, Counters &c) const {
    uint32_t s = prim_shape[prim], k = s >> 30, i = s & 0x3fffffffu;
    if (k == PT_SHAPE_TRIANGLE) {
        c.tri_tests++;
        Float t, b[3];
        if (!tri_hit_params(i, r, t, b)) return false;
        if (!tri_has_alpha(i)) return true;              // triangle.rs:497 branch not taken
        V3 dpdu, dpdv;
        if (!tri_partials(i, dpdu, dpdv)) return false;  // :516-523: with an alpha mask intersect_p rejects degenerate triangles too
        return !tri_alpha_rejects(i, b, true);
    }
    c.sphere_tests++;
    return sphere_intersect_p(i, r);
}

// TransformedPrimitive::intersect / intersect_p (primitive.rs:58-88), static transform
inline bool Scene::ref_intersect(uint32_t ref, Ray &r, SurfaceInteraction &si, Counters &c) const {
    if (!(ref & PT_TOP_INSTANCE)) return prim_intersect(ref, r, si, c);
    const PtInstance &I = instances[ref & ~PT_TOP_INSTANCE];
    const PtObject &O = objects[I.object];
    Ray ray = xf_ray(m4_from(I.world_to_instance), r);
    ORC_STUDY_INSTANCE(*this, ref & ~PT_TOP_INSTANCE, r, ray);
    SurfaceInteraction tmp = si;
    bool hit = (O.n_prims == 1) ? prim_intersect(O.first_prim, ray, tmp, c)
                                : accel_intersect(obj_accel[I.object].nodes, obj_accel[I.object].ordered, false, ray, tmp, c);
    if (!hit) return false;
    r.t_max = ray.t_max;  // primitive.rs:70 (the dt of transform_ray is not added back)
    si = tmp; si.inst = ref & ~PT_TOP_INSTANCE;
    return true;
}
inline bool Scene::ref_intersect_p(uint32_t ref, const Ray &r, Counters &c) const {
    if (!(ref & PT_TOP_INSTANCE)) return prim_intersect_p(ref, r, c);
    const PtInstance &I = instances[ref & ~PT_TOP_INSTANCE];
    const PtObject &O = objects[I.object];
    Ray ray = xf_ray(m4_from(I.world_to_instance), r);
    ORC_STUDY_INSTANCE(*this, ref & ~PT_TOP_INSTANCE, r, ray);
    return (O.n_prims == 1) ? prim_intersect_p(O.first_prim, ray, c) : accel_intersect_p(obj_accel[I.object].nodes, obj_accel[I.object].ordered, false, ray, c);
}

// BVHAccel::intersect (bvh.rs:705-760)
inline bool Scene::accel_intersect(const std::vector<PtBVHNode> &nn, const std::vector<uint32_t> &ord, bool top, Ray &r, SurfaceInteraction &si, Counters &c) const {
    if (nn.empty()) return false;
    bool hit = false;
    V3 inv_dir(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    int neg[3] = {inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f};
    uint32_t to_visit = 0, cur = 0;
    uint32_t stack[64];
    for (;;) {
        const PtBVHNode &node = nn[cur];
        c.nodes++;
        if (bounds_intersect_p2(node, r, inv_dir, neg)) {
            if (node.n_prims > 0) {
                ORC_STUDY_LEAF(*this, ord, top, node, r, false);
                for (uint32_t i = 0; i < node.n_prims; ++i) {
                    uint32_t e = ord[node.offset + i];
                    if (ref_intersect(top ? top_ref(e) : e, r, si, c)) hit = true;
                }
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            } else {
                if (neg[node.axis]) { stack[to_visit++] = cur + 1; cur = node.offset; }
                else { stack[to_visit++] = node.offset; cur = cur + 1; }
            }
        } else {
            if (to_visit == 0) break;
            cur = stack[--to_visit];
        }
    }
    return hit;
}
// Scene::intersect (scene.rs:54-59) + deferred construction of the SurfaceInteraction of the final hit
inline bool Scene::intersect(Ray &r, SurfaceInteraction &si, Counters &c) const {
    c.intersect_tests++;
    Ray r0 = r;
    si.inst = PT_NONE;
    bool hit = accel_intersect(nodes, ordered, true, r, si, c);
    if (hit) {
        uint32_t s = prim_shape[si.prim];
        const bool inst = si.inst != PT_NONE;
        Ray rl = inst ? xf_ray(m4_from(instances[si.inst].world_to_instance), r0) : r0;
        if ((s >> 30) == PT_SHAPE_TRIANGLE) {
            Float t = si.t, b[3] = {si.b[0], si.b[1], si.b[2]};
            uint32_t prim = si.prim, ii = si.inst;
            tri_fill_interaction(s & 0x3fffffffu, rl, t, b, true, si);
            si.prim = prim; si.inst = ii;
        }
        if (inst) {  // transform_surface_interaction (transform.rs:607-636) unless the instance transform is the identity
            M4 i2w = m4_from(instances[si.inst].instance_to_world), w2i = m4_from(instances[si.inst].world_to_instance);
            if (!m4_is_identity(i2w)) {
                SurfaceInteraction ret = si;
                ret.p = xf_point_abs_err(i2w, si.p, si.p_error, ret.p_error);
                ret.n = normalize(xf_normal_inv(w2i, si.n));
                ret.wo = normalize(xf_vector(i2w, si.wo));
                ret.dpdu = xf_vector(i2w, si.dpdu); ret.dpdv = xf_vector(i2w, si.dpdv);
                ret.sh_n = normalize(xf_normal_inv(w2i, si.sh_n));
                ret.sh_dpdu = xf_vector(i2w, si.sh_dpdu); ret.sh_dpdv = xf_vector(i2w, si.sh_dpdv);
                ret.sh_dndu = xf_normal_inv(w2i, si.sh_dndu); ret.sh_dndv = xf_normal_inv(w2i, si.sh_dndv);
                ret.sh_n = face_forward(ret.sh_n, ret.n);
                si = ret;
            }
        }
    }
    return hit;
}
// BVHAccel::intersect_p (bvh.rs:762-814)
inline bool Scene::accel_intersect_p(const std::vector<PtBVHNode> &nn, const std::vector<uint32_t> &ord, bool top, const Ray &r, Counters &c) const {
    if (nn.empty()) return false;
    V3 inv_dir(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
    int neg[3] = {inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f};
    uint32_t to_visit = 0, cur = 0;
    uint32_t stack[64];
    for (;;) {
        const PtBVHNode &node = nn[cur];
        c.nodes++;
        if (bounds_intersect_p2(node, r, inv_dir, neg)) {
            if (node.n_prims > 0) {
                ORC_STUDY_LEAF(*this, ord, top, node, r, true);
                for (uint32_t i = 0; i < node.n_prims; ++i) {
                    uint32_t e = ord[node.offset + i];
                    if (ref_intersect_p(top ? top_ref(e) : e, r, c)) return true;
                }
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            } else {
                if (neg[node.axis]) { stack[to_visit++] = cur + 1; cur = node.offset; }
                else { stack[to_visit++] = node.offset; cur = cur + 1; }
            }
        } else {
            if (to_visit == 0) break;
            cur = stack[--to_visit];
        }
    }
    return false;
}
inline bool Scene::intersect_p(const Ray &r, Counters &c) const {  // Scene::intersect_p (scene.rs:61-66)
    c.shadow_tests++;
    return accel_intersect_p(nodes, ordered, true, r, c);
}

// ---- SAH build (bvh.rs:145-375, 662-693) ---------------------------------------------------
namespace bvhbuild {
struct PrimInfo { uint32_t number; Bounds3 bounds; V3 centroid; };
struct BuildNode { Bounds3 bounds; int left = -1, right = -1; int axis = 0; uint32_t first = 0, n = 0; };

// core::iter::Iterator::partition_in_place (Rust nightly) as used at bvh.rs:292,362:
// repeatedly find the first `false` from the front and swap it with the last `true` from the back.
template <class Pred> size_t partition_in_place(PrimInfo *a, size_t n, Pred pred) {
    size_t i = 0, j = n, true_count = 0;
    for (;;) {
        while (i < j && pred(a[i])) { ++i; ++true_count; }
        if (i >= j) break;
        size_t head = i++;
        while (j > i && !pred(a[j - 1])) --j;
        if (j <= i) break;
        size_t tail = --j;
        std::swap(a[head], a[tail]);
        ++true_count;
    }
    return true_count;
}

struct Builder {
    uint32_t max_prims;
    std::vector<PrimInfo> info;
    std::vector<BuildNode> arena;
    std::vector<uint32_t> ordered;
    explicit Builder(uint32_t mp) : max_prims(std::min(255u, mp)) {}

    int make_leaf(int node, size_t start, size_t end, const Bounds3 &bounds) {
        uint32_t off = (uint32_t)ordered.size();
        for (size_t i = start; i < end; ++i) ordered.push_back(info[i].number);
        arena[node].first = off; arena[node].n = (uint32_t)(end - start); arena[node].bounds = bounds;
        return node;
    }
    int recursive_build(size_t start, size_t end) {
        int node = (int)arena.size();
        arena.push_back(BuildNode());
        Bounds3 bounds;
        for (size_t i = start; i < end; ++i) bounds = union_b(bounds, info[i].bounds);
        size_t nprims = end - start;
        if (nprims == 1) return make_leaf(node, start, end, bounds);
        Bounds3 cb;
        for (size_t i = start; i < end; ++i) cb = union_p(cb, info[i].centroid);
        int dim = cb.maximum_extent();
        size_t mid = (start + end) / 2;
        if (cb.pmax[dim] == cb.pmin[dim]) return make_leaf(node, start, end, bounds);
        // split_sah (bvh.rs:308-375)
        if (nprims <= 2) {
            mid = (start + end) / 2;
            if (start != end - 1 && info[end - 1].centroid[dim] < info[start].centroid[dim]) std::swap(info[start], info[end - 1]);
        } else {
            const int NB = 12;
            struct Bucket { size_t count = 0; Bounds3 bounds; } buckets[NB];
            for (size_t i = start; i < end; ++i) {
                size_t b = (size_t)f2u_sat((Float)NB * cb.offset(info[i].centroid)[dim]);
                if (b == (size_t)NB) b = NB - 1;
                buckets[b].count++;
                buckets[b].bounds = union_b(buckets[b].bounds, info[i].bounds);
            }
            Float cost[NB - 1];
            for (int i = 0; i < NB - 1; ++i) {
                Bounds3 b0, b1; size_t c0 = 0, c1 = 0;
                for (int j = 0; j <= i; ++j) { b0 = union_b(b0, buckets[j].bounds); c0 += buckets[j].count; }
                for (int j = i + 1; j < NB; ++j) { b1 = union_b(b1, buckets[j].bounds); c1 += buckets[j].count; }
                cost[i] = 1.0f + ((Float)c0 * b0.surface_area() + (Float)c1 * b1.surface_area()) / bounds.surface_area();
            }
            Float min_cost = cost[0]; int min_bucket = 0;
            for (int i = 1; i < NB - 1; ++i) if (cost[i] < min_cost) { min_cost = cost[i]; min_bucket = i; }
            Float leaf_cost = (Float)nprims;
            if (nprims > max_prims || min_cost < leaf_cost) {
                size_t pm = partition_in_place(info.data() + start, nprims, [&](const PrimInfo &pi) {
                    size_t b = (size_t)f2u_sat((Float)NB * cb.offset(pi.centroid)[dim]);
                    if (b == (size_t)NB) b = NB - 1;
                    return (int)b <= min_bucket;
                });
                mid = pm + start;
            } else {
                return make_leaf(node, start, end, bounds);
            }
        }
        int right = recursive_build(mid, end);   // right subtree first (bvh.rs:275-276)
        int left = recursive_build(start, mid);
        arena[node].left = left; arena[node].right = right;
        arena[node].bounds = union_b(arena[left].bounds, arena[right].bounds);
        arena[node].axis = dim; arena[node].n = 0;
        return node;
    }
    uint32_t flatten(std::vector<PtBVHNode> &out, int node, uint32_t &offset) {
        uint32_t my = offset++;
        const BuildNode &bn = arena[node];
        PtBVHNode ln; std::memset(&ln, 0, sizeof ln);
        ln.bmin[0] = bn.bounds.pmin.x; ln.bmin[1] = bn.bounds.pmin.y; ln.bmin[2] = bn.bounds.pmin.z;
        ln.bmax[0] = bn.bounds.pmax.x; ln.bmax[1] = bn.bounds.pmax.y; ln.bmax[2] = bn.bounds.pmax.z;
        if (bn.n > 0) { ln.n_prims = (uint16_t)bn.n; ln.offset = bn.first; ln.axis = 0; out[my] = ln; }
        else {
            flatten(out, bn.left, offset);
            ln.n_prims = 0; ln.axis = (uint8_t)bn.axis;
            ln.offset = flatten(out, bn.right, offset);
            out[my] = ln;
        }
        return my;
    }
};
}  // namespace bvhbuild

namespace hlbvh { void build(const std::vector<Bounds3> &bounds, uint32_t max_node_prims, std::vector<PtBVHNode> &nodes, std::vector<uint32_t> &ordered); }  // ref_hlbvh.h
// BVHAccel::new over `bounds[i]` (item i keeps number i): fills nodes + ordered item numbers
inline void build_accel(const std::vector<Bounds3> &bounds, uint32_t max_node_prims, std::vector<PtBVHNode> &nodes, std::vector<uint32_t> &ordered, uint32_t split_method = PT_SPLIT_SAH) {
    if (split_method == PT_SPLIT_HLBVH) { hlbvh::build(bounds, max_node_prims, nodes, ordered); return; }
    nodes.clear(); ordered.clear();
    size_t n = bounds.size();
    if (n == 0) return;
    bvhbuild::Builder b(max_node_prims);
    b.info.resize(n);
    for (size_t i = 0; i < n; ++i) {
        b.info[i].number = (uint32_t)i; b.info[i].bounds = bounds[i];
        b.info[i].centroid = bounds[i].pmin * 0.5f + bounds[i].pmax * 0.5f;
    }
    b.arena.reserve(2 * n);
    int root = b.recursive_build(0, n);
    nodes.resize(b.arena.size());
    uint32_t off = 0;
    b.flatten(nodes, root, off);
    ordered.swap(b.ordered);
}
inline Bounds3 xf_bounds(const M4 &t, const Bounds3 &b) {  // transform.rs:592-605
    V3 lo = b.pmin, hi = b.pmax;
    V3 c[8] = {V3(lo.x, lo.y, lo.z), V3(hi.x, lo.y, lo.z), V3(lo.x, hi.y, lo.z), V3(lo.x, lo.y, hi.z),
               V3(lo.x, hi.y, hi.z), V3(hi.x, hi.y, lo.z), V3(hi.x, lo.y, hi.z), V3(hi.x, hi.y, hi.z)};
    V3 p0 = xf_point(t, c[0]);
    Bounds3 ret; ret.pmin = p0; ret.pmax = p0;
    for (int i = 1; i < 8; ++i) ret = union_p(ret, xf_point(t, c[i]));
    return ret;
}
inline Bounds3 Scene::ref_world_bound(uint32_t ref) const {
    if (!(ref & PT_TOP_INSTANCE)) return prim_world_bound(ref);
    const PtInstance &I = instances[ref & ~PT_TOP_INSTANCE];
    const PtObject &O = objects[I.object];
    Bounds3 inner;
    if (O.n_prims == 1) inner = prim_world_bound(O.first_prim);
    else { const PtBVHNode &r = obj_accel[I.object].nodes[0]; inner.pmin = V3(r.bmin[0], r.bmin[1], r.bmin[2]); inner.pmax = V3(r.bmax[0], r.bmax[1], r.bmax[2]); }
    return xf_bounds(m4_from(I.instance_to_world), inner);  // TransformedPrimitive::world_bound -> motion_bounds (transform.rs:1564-1567)
}
inline void Scene::build_object_accels() {  // api.rs:1692-1700: one BVH per object with more than one primitive
    obj_accel.assign(objects.size(), Accel());
    for (size_t o = 0; o < objects.size(); ++o) {
        const PtObject &O = objects[o];
        if (O.n_prims <= 1) continue;
        std::vector<Bounds3> bb(O.n_prims);
        for (uint32_t i = 0; i < O.n_prims; ++i) bb[i] = prim_world_bound(O.first_prim + i);
        build_accel(bb, max_node_prims, obj_accel[o].nodes, obj_accel[o].ordered, split_method);
        for (auto &e : obj_accel[o].ordered) e += O.first_prim;  // item number -> primitive index
    }
}
inline void Scene::build_bvh() {
    size_t n = n_top();
    std::vector<Bounds3> bb(n);
    for (size_t i = 0; i < n; ++i) bb[i] = ref_world_bound(top_ref((uint32_t)i));
    build_accel(bb, max_node_prims, nodes, ordered, split_method);
}

}  // namespace ref
