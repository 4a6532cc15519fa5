// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_math.h header).
// ref_kats_shapes.cpp: the reference's own sampling / solid-angle tests for shapes whose code this oracle restates, run in C++ for
// speed and reported to tests/test_oracle_kats.py. They pin the area-light half of the path: the
// restatements of Triangle::sample + Shape::sample_interaction (triangle.rs:556-584, shape.rs:40-58), Triangle::intersect_p,
// Sphere::intersect_p, Disk::sample / Disk::intersect_p inside ref_shading.cpp / ref_scene.h / ref_sphere.cpp -- the very functions
// LightSampler::sample_li / pdf_li call -- against the reference's independent estimates:
//   tests/shapes.rs:226-299  triangle_sampling     (sample_interaction pdf vs uniform-sphere Monte Carlo over intersect_p)
//   tests/shapes.rs:301-352  triangle_solid_angle  (the same pdf vs Triangle::solid_angle, Girard's theorem, triangle.rs:586-624)
//   tests/shapes.rs:354-389  mc_solid_angle + sphere_solid_angle (Sphere::intersect_p of a rotated + translated sphere vs 4 pi and
//                            Sphere::solid_angle, sphere.rs:397-407)
//   tests/shapes.rs:407-419  disk_solid_angle      (Disk::intersect_p Monte Carlo vs the default Shape::solid_angle, shape.rs:84-107:
//                            Disk::sample through sample_interaction + a self-occlusion intersect_p)
//   tests/shapes.rs:36-146   triangle_watertight   (Triangle::intersect over a closed triangulated sphere: every ray from inside hits;
//                            the reference keeps this test DISABLED -- `//#[test]` at :35 -- see orc_test_triangle_watertight)
// Same seeds (RNG::new(i) / RNG::new(100 + i)), same sample counts, same radical-inverse points, same tolerances.
#include "ref_shading.h"
#include <cmath>

namespace {
using namespace ref;

Float punif(RNG &rng, Float range) { return lerp(rng.uniform_float(), -range, range); }   // tests/shapes.rs:30-32

// get_random_trianlge (tests/shapes.rs:147-170) as a one-triangle scene with that triangle as light 0 (an area light is how the
// oracle reaches Shape::sample_interaction). Returns false for the degenerate triangles the reference skips.
template <class F> bool random_triangle_scene(F value, Scene &s) {
    V3 v[3];
    for (int j = 0; j < 3; ++j) { v[j].x = value(); v[j].y = value(); v[j].z = value(); }
    if (length_squared(cross(v[1] - v[0], v[2] - v[0])) < 1.0e-20f) return false;
    s.P = {v[0], v[1], v[2]}; s.idx = {0, 1, 2}; s.tri_flags = {0};
    s.prim_shape = {(uint32_t)PT_SHAPE_TRIANGLE << 30}; s.prim_material = {PT_NONE}; s.prim_light = {0};
    PtLight L; std::memset(&L, 0, sizeof L); L.type = PT_LIGHT_DIFFUSE_AREA; L.prim = 0; L.L[0] = L.L[1] = L.L[2] = 1.0f; L.two_sided = 1;
    s.lights = {L};
    return true;
}
// the reference point of both triangle tests (tests/shapes.rs:236-247)
V3 reference_point(RNG &rng, Float range) {
    V3 pc; pc.x = punif(rng, range); pc.y = punif(rng, range); pc.z = punif(rng, range);
    const uint32_t idx = rng.uniform_u32() % 3u;
    const Float v = rng.uniform_float() > 0.5f ? (-range - 3.0f) : (range + 3.0f);
    if (idx == 0) pc.x = v; else if (idx == 1) pc.y = v; else pc.z = v;
    return pc;
}
Float error_measure(Float a, Float b) {   // tests/shapes.rs:277-283: absolute for small solid angles, relative for large
    if (std::fabs(a) < 1.0e-4f || std::fabs(b) < 1.0e-4f) return std::fabs(a - b);
    return std::fabs((a - b) / b);
}
// sum over `count` radical-inverse points of 1 / (count * pdf) with pdf from Shape::sample_interaction at `pc` (the InteractionData of
// tests/shapes.rs:262-264: p = pc, n = 0, p_error = 0); *bad counts the samples whose pdf is not > 0 (the reference asserts it is)
double sample_interaction_estimate(const Scene &s, V3 pc, int count, int *bad) {
    LightSampler ls; ls.init(s, PT_LS_UNIFORM);
    IData ref; ref.p = pc; ref.p_error = V3(0.0f, 0.0f, 0.0f); ref.n = V3(0.0f, 0.0f, 0.0f);
    double est = 0.0;
    for (int j = 0; j < count; ++j) {
        const P2 u(radical_inverse(0, (uint64_t)j), radical_inverse(1, (uint64_t)j));
        V3 wi(0.0f, 0.0f, 0.0f); Float pdf = 0.0f; IData p1;
        (void)ls.sample_li(0, ref, u, wi, pdf, p1);
        if (!(pdf > 0.0f)) { if (bad) ++*bad; continue; }
        est += 1.0 / ((double)count * (double)pdf);
    }
    return est;
}
// Triangle::solid_angle (triangle.rs:586-624): Girard's theorem on the vertices projected onto the unit sphere around p
Float triangle_solid_angle(const Scene &s, V3 p) {
    V3 p0, p1, p2; s.tri_positions(0, p0, p1, p2);
    const V3 a = normalize(p0 - p), b = normalize(p1 - p), c = normalize(p2 - p);
    V3 c01 = cross(a, b), c12 = cross(b, c), c20 = cross(c, a);
    if (length_squared(c01) > 0.0f) c01 = normalize(c01);
    if (length_squared(c12) > 0.0f) c12 = normalize(c12);
    if (length_squared(c20) > 0.0f) c20 = normalize(c20);
    return std::fabs(std::acos(clampv(dot(c01, -c12), -1.0f, 1.0f)) + std::acos(clampv(dot(c12, -c20), -1.0f, 1.0f)) + std::acos(clampv(dot(c20, -c01), -1.0f, 1.0f)) - PI);
}
// mc_solid_angle (tests/shapes.rs:354-367) for quadric 0 of `s`: uniform directions through Shape::intersect_p
Float mc_solid_angle_quadric(const Scene &s, V3 p, int nsamples) {
    int nhits = 0;
    for (int i = 0; i < nsamples; ++i) {
        const P2 u(radical_inverse(0, (uint64_t)i), radical_inverse(1, (uint64_t)i));
        const Ray ray(p, uniform_sample_sphere(u), INF, 0.0f);
        if (s.sphere_intersect_p(0, ray)) ++nhits;
    }
    return (Float)nhits / (INV4_PI * (Float)nsamples);   // uniform_sphere_pdf() = INV4_PI (sampling.rs:220-222)
}
void quadric_scene(const PtSphere &S, Scene &s) {
    s.spheres = {S};
    s.prim_shape = {((uint32_t)PT_SHAPE_SPHERE << 30) | 0u}; s.prim_material = {PT_NONE}; s.prim_light = {0};
    PtLight L; std::memset(&L, 0, sizeof L); L.type = PT_LIGHT_DIFFUSE_AREA; L.prim = 0; L.L[0] = L.L[1] = L.L[2] = 1.0f; L.two_sided = 1;
    s.lights = {L};
}
}  // namespace

extern "C" {
// tests/shapes.rs:226-299. For every seed i < n_seeds (the reference runs 30 with count = 512 * 1024): out[4 i ..] = {unif_estimate,
// tri_sample_estimate, error(...), compared ? 1 : 0}. Returns the number of violated assertions (a pdf that is not > 0, or an error >= 0.1
// where the reference compares: tri_sample_estimate > 1e-3).
int orc_test_triangle_sampling(int n_seeds, int count, double *out) {
    int failures = 0;
    for (int i = 0; i < n_seeds; ++i) {
        const Float range = 10.0f;
        RNG rng((uint64_t)i);
        Scene s;
        if (out) out[4 * i] = out[4 * i + 1] = out[4 * i + 2] = out[4 * i + 3] = 0.0;
        if (!random_triangle_scene([&]() { return punif(rng, range); }, s)) continue;
        const V3 pc = reference_point(rng, range);
        int hits = 0;
        for (int j = 0; j < count; ++j) {   // uniform spherical sampling over Triangle::intersect_p
            const P2 u(radical_inverse(0, (uint64_t)j), radical_inverse(1, (uint64_t)j));
            const Ray ray(pc, uniform_sample_sphere(u), INF, 0.0f);
            Float t, b[3];
            if (s.tri_hit_params(0, ray, t, b)) ++hits;
        }
        const double unif_estimate = (double)hits / ((double)count * (double)INV4_PI);
        int bad = 0;
        const double tri_estimate = sample_interaction_estimate(s, pc, count, &bad);
        failures += bad;
        const bool compared = tri_estimate > 1.0e-3;
        const Float err = error_measure((Float)tri_estimate, (Float)unif_estimate);
        if (compared && !(err < 0.1f)) ++failures;
        if (out) { out[4 * i] = unif_estimate; out[4 * i + 1] = tri_estimate; out[4 * i + 2] = err; out[4 * i + 3] = compared ? 1.0 : 0.0; }
    }
    return failures;
}

// tests/shapes.rs:301-352, seeds RNG::new(100 + i), count = 64 * 1024, tolerance 0.015. out[3 i ..] = {spherical_area, tri_sample_estimate, error}.
int orc_test_triangle_solid_angle(int n_seeds, int count, double *out) {
    int failures = 0;
    for (int i = 0; i < n_seeds; ++i) {
        const Float range = 10.0f;
        RNG rng((uint64_t)(100 + i));
        Scene s;
        if (out) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = 0.0;
        if (!random_triangle_scene([&]() { return punif(rng, range); }, s)) continue;
        const V3 pc = reference_point(rng, range);
        int bad = 0;
        const double tri_estimate = sample_interaction_estimate(s, pc, count, &bad);
        failures += bad;
        const Float area = triangle_solid_angle(s, pc);
        const Float err = error_measure(area, (Float)tri_estimate);
        if (!(err < 0.015f)) ++failures;
        if (out) { out[3 * i] = area; out[3 * i + 1] = tri_estimate; out[3 * i + 2] = err; }
    }
    return failures;
}

// tests/shapes.rs:369-389 on the sphere the caller built (Translate(1, .5, -.8) * RotateX(30), radius 1): out = {mc(pinside),
// solid_angle(pinside), mc(p), solid_angle(p)}; returns the number of violated assertions (written as the reference writes them:
// `mc.abs() - 4 pi < 0.01`, `sa.abs() - 4 pi < 0.01`, `(mc - sa).abs() < 0.001`).
int orc_test_sphere_solid_angle(const PtSphere *S, int nsamples, double *out) {
    Scene s; quadric_scene(*S, s);
    auto solid_angle = [&](V3 p) {   // Sphere::solid_angle (sphere.rs:397-407)
        const V3 pcenter = xf_point(m4_from(S->object_to_world), V3(0.0f, 0.0f, 0.0f));
        if (distance_squared(p, pcenter) <= S->radius * S->radius) return 4.0f * PI;
        const Float sin_theta2 = S->radius * S->radius / distance_squared(p, pcenter);
        const Float cos_theta = std::sqrt(fmax_(1.0f - sin_theta2, 0.0f));
        return 2.0f * PI * (1.0f - cos_theta);
    };
    int failures = 0;
    const V3 pinside(1.0f, 0.9f, -0.8f), p(-1.25f, -1.0f, 0.8f);
    const Float mc_in = mc_solid_angle_quadric(s, pinside, nsamples), sa_in = solid_angle(pinside);
    if (!(std::fabs(mc_in) - 4.0f * PI < 0.01f)) ++failures;
    if (!(std::fabs(sa_in) - 4.0f * PI < 0.01f)) ++failures;
    const Float mcsa = mc_solid_angle_quadric(s, p, nsamples), sa = solid_angle(p);
    if (!(std::fabs(mcsa - sa) < 0.001f)) ++failures;
    if (out) { out[0] = mc_in; out[1] = sa_in; out[2] = mcsa; out[3] = sa; }
    return failures;
}

// tests/shapes.rs:407-419 on the disk the caller built (same transform, height 0, radius 1.25): out = {mc_solid_angle, Shape::solid_angle}.
int orc_test_disk_solid_angle(const PtSphere *D, int nsamples, double *out) {
    Scene s; quadric_scene(*D, s);
    const V3 p(0.5f, -0.8f, 0.5f);
    const Float mc = mc_solid_angle_quadric(s, p, nsamples);
    // Shape::solid_angle (shape.rs:84-107): sample_interaction from p, count 1 / pdf where the segment p -> sample (t_max 0.999) is
    // not blocked by the shape itself
    LightSampler ls; ls.init(s, PT_LS_UNIFORM);
    IData ref; ref.p = p; ref.p_error = V3(0.0f, 0.0f, 0.0f); ref.n = V3(0.0f, 0.0f, 0.0f);
    double acc = 0.0;
    for (int i = 0; i < nsamples; ++i) {
        const P2 u(radical_inverse(0, (uint64_t)i), radical_inverse(1, (uint64_t)i));
        V3 wi(0.0f, 0.0f, 0.0f); Float pdf = 0.0f; IData pshape;
        (void)ls.sample_li(0, ref, u, wi, pdf, pshape);
        // (sample_li returns the sampled interaction only when pdf > 0 and the point differs from p: exactly the samples the sum uses)
        if (!(pdf > 0.0f)) continue;
        const Ray r(p, pshape.p - p, 0.999f, 0.0f);
        if (!s.sphere_intersect_p(0, r)) acc += 1.0 / (double)pdf;
    }
    const Float dsa = (Float)(acc / (double)nsamples);
    if (out) { out[0] = mc; out[1] = dsa; }
    return std::fabs(mc - dsa) < 0.001f ? 0 : 1;
}

// tests/shapes.rs:36-146 triangle_watertight. The mesh (:37-107): a 16 x 16 (theta, phi) grid of vertices on a sphere whose interior
// vertices are pushed out along their normal by 5 * RNG::new(12111).uniform_float(), the two poles coincident, triangulated as a fan at
// each pole and two triangles per quad between (420 triangles over 256 vertices). Two forms:
//   as_written != 0  the vertex loop exactly as the Rust file has it. Its seam branch tests `t == nphi - 1` (:60); with ntheta == nphi that
//                    row was already taken by the `t == ntheta - 1` branch above it, so the branch is dead, the column phi = 2 pi gets radii of
//                    its own and the mesh is NOT closed: a sheet whose two ends at phi = 0 and phi = 2 pi sit at different radii, with a slit in the
//                    half plane y = 0, x > 0 between them. Rays through the slit hit nothing -- which is why the reference carries the
//                    test commented out (`//#[test]`, :35). pbrt-v3's C++ original tests `p == nPhi - 1`.
//   as_written == 0  that branch on `p == nphi - 1`, as evidently meant ("Close it up exactly at the end", :61): vertices.len() - (nphi - 1) is the
//                    row's first vertex, and the mesh is closed. This is the form whose assertion can hold.
// The loop (:109-145): for every seed i < n_seeds, r = RNG::new(i): a point in the ball of radius 0.5 (uniform_sample_sphere * 0.5), a uniform
// direction, t_max = INFINITY; count the triangles Triangle::intersect reports (every triangle against the same t_max, as the test does:
// it calls the shape, not the primitive); then "now tougher: shoot directly at a vertex": d = vertices[r.uniform_int32_2(256)] - o.
// Outputs (any may be null): verts[256 * 3], indices[420 * 3], rays_o / rays_d[2 * n_seeds * 3] (ray 2 i = the random direction of seed i,
// 2 i + 1 = its vertex ray), nhits[2 * n_seeds]. Returns the number of rays with nhits < 1 (the reference's `assert!(nhits >= 1)`).
// cos / sin are the oracle's deterministic ones (ref_math.h), <= 1 ulp from the platform libm the reference calls.
int orc_test_triangle_watertight(int n_seeds, int as_written, float *verts, uint32_t *indices, float *rays_o, float *rays_d, int *nhits) {
    RNG rng(12111);
    const int ntheta = 16, nphi = 16;
    const int nvertices = ntheta * nphi;
    std::vector<V3> vertices; vertices.reserve(nvertices);
    for (int t = 0; t < ntheta; ++t) {
        const Float theta = PI * (Float)t / (Float)(ntheta - 1);
        const Float cos_theta = dm_cosf(theta), sin_theta = dm_sinf(theta);
        for (int p = 0; p < nphi; ++p) {
            const Float phi = 2.0f * PI * (Float)p / (Float)(nphi - 1);
            Float radius = 1.0f;
            if (t == 0) vertices.push_back(V3(0.0f, 0.0f, radius));
            else if (t == ntheta - 1) vertices.push_back(V3(0.0f, 0.0f, -radius));
            else if ((as_written ? t : p) == nphi - 1) vertices.push_back(vertices[vertices.size() - (size_t)(nphi - 1)]);
            else {
                radius += 5.0f * rng.uniform_float();
                const V3 dir(sin_theta * dm_cosf(phi), sin_theta * dm_sinf(phi), cos_theta);   // spherical_direction (geometry.rs:26-32)
                vertices.push_back(V3(0.0f, 0.0f, 0.0f) + dir * radius);
            }
        }
    }
    if ((int)vertices.size() != nvertices) return -1;
    std::vector<uint32_t> idx;
    auto offset = [&](int t, int p) { return (uint32_t)(t * nphi + p); };
    for (int p = 0; p < nphi - 1; ++p) { idx.push_back(offset(0, 0)); idx.push_back(offset(1, p)); idx.push_back(offset(1, p + 1)); }
    for (int t = 1; t < ntheta - 2; ++t)
        for (int p = 0; p < nphi - 1; ++p) {
            idx.push_back(offset(t, p)); idx.push_back(offset(t + 1, p)); idx.push_back(offset(t + 1, p + 1));
            idx.push_back(offset(t, p)); idx.push_back(offset(t + 1, p + 1)); idx.push_back(offset(t, p + 1));
        }
    for (int p = 0; p < nphi - 1; ++p) { idx.push_back(offset(ntheta - 1, 0)); idx.push_back(offset(ntheta - 2, p)); idx.push_back(offset(ntheta - 2, p + 1)); }
    const uint32_t ntris = (uint32_t)(idx.size() / 3);
    Scene s;
    s.P = vertices; s.idx = idx; s.tri_flags.assign(ntris, 0);
    if (verts) for (int i = 0; i < nvertices; ++i) { verts[3 * i] = vertices[i].x; verts[3 * i + 1] = vertices[i].y; verts[3 * i + 2] = vertices[i].z; }
    if (indices) for (size_t i = 0; i < idx.size(); ++i) indices[i] = idx[i];
    auto count_hits = [&](const Ray &ray) {
        int n = 0;
        for (uint32_t tri = 0; tri < ntris; ++tri) { Float t, b[3]; if (s.tri_intersect(tri, ray, t, b)) ++n; }
        return n;
    };
    auto put = [&](float *dst, size_t k, V3 v) { if (dst) { dst[3 * k] = v.x; dst[3 * k + 1] = v.y; dst[3 * k + 2] = v.z; } };
    int failures = 0;
    for (int i = 0; i < n_seeds; ++i) {
        RNG r((uint64_t)i);
        P2 u; u.x = r.uniform_float(); u.y = r.uniform_float();
        const V3 p = V3(0.0f, 0.0f, 0.0f) + uniform_sample_sphere(u) * 0.5f;
        u.x = r.uniform_float(); u.y = r.uniform_float();
        Ray ray(p, uniform_sample_sphere(u), INF, 0.0f);
        int n = count_hits(ray);
        if (n < 1) ++failures;
        put(rays_o, 2 * (size_t)i, ray.o); put(rays_d, 2 * (size_t)i, ray.d); if (nhits) nhits[2 * i] = n;
        const V3 pvertex = vertices[r.uniform_u32_bounded((uint32_t)vertices.size())];
        ray.d = pvertex - ray.o;
        n = count_hits(ray);
        if (n < 1) ++failures;
        put(rays_o, 2 * (size_t)i + 1, ray.o); put(rays_d, 2 * (size_t)i + 1, ray.d); if (nhits) nhits[2 * i + 1] = n;
    }
    return failures;
}

// tests/shapes.rs:490-535 partial_sphere_normal: for i < n_seeds, RNG::new(i): a random partial sphere (pexp(rng, 4) radius, zmin / zmax / phimax each clipped with
// probability 1/2), a ray from pexp(rng, 8)^3 to a random point of the shape's bounding box (normalised with probability 1/2); where Sphere::intersect finds a hit, the
// interaction's normal must point along the hit point: dot(normalize(n), normalize(p)) = 1. The Rust file evaluates `relative_eq!(1.0, dot, epsilon = 1e-5)` and drops the
// result; here it is an assertion. Returns the number of hits with |dot - 1| > 1e-5 (relative_eq's max(|a|, |b|) * epsilon form); *n_tested = hits found; *worst = max |dot - 1|.
int orc_test_partial_sphere_normal(int n_seeds, int *n_tested, double *worst) {
    auto pexp = [](RNG &rng, float e) { const float logu = lerp(rng.uniform_float(), -e, e); return std::pow(10.0f, logu); };
    int failures = 0, tested = 0; double w = 0.0;
    for (int i = 0; i < n_seeds; ++i) {
        RNG rng((uint64_t)i);
        const float radius = pexp(rng, 4.0f);
        const float zmin = (rng.uniform_float() < 0.5f) ? -radius : lerp(rng.uniform_float(), -radius, radius);
        const float zmax = (rng.uniform_float() < 0.5f) ? radius : lerp(rng.uniform_float(), -radius, radius);
        const float phimax = (rng.uniform_float() < 0.5f) ? 360.0f : rng.uniform_float() * 360.0f;
        PtSphere S; std::memset(&S, 0, sizeof S);   // Sphere::new (sphere.rs:31-50)
        for (int k = 0; k < 4; ++k) S.object_to_world[5 * k] = S.world_to_object[5 * k] = 1.0f;
        S.radius = radius;
        S.z_min = clampv(std::fmin(zmin, zmax), -radius, radius); S.z_max = clampv(std::fmax(zmin, zmax), -radius, radius);
        S.theta_min = std::acos(clampv(std::fmin(zmin, zmax) / radius, -1.0f, 1.0f));
        S.theta_max = std::acos(clampv(std::fmax(zmin, zmax) / radius, -1.0f, 1.0f));
        S.phi_max = (PI / 180.0f) * clampv(phimax, 0.0f, 360.0f);
        Scene s; s.spheres.push_back(S);
        V3 o; o.x = pexp(rng, 8.0f); o.y = pexp(rng, 8.0f); o.z = pexp(rng, 8.0f);
        const Bounds3 bbox = s.sphere_world_bound(0);
        V3 t; t.x = rng.uniform_float(); t.y = rng.uniform_float(); t.z = rng.uniform_float();
        const V3 p2 = bbox.lerp3(t);
        Ray r(o, p2 - o, INF, 0.0f);
        if (rng.uniform_float() < 0.5f) r.d = normalize(r.d);
        SurfaceInteraction isect; Float thit;
        if (!s.sphere_intersect(0, r, thit, isect, true)) continue;
        ++tested;
        const Float d = dot(normalize(isect.n), normalize(isect.p));
        const double err = std::fabs((double)d - 1.0);
        if (err > w) w = err;
        if (!(err <= 1.0e-5 * std::fmax(1.0, std::fabs((double)d)))) ++failures;
    }
    if (n_tested) *n_tested = tested;
    if (worst) *worst = w;
    return failures;
}
}  // extern "C"
