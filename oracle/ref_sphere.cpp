// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_math.h header).
// ref_sphere.cpp: Sphere::intersect / intersect_p / object+world bounds with EFloat running error bounds.
//   shapes/sphere.rs:31-305; core/efloat.rs (EFloat ops, quadratic :211-231); core/transform.rs:434-495,543-636
//   (transform_point_error, transform_vector_error, transform_ray_error, transform_point_abs_error,
//    transform_bounds, transform_surface_interaction); core/interaction.rs:186-216 (SurfaceInteraction::new, shape None).
#include "ref_scene.h"
#include "ref_efloat.h"

namespace ref {

static V3 xf_vector_err(const M4 &t, V3 v, V3 &err) {  // transform.rs:510-527
    Float x = v.x, y = v.y, z = v.z;
    Float g = gamma(3);
    err.x = g * (std::fabs(x * t.m[0][0]) + std::fabs(y * t.m[0][1]) + std::fabs(z * t.m[0][2]));
    err.y = g * (std::fabs(x * t.m[1][0]) + std::fabs(y * t.m[1][1]) + std::fabs(z * t.m[1][2]));
    err.z = g * (std::fabs(x * t.m[2][0]) + std::fabs(y * t.m[2][1]) + std::fabs(z * t.m[2][2]));
    return xf_vector(t, v);
}
// Shared part of intersect / intersect_p: returns the object-space hit (shapes/sphere.rs:59-152 == :198-286, including the
// `phi += 2.0 * phi` quirk of intersect_p and of intersect's second branch, App. A #8).
static bool sphere_hit(const PtSphere &S, const Ray &r, bool is_intersect_p, Float &t_out, V3 &p_hit_out, Float &phi_out, Ray &ray_obj) {
    M4 w2o = m4_from(S.world_to_object);
    V3 oerr, derr;
    V3 o = xf_point_err(w2o, r.o, oerr);
    V3 d = xf_vector_err(w2o, r.d, derr);
    Float l2 = length_squared(d);
    if (l2 > 0.0f) { Float dt = dot(vabs(d), oerr) / l2; o = o + d * dt; }  // transform_ray_error :578-590
    ray_obj = Ray(o, d, r.t_max, r.time);
    EFloat ox(o.x, oerr.x), oy(o.y, oerr.y), oz(o.z, oerr.z), dx(d.x, derr.x), dy(d.y, derr.y), dz(d.z, derr.z);
    EFloat a = dx * dx + dy * dy + dz * dz;
    EFloat b = EFloat(2.0f) * (dx * ox + dy * oy + dz * oz);
    EFloat c = ox * ox + oy * oy + oz * oz - EFloat(S.radius) * EFloat(S.radius);
    EFloat t0, t1;
    if (!quadratic(a, b, c, t0, t1)) return false;
    if (t0.high > r.t_max || t1.low <= 0.0f) return false;
    EFloat ts = t0;
    bool used_t1 = false;
    if (ts.low <= 0.0f) { ts = t1; used_t1 = true; if (ts.high > r.t_max) return false; }
    auto refine = [&](Float t, V3 &ph) {
        ph = o + d * t;
        Float s = S.radius / length(ph);
        ph = V3(ph.x * s, ph.y * s, ph.z * s);
        if (ph.x == 0.0f && ph.y == 0.0f) ph.x = 1e-5f * S.radius;
    };
    V3 ph; refine(ts.v, ph);
    Float phi = dm_atan2f(ph.y, ph.x);
    if (phi < 0.0f) phi += is_intersect_p ? 2.0f * phi : 2.0f * PI;
    auto clipped = [&](V3 p, Float ph_) { return (S.z_min > -S.radius && p.z < S.z_min) || (S.z_max < S.radius && p.z > S.z_max) || ph_ > S.phi_max; };
    if (clipped(ph, phi)) {
        if (ts.v == t1.v) return false;   // PartialEq compares v (efloat.rs:205-209)
        (void)used_t1;
        if (t1.high > r.t_max) return false;
        ts = t1;
        refine(ts.v, ph);
        phi = dm_atan2f(ph.y, ph.x);
        if (phi < 0.0f) phi += 2.0f * phi;
        if (clipped(ph, phi)) return false;
    }
    t_out = ts.v; p_hit_out = ph; phi_out = phi;
    return true;
}

Bounds3 Scene::sphere_world_bound(uint32_t si) const {  // shape.rs:23-25 + sphere.rs:53-57 + transform.rs:592-605
    const PtSphere &S = spheres[si];
    M4 o2w = m4_from(S.object_to_world);
    V3 lo(-S.radius, -S.radius, S.z_min), hi(S.radius, S.radius, S.z_max);
    V3 c[8] = {V3(lo.x, lo.y, lo.z), V3(hi.x, lo.y, lo.z), V3(lo.x, hi.y, lo.z), V3(lo.x, lo.y, hi.z),
               V3(lo.x, hi.y, hi.z), V3(hi.x, hi.y, lo.z), V3(hi.x, lo.y, hi.z), V3(hi.x, hi.y, hi.z)};
    V3 p0 = xf_point(o2w, c[0]);
    Bounds3 ret; ret.pmin = p0; ret.pmax = p0;
    for (int i = 1; i < 8; ++i) ret = union_p(ret, xf_point(o2w, c[i]));
    return ret;
}

// ---- Disk (shapes/disk.rs:55-118): the same PtSphere record with kind == PT_QUADRIC_DISK (z_min = height)
static bool disk_hit(const PtSphere &S, const Ray &r, bool is_intersect_p, Float &t_out, V3 &p_hit_out, Float &phi_out, Ray &ray_obj) {
    M4 w2o = m4_from(S.world_to_object);
    V3 oerr, derr;
    V3 o = xf_point_err(w2o, r.o, oerr);
    V3 d = xf_vector_err(w2o, r.d, derr);
    Float l2 = length_squared(d);
    if (l2 > 0.0f) { Float dt = dot(vabs(d), oerr) / l2; o = o + d * dt; }   // transform_ray_error
    ray_obj = Ray(o, d, r.t_max, r.time);
    if (d.z == 0.0f) return false;
    Float t = (S.z_min - o.z) / (is_intersect_p ? d.z : r.d.z);   // `r.d.z`: the world ray, as written at disk.rs:66
    if (t <= 0.0f || t >= r.t_max) return false;
    V3 ph = o + d * t;
    Float dist2 = ph.x * ph.x + ph.y * ph.y;
    if (dist2 > S.radius * S.radius || dist2 < S.inner_radius * S.inner_radius) return false;
    Float phi = dm_atan2f(ph.y, ph.x);
    if (phi < 0.0f) phi += 2.0f * PI;
    if (phi > S.phi_max) return false;
    t_out = t; p_hit_out = ph; phi_out = phi;
    return true;
}

bool Scene::sphere_intersect_p(uint32_t si, const Ray &r) const {
    Float t, phi; V3 ph; Ray ro;
    if (spheres[si].kind == PT_QUADRIC_DISK) return disk_hit(spheres[si], r, true, t, ph, phi, ro);
    return sphere_hit(spheres[si], r, true, t, ph, phi, ro);
}

bool Scene::sphere_intersect(uint32_t si, const Ray &r, Float &thit, SurfaceInteraction &out, bool with_shape) const {
    const PtSphere &S = spheres[si];
    Float t, phi; V3 p_hit; Ray ray;
    if (S.kind == PT_QUADRIC_DISK) {   // disk.rs:55-99
        if (!disk_hit(S, r, false, t, p_hit, phi, ray)) return false;
        Float dist2 = p_hit.x * p_hit.x + p_hit.y * p_hit.y;
        Float r_hit = std::sqrt(dist2);
        Float u = phi / S.phi_max, v = (S.radius - r_hit) / (S.radius - S.inner_radius);
        V3 dpdu(-S.phi_max * p_hit.y, S.phi_max * p_hit.x, 0.0f);
        V3 dpdv = V3(p_hit.x, p_hit.y, 0.0f) * (S.inner_radius - S.radius) / r_hit;
        p_hit.z = S.z_min;
        // SurfaceInteraction::new (interaction.rs:186-216): the normal flips only when the interaction knows its shape
        bool flip = with_shape && ((S.reverse_orientation != 0) != (S.transform_swaps_handedness != 0));
        V3 n = normalize(cross(dpdu, dpdv));
        if (flip) n = -n;
        V3 wo = normalize(-ray.d);
        M4 o2w = m4_from(S.object_to_world), w2o = m4_from(S.world_to_object);
        SurfaceInteraction ret;
        ret.p = xf_point_abs_err(o2w, p_hit, V3(0.0f, 0.0f, 0.0f), ret.p_error);
        ret.n = normalize(xf_normal_inv(w2o, n));
        ret.wo = normalize(xf_vector(o2w, wo));
        ret.uv = P2(u, v);
        ret.dpdu = xf_vector(o2w, dpdu); ret.dpdv = xf_vector(o2w, dpdv);
        ret.sh_n = face_forward(normalize(xf_normal_inv(w2o, n)), ret.n);
        ret.sh_dpdu = ret.dpdu; ret.sh_dpdv = ret.dpdv;
        ret.sh_dndu = V3(0.0f, 0.0f, 0.0f); ret.sh_dndv = V3(0.0f, 0.0f, 0.0f);
        ret.has_shape = with_shape; ret.shape_flip = flip;
        out = ret;
        thit = t;
        return true;
    }
    if (!sphere_hit(S, r, false, t, p_hit, phi, ray)) return false;
    // sphere.rs:148-192
    Float u = phi / S.phi_max;
    Float theta = dm_acosf(clampv(p_hit.z / S.radius, -1.0f, 1.0f));
    Float v = (theta - S.theta_min) / (S.theta_max - S.theta_min);
    Float zradius = std::sqrt(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
    Float inv_radius = 1.0f / zradius;
    Float cos_phi = p_hit.x * inv_radius, sin_phi = p_hit.y * inv_radius;
    V3 dpdu(-S.phi_max * p_hit.y, S.phi_max * p_hit.x, 0.0f);
    V3 dpdv = V3(p_hit.z * cos_phi, p_hit.z * sin_phi, -S.radius * dm_sinf(theta)) * (S.theta_max - S.theta_min);
    // dndu / dndv from the fundamental forms (sphere.rs:165-184)
    V3 d2pduu = V3(p_hit.x, p_hit.y, 0.0f) * -S.phi_max * S.phi_max;
    V3 d2pduv = V3(-sin_phi, cos_phi, 0.0f) * (S.theta_max - S.theta_min) * p_hit.z * S.phi_max;
    V3 d2pdvv = V3(p_hit.x, p_hit.y, p_hit.z) * -(S.theta_max - S.theta_min) * (S.theta_max - S.theta_min);
    Float E = dot(dpdu, dpdu), Fm = dot(dpdu, dpdv), G = dot(dpdv, dpdv);
    V3 Nn = normalize(cross(dpdu, dpdv));
    Float e = dot(Nn, d2pduu), f = dot(Nn, d2pduv), g = dot(Nn, d2pdvv);
    Float inv_EGF2 = 1.0f / (E * G - Fm * Fm);
    V3 dndu = dpdu * (f * Fm - e * G) * inv_EGF2 + dpdv * (e * Fm - f * E) * inv_EGF2;
    V3 dndv = dpdu * (g * Fm - f * G) * inv_EGF2 + dpdv * (f * Fm - g * E) * inv_EGF2;
    V3 p_error = vabs(p_hit) * gamma(5);
    // SurfaceInteraction::new(.., shape = None): n = normalize(dpdu x dpdv), wo = normalize(-ray.d) (interaction.rs:186-216)
    V3 n = normalize(cross(dpdu, dpdv));
    V3 wo = normalize(-ray.d);
    // transform_surface_interaction (transform.rs:607-636)
    M4 o2w = m4_from(S.object_to_world), w2o = m4_from(S.world_to_object);
    SurfaceInteraction ret;
    ret.p = xf_point_abs_err(o2w, p_hit, p_error, ret.p_error);
    ret.n = normalize(xf_normal_inv(w2o, n));
    ret.wo = normalize(xf_vector(o2w, wo));
    ret.uv = P2(u, v);
    ret.dpdu = xf_vector(o2w, dpdu); ret.dpdv = xf_vector(o2w, dpdv);
    ret.sh_n = normalize(xf_normal_inv(w2o, n));
    ret.sh_dpdu = xf_vector(o2w, dpdu); ret.sh_dpdv = xf_vector(o2w, dpdv);
    ret.sh_dndu = xf_normal_inv(w2o, dndu); ret.sh_dndv = xf_normal_inv(w2o, dndv);
    ret.has_shape = false; ret.shape_flip = false;
    ret.sh_n = face_forward(ret.sh_n, ret.n);
    out = ret;
    thit = t;
    return true;
}

}  // namespace ref
