// dev_sphere.h -- Sphere::intersect / intersect_p on device with EFloat running error bounds.
//   shapes/sphere.rs:59-305; core/efloat.rs; core/transform.rs:434-495,510-527,578-590,607-636;
//   core/interaction.rs:186-216 (SurfaceInteraction::new with shape = None, App. A #6).
#pragma once
#include "dev_scene.h"

namespace ptd {

struct EFloat {   // core/efloat.rs:4-28
    float v, low, high;
    PT_DEV EFloat() : v(0), low(0), high(0) {}
    PT_DEV EFloat(float v_, float err) : v(v_) {
        if (err == 0.0f) { low = v; high = v; }
        else { low = next_float_down(v - err); high = next_float_up(v + err); }
    }
    PT_DEV explicit EFloat(float f) : v(f), low(f), high(f) {}
};
PT_DEV EFloat operator+(EFloat a, EFloat b) { EFloat r; r.v = a.v + b.v; r.low = next_float_down(a.low + b.low); r.high = next_float_up(a.high + b.high); return r; }
PT_DEV EFloat operator-(EFloat a, EFloat b) { EFloat r; r.v = a.v - b.v; r.low = next_float_down(a.low - b.high); r.high = next_float_up(a.high - b.low); return r; }
PT_DEV EFloat operator*(EFloat a, EFloat b) {
    EFloat r; r.v = a.v * b.v;
    const float p0 = a.low * b.low, p1 = a.high * b.low, p2 = a.low * b.high, p3 = a.high * b.high;
    r.low = next_float_down(minf(minf(p0, p1), minf(p2, p3)));
    r.high = next_float_up(maxf(maxf(p0, p1), maxf(p2, p3)));
    return r;
}
PT_DEV EFloat operator/(EFloat a, EFloat b) {  // efloat.rs:124-146 (straddle test on the numerator, as written there)
    EFloat r; r.v = a.v / b.v;
    if (a.low < 0.0f && a.high > 0.0f) { r.low = -PT_INF; r.high = PT_INF; }
    else {
        const float d0 = a.low / b.low, d1 = a.high / b.low, d2 = a.low / b.high, d3 = a.high / b.high;
        r.low = next_float_down(minf(minf(d0, d1), minf(d2, d3)));
        r.high = next_float_up(maxf(maxf(d0, d1), maxf(d2, d3)));
    }
    return r;
}
PT_DEV bool ef_quadratic(EFloat a, EFloat b, EFloat c, EFloat &t0, EFloat &t1) {  // efloat.rs:211-231
    const double discrim = (double)b.v * (double)b.v - 4.0 * (double)a.v * (double)c.v;
    if (discrim < 0.0) return false;
    const double root = __builtin_sqrt(discrim);
    const EFloat frd((float)root, (float)((double)kMachEps * root));
    const EFloat q = (b.v < 0.0f) ? (EFloat(-0.5f) * (b - frd)) : (EFloat(-0.5f) * (b + frd));
    t0 = q / a; t1 = c / q;
    if (t0.v > t1.v) { EFloat tmp = t0; t0 = t1; t1 = tmp; }
    return true;
}
PT_DEV V3 xf_vector_err(const M4 &t, V3 v, V3 &err) {  // transform.rs:510-527
    const float x = v.x, y = v.y, z = v.z, g = gammaf(3);
    err.x = g * (fabsf(x * t.m[0]) + fabsf(y * t.m[1]) + fabsf(z * t.m[2]));
    err.y = g * (fabsf(x * t.m[4]) + fabsf(y * t.m[5]) + fabsf(z * t.m[6]));
    err.z = g * (fabsf(x * t.m[8]) + fabsf(y * t.m[9]) + fabsf(z * t.m[10]));
    return xf_vector(t, v);
}
PT_DEV V3 xf_point_abs_err(const M4 &t, V3 p, V3 perr, V3 &abs_err) {  // transform.rs:461-494
    const float x = p.x, y = p.y, z = p.z, g = gammaf(3);
    const float xp = x * t.m[0] + y * t.m[1] + z * t.m[2] + t.m[3];
    const float yp = x * t.m[4] + y * t.m[5] + z * t.m[6] + t.m[7];
    const float zp = x * t.m[8] + y * t.m[9] + z * t.m[10] + t.m[11];
    const float wp = x * t.m[12] + y * t.m[13] + z * t.m[14] + t.m[15];
    abs_err.x = (g + 1.0f) * (fabsf(t.m[0]) * perr.x + fabsf(t.m[1]) * perr.y + fabsf(t.m[2]) * perr.z) +
                g * (fabsf(t.m[0] * x) + fabsf(t.m[1] * y) + fabsf(t.m[2] * z) + fabsf(t.m[3]));
    abs_err.y = (g + 1.0f) * (fabsf(t.m[4]) * perr.x + fabsf(t.m[5]) * perr.y + fabsf(t.m[6]) * perr.z) +
                g * (fabsf(t.m[4] * x) + fabsf(t.m[5] * y) + fabsf(t.m[6] * z) + fabsf(t.m[7]));
    abs_err.z = (g + 1.0f) * (fabsf(t.m[8]) * perr.x + fabsf(t.m[9]) * perr.y + fabsf(t.m[10]) * perr.z) +
                g * (fabsf(t.m[8] * x) + fabsf(t.m[9] * y) + fabsf(t.m[10] * z) + fabsf(t.m[11]));
    if (wp == 1.0f) return V3(xp, yp, zp);
    return V3(xp, yp, zp) / wp;
}
PT_DEV V3 xf_normal_inv(const M4 &minv, V3 n) {  // transform.rs:529-541 (caller passes the inverse matrix)
    const float x = n.x, y = n.y, z = n.z;
    return V3(x * minv.m[0] + y * minv.m[4] + z * minv.m[8], x * minv.m[1] + y * minv.m[5] + z * minv.m[9], x * minv.m[2] + y * minv.m[6] + z * minv.m[10]);
}
PT_DEV M4 ldm4g(const float *p) { M4 m; for (int i = 0; i < 16; ++i) m.m[i] = p[i]; return m; }

// Shared part of intersect / intersect_p: object-space hit (sphere.rs:59-152 == :198-286 incl. the `phi += 2*phi` quirks).
PT_DEV bool sphere_hit(const PtSphere &S, V3 r_o, V3 r_d, float r_tmax, bool is_intersect_p, float &t_out, V3 &p_hit_out, float &phi_out, V3 &d_obj) {
    const M4 w2o = ldm4g(S.world_to_object);
    V3 oerr, derr;
    V3 o = xf_point_err(w2o, r_o, oerr);
    const V3 d = xf_vector_err(w2o, r_d, derr);
    const float l2 = length_squared(d);
    if (l2 > 0.0f) { const float dt = dot(vabs(d), oerr) / l2; o = o + d * dt; }
    d_obj = d;
    if (S.kind == PT_QUADRIC_DISK) {   // Disk::intersect / intersect_p (disk.rs:55-118)
        if (d.z == 0.0f) return false;
        const float t = (S.z_min - o.z) / (is_intersect_p ? d.z : r_d.z);   // intersect divides by the WORLD ray's d.z, as written there (disk.rs:66)
        if (t <= 0.0f || t >= r_tmax) return false;
        V3 ph = o + d * t;
        const float dist2 = ph.x * ph.x + ph.y * ph.y;
        if (dist2 > S.radius * S.radius || dist2 < S.inner_radius * S.inner_radius) return false;
        float phi = dm_atan2f(ph.y, ph.x);
        if (phi < 0.0f) phi += 2.0f * kPi;
        if (phi > S.phi_max) return false;
        t_out = t; p_hit_out = ph; phi_out = phi;
        return true;
    }
    const EFloat ox(o.x, oerr.x), oy(o.y, oerr.y), oz(o.z, oerr.z), dx(d.x, derr.x), dy(d.y, derr.y), dz(d.z, derr.z);
    const EFloat a = dx * dx + dy * dy + dz * dz;
    const EFloat b = EFloat(2.0f) * (dx * ox + dy * oy + dz * oz);
    const EFloat c = ox * ox + oy * oy + oz * oz - EFloat(S.radius) * EFloat(S.radius);
    EFloat t0, t1;
    if (!ef_quadratic(a, b, c, t0, t1)) return false;
    if (t0.high > r_tmax || t1.low <= 0.0f) return false;
    EFloat ts = t0;
    if (ts.low <= 0.0f) { ts = t1; if (ts.high > r_tmax) return false; }
    V3 ph = o + d * ts.v;
    { const float sc = S.radius / length(ph); ph = V3(ph.x * sc, ph.y * sc, ph.z * sc); }
    if (ph.x == 0.0f && ph.y == 0.0f) ph.x = 1e-5f * S.radius;
    float phi = dm_atan2f(ph.y, ph.x);
    if (phi < 0.0f) phi += is_intersect_p ? 2.0f * phi : 2.0f * kPi;
    if ((S.z_min > -S.radius && ph.z < S.z_min) || (S.z_max < S.radius && ph.z > S.z_max) || phi > S.phi_max) {
        if (ts.v == t1.v) return false;
        if (t1.high > r_tmax) return false;
        ts = t1;
        ph = o + d * ts.v;
        { const float sc = S.radius / length(ph); ph = V3(ph.x * sc, ph.y * sc, ph.z * sc); }
        if (ph.x == 0.0f && ph.y == 0.0f) ph.x = 1e-5f * S.radius;
        phi = dm_atan2f(ph.y, ph.x);
        if (phi < 0.0f) phi += 2.0f * phi;
        if ((S.z_min > -S.radius && ph.z < S.z_min) || (S.z_max < S.radius && ph.z > S.z_max) || phi > S.phi_max) return false;
    }
    t_out = ts.v; p_hit_out = ph; phi_out = phi;
    return true;
}

// Sphere::intersect's SurfaceInteraction (sphere.rs:148-192) for a ray known to hit (re-evaluated at shade time with
// t_max = inf: the accepted root does not depend on t_max once the hit was accepted).
// `with_shape`: the interaction knows its shape (GeometricPrimitive::intersect passes it, Shape::pdf_wi does not): only then
// SurfaceInteraction::new flips the normal for reverse_orientation ^ transform_swaps_handedness (interaction.rs:194-199).
PT_DEV bool sphere_fill_interaction(const PtSphere &S, V3 r_o, V3 r_d, SurfaceInteraction &si, bool with_shape = true) {
    float t, phi; V3 p_hit, d_obj;
    if (!sphere_hit(S, r_o, r_d, PT_INF, false, t, p_hit, phi, d_obj)) return false;
    if (S.kind == PT_QUADRIC_DISK) {   // disk.rs:78-96
        const float dist2 = p_hit.x * p_hit.x + p_hit.y * p_hit.y;
        const float r_hit = sqrtf(dist2);
        si.uv = P2(phi / S.phi_max, (S.radius - r_hit) / (S.radius - S.inner_radius));
        const V3 dpdu(-S.phi_max * p_hit.y, S.phi_max * p_hit.x, 0.0f);
        const V3 dpdv = V3(p_hit.x, p_hit.y, 0.0f) * (S.inner_radius - S.radius) / r_hit;
        p_hit.z = S.z_min;
        const bool flip = with_shape && ((S.reverse_orientation != 0) != (S.transform_swaps_handedness != 0));
        V3 n = normalize(cross(dpdu, dpdv));
        if (flip) n = -n;
        const V3 wo = normalize(-d_obj);
        const M4 o2w = ldm4g(S.object_to_world), w2o = ldm4g(S.world_to_object);
        si.p = xf_point_abs_err(o2w, p_hit, V3(0.0f, 0.0f, 0.0f), si.p_error);
        si.n = normalize(xf_normal_inv(w2o, n));
        si.wo = normalize(xf_vector(o2w, wo));
        si.dpdu = xf_vector(o2w, dpdu); si.dpdv = xf_vector(o2w, dpdv);
        si.sh_n = face_forward(normalize(xf_normal_inv(w2o, n)), si.n);
        si.sh_dpdu = si.dpdu; si.sh_dpdv = si.dpdv;
        si.sh_dndu = V3(0.0f, 0.0f, 0.0f); si.sh_dndv = V3(0.0f, 0.0f, 0.0f);
        si.has_shape = with_shape; si.shape_flip = flip;
        return true;
    }
    const float theta = dm_acosf(clampf(p_hit.z / S.radius, -1.0f, 1.0f));
    si.uv = P2(phi / S.phi_max, (theta - S.theta_min) / (S.theta_max - S.theta_min));   // sphere.rs:148-152
    const float zradius = sqrtf(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
    const float inv_radius = 1.0f / zradius;
    const float cos_phi = p_hit.x * inv_radius, sin_phi = p_hit.y * inv_radius;
    const V3 dpdu(-S.phi_max * p_hit.y, S.phi_max * p_hit.x, 0.0f);
    const V3 dpdv = V3(p_hit.z * cos_phi, p_hit.z * sin_phi, -S.radius * dm_sinf(theta)) * (S.theta_max - S.theta_min);
    // dndu / dndv from the fundamental forms (sphere.rs:165-184); read by bump mapping only
    const V3 d2pduu = V3(p_hit.x, p_hit.y, 0.0f) * -S.phi_max * S.phi_max;
    const V3 d2pduv = V3(-sin_phi, cos_phi, 0.0f) * (S.theta_max - S.theta_min) * p_hit.z * S.phi_max;
    const V3 d2pdvv = V3(p_hit.x, p_hit.y, p_hit.z) * -(S.theta_max - S.theta_min) * (S.theta_max - S.theta_min);
    const float E = dot(dpdu, dpdu), Fm = dot(dpdu, dpdv), G = dot(dpdv, dpdv);
    const V3 Nn = normalize(cross(dpdu, dpdv));
    const float e = dot(Nn, d2pduu), f = dot(Nn, d2pduv), g = dot(Nn, d2pdvv);
    const float inv_EGF2 = 1.0f / (E * G - Fm * Fm);
    const V3 dndu = dpdu * (f * Fm - e * G) * inv_EGF2 + dpdv * (e * Fm - f * E) * inv_EGF2;
    const V3 dndv = dpdu * (g * Fm - f * G) * inv_EGF2 + dpdv * (f * Fm - g * E) * inv_EGF2;
    const V3 p_error = vabs(p_hit) * gammaf(5);
    const V3 n = normalize(cross(dpdu, dpdv));
    const V3 wo = normalize(-d_obj);
    const M4 o2w = ldm4g(S.object_to_world), w2o = ldm4g(S.world_to_object);
    si.p = xf_point_abs_err(o2w, p_hit, p_error, si.p_error);
    si.n = normalize(xf_normal_inv(w2o, n));
    si.wo = normalize(xf_vector(o2w, wo));
    si.dpdu = xf_vector(o2w, dpdu); si.dpdv = xf_vector(o2w, dpdv);
    si.sh_n = face_forward(normalize(xf_normal_inv(w2o, n)), si.n);
    si.sh_dpdu = si.dpdu; si.sh_dpdv = si.dpdv;
    si.sh_dndu = xf_normal_inv(w2o, dndu); si.sh_dndv = xf_normal_inv(w2o, dndv);
    si.has_shape = false; si.shape_flip = false;
    return true;
}

}  // namespace ptd
