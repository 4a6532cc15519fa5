// dev_light.h -- light sampling/evaluation and light-selection distributions on device.
//   lights/diffuse.rs:71-112; core/shape.rs:40-82; shapes/triangle.rs:556-584; lights/distant.rs:64-84;
//   lights/point.rs:52-69; lights/infinite.rs:118-177; core/mipmap.rs:202-223,295-327;
//   core/sampling.rs:38-85,94-145; core/lightdistrib.rs:112-340; core/pbrt.rs:184-204.
#pragma once
#include "dev_scene.h"
#include "dev_sampler.h"
#include "dev_sphere.h"

namespace ptd {

// core/pbrt.rs:184-204 over a cdf array: pred(i) = cdf[i] <= u
PT_DEV int find_interval_cdf(const float *cdf, int size, float u) {
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int r = first - 1;
    return r < 0 ? 0 : (r > size - 2 ? size - 2 : r);
}

// A Distribution1D stored as func[n], cdf[n+1], func_int (sampling.rs:6-34 built on device or host).
struct Dist1D { const float *func; const float *cdf; float func_int; int n; };

PT_DEV int dist_sample_discrete(const Dist1D &d, float u, float &pdf) {  // sampling.rs:66-85
    if (d.n >= 1 && d.n <= 3) {   // one to three lights (an empty distribution takes the general path, which reads nothing): the whole cdf, func and func_int in one round trip to memory; the SAME bisection, over registers
        const int n = d.n;
        const float c0 = d.cdf[0], c1 = d.cdf[1], c2 = d.cdf[n < 2 ? n : 2], c3 = d.cdf[n < 3 ? n : 3];
        const float f0 = d.func[0], f1 = d.func[n < 2 ? 0 : 1], f2 = d.func[n < 3 ? 0 : 2];
        int first = 0, len = n + 1;
        while (len > 0) {
            const int half = len >> 1, middle = first + half;
            const float cm = middle == 0 ? c0 : middle == 1 ? c1 : middle == 2 ? c2 : c3;
            if (cm <= u) { first = middle + 1; len -= half + 1; }
            else len = half;
        }
        const int r = first - 1;
        const int off = r < 0 ? 0 : (r > n - 1 ? n - 1 : r);
        const float fo = off == 0 ? f0 : off == 1 ? f1 : f2;
        pdf = (d.func_int > 0.0f) ? fo / (d.func_int * (float)n) : 0.0f;
        return off;
    }
    int off = find_interval_cdf(d.cdf, d.n + 1, u);
    pdf = (d.func_int > 0.0f) ? d.func[off] / (d.func_int * (float)d.n) : 0.0f;
    return off;
}
PT_DEV float dist_sample_continuous(const Dist1D &d, float u, float &pdf, int &off) {  // sampling.rs:38-64
    off = find_interval_cdf(d.cdf, d.n + 1, u);
    float du = u - d.cdf[off];
    float diff = d.cdf[off + 1] - d.cdf[off];
    if (diff > 0.0f) du /= diff;
    pdf = (d.func_int > 0.0f) ? d.func[off] / d.func_int : 0.0f;
    return ((float)off + du) / (float)d.n;
}

struct IData { V3 p, p_error, n; };

PT_DEV void spawn_ray(const IData &it, V3 d, V3 &o) { o = offset_ray_origin(it.p, it.p_error, it.n, d); }  // interaction.rs:32-36
PT_DEV void spawn_ray_to(const IData &a, const IData &b, V3 &o, V3 &d) {  // interaction.rs:45-52
    o = offset_ray_origin(a.p, a.p_error, a.n, b.p - a.p);
    V3 t = offset_ray_origin(b.p, b.p_error, b.n, o - b.p);
    d = t - o;
}

// MIPMap::lookup(st, 0.0) == triangle(level 0) with Repeat wrap (mipmap.rs:202-223,295-327)
PT_DEV RGB env_lookup(const DeviceScene &s, P2 st) {
    int w = (int)s.env_w, h = (int)s.env_h;
    float sf = st.x * (float)w - 0.5f, tf = st.y * (float)h - 0.5f;
    int64_t s0 = f2i_sat(floorf(sf)), t0 = f2i_sat(floorf(tf));
    float ds = sf - (float)s0, dt = tf - (float)t0;
    // mod_(s, u) (mipmap.rs:303): st lies in [0,1] for every caller, so s0 is in [-1, w]; wrap with one compare each side
    // (a 64-bit `%` costs > 100 instructions on this ISA); anything outside [-w, 2w) takes the general path.
    auto wrap = [](int64_t v, int n) -> int {
        if (v >= -(int64_t)n && v < 2 * (int64_t)n) { int r = (int)v; if (r < 0) r += n; else if (r >= n) r -= n; return r; }
        int64_t r = v % n; if (r < 0) r += n;
        return (int)r;
    };
    const int sa = wrap(s0, w), sb = wrap(s0 + 1, w), ta = wrap(t0, h), tb = wrap(t0 + 1, h);
    auto texel = [&](int si, int ti) -> RGB {
        const float *p = s.env_texels + 3 * ((size_t)ti * w + si);
        return RGB(p[0], p[1], p[2]);
    };
    RGB tmp1 = texel(sb, tb) * (ds * dt);
    RGB tmp2 = texel(sb, ta) * (ds * (1.0f - dt));
    RGB tmp3 = texel(sa, tb) * ((1.0f - ds) * dt);
    RGB tmp4 = texel(sa, ta) * ((1.0f - ds) * (1.0f - dt));
    return tmp4 + tmp3 + tmp2 + tmp1;
}

PT_DEV bool light_is_delta(const PtLight &L) { return L.type == PT_LIGHT_DISTANT || L.type == PT_LIGHT_POINT || L.type == PT_LIGHT_SPOT; }

PT_DEV RGB area_l(const PtLight &L, V3 n, V3 w) {  // diffuse.rs:71-79
    if (L.two_sided || dot(n, w) > 0.0f) return RGB(L.L[0], L.L[1], L.L[2]);
    return RGB(0.0f);
}
// A triangle area light from DeviceScene::light_rec
struct LightTri { V3 p0, p1, p2, n_sample, n_pdf; uint32_t fl, tri; float area, inv_area; RGB Lemit; bool two_sided, degenerate; };
PT_DEV bool load_light_tri(const DeviceScene &s, uint32_t li, LightTri &t, bool for_pdf) {
    const float4 *r = s.light_rec + 6 * (size_t)li;
    const float4 q0 = r[0], q1 = r[1], q2 = r[2];
    t.p0 = V3(q0.x, q0.y, q0.z); t.p1 = V3(q1.x, q1.y, q1.z); t.p2 = V3(q2.x, q2.y, q2.z);
    const uint32_t bits = __float_as_uint(q0.w);
    t.fl = bits & 0xffu; t.degenerate = (bits & 0x200u) != 0u; t.tri = __float_as_uint(q1.w); t.area = q2.w;
    if (for_pdf) { const float4 q5 = r[5]; t.n_pdf = V3(q5.x, q5.y, q5.z); }
    else {
        const float4 q3 = r[3], q4 = r[4];
        t.Lemit = RGB(q3.x, q3.y, q3.z); t.two_sided = __float_as_uint(q3.w) != 0u;
        t.n_sample = V3(q4.x, q4.y, q4.z); t.inv_area = q4.w;
    }
    return (bits & 0x100u) != 0u;
}
PT_DEV RGB area_l(const LightTri &t, V3 n, V3 w) { return (t.two_sided || dot(n, w) > 0.0f) ? t.Lemit : RGB(0.0f); }   // diffuse.rs:71-79
PT_DEV M4 ldm4(const float *p) { M4 m; for (int i = 0; i < 16; ++i) m.m[i] = p[i]; return m; }

PT_DEV RGB light_le(const DeviceScene &s, const PtLight &L, V3 ray_d) {  // infinite.rs:118-126 (others: 0)
    if (L.type != PT_LIGHT_INFINITE) return RGB(0.0f);
    V3 w = normalize(xf_vector(ldm4(L.world_to_light), ray_d));
    P2 st(spherical_phi(w) * kInv2Pi, spherical_theta(w) * kInvPi);
    return env_lookup(s, st);
}

// Distribution2D::sample_continuous / pdf over the env importance image (sampling.rs:119-145)
PT_DEV P2 env_sample_continuous(const DeviceScene &s, P2 u, float &pdf) {
    int nu = 2 * (int)s.env_w, nv = 2 * (int)s.env_h;
    Dist1D marg{s.env_marg_func, s.env_marg_cdf, s.env_marg_int, nv};
    float pdf1, pdf0; int v, dummy;
    float d1 = dist_sample_continuous(marg, u.y, pdf1, v);
    Dist1D cond{s.env_func + (size_t)v * nu, s.env_cdf + (size_t)v * (nu + 1), s.env_func_int[v], nu};
    float d0 = dist_sample_continuous(cond, u.x, pdf0, dummy);
    pdf = pdf0 * pdf1;
    return P2(d0, d1);
}
PT_DEV float env_pdf(const DeviceScene &s, P2 p) {
    int nu = 2 * (int)s.env_w, nv = 2 * (int)s.env_h;
    uint32_t iu = f2u32_sat(p.x * (float)nu); if (iu > (uint32_t)(nu - 1)) iu = nu - 1;
    uint32_t iv = f2u32_sat(p.y * (float)nv); if (iv > (uint32_t)(nv - 1)) iv = nv - 1;
    return s.env_func[(size_t)iv * nu + iu] / s.env_marg_int;
}

// Sphere::sample_interaction (sphere.rs:313-378) incl. Sphere::sample (:295-311). The cone branch leaves it.n = 0
// (SURVEY App. A #7): one-sided sphere lights return L = 0 from sample_li; the far end of the shadow ray is not offset.
PT_DEV IData sphere_sample_interaction(const PtSphere &S, const IData &ref, P2 u, float &pdf) {
    const M4 o2w = ldm4g(S.object_to_world), w2o = ldm4g(S.world_to_object);
    const V3 pcenter = xf_point(o2w, V3(0.0f, 0.0f, 0.0f));
    const V3 porigin = offset_ray_origin(ref.p, ref.p_error, ref.n, pcenter - ref.p);
    IData it;
    if (distance_squared(porigin, pcenter) <= S.radius * S.radius) {
        const float z = 1.0f - 2.0f * u.x;                           // uniform_sample_sphere, sampling.rs:212-218
        const float r = sqrtf(maxf(1.0f - z * z, 0.0f));
        const float phi = 2.0f * kPi * u.y;
        float sn, cs; dm_sincosf(phi, sn, cs);
        V3 pobj = V3(0.0f, 0.0f, 0.0f) + V3(r * cs, r * sn, z) * S.radius;
        it.n = normalize(xf_normal_inv(w2o, pobj));
        if (S.reverse_orientation) it.n = it.n * -1.0f;
        pobj = pobj * (S.radius / length(pobj));
        const V3 perr = vabs(pobj) * gammaf(5);
        it.p = xf_point_abs_err(o2w, pobj, perr, it.p_error);
        pdf = 1.0f / (S.phi_max * S.radius * (S.z_max - S.z_min));
        V3 wi = it.p - ref.p;
        if (length_squared(wi) == 0.0f) pdf = 0.0f;
        else { wi = normalize(wi); pdf *= distance_squared(ref.p, it.p) / abs_dot(it.n, -wi); }
        if (__builtin_isinf(pdf)) pdf = 0.0f;
        return it;
    }
    const float dc = length(ref.p - pcenter);
    const float invdc = 1.0f / dc;
    const V3 wc = (pcenter - ref.p) * invdc; V3 wcx, wcy;
    coordinate_system(wc, wcx, wcy);
    const float sin_thetamax = S.radius * invdc;
    const float sin_thetamax2 = sin_thetamax * sin_thetamax;
    const float inv_sin_thetamax = 1.0f / sin_thetamax;
    const float cos_thetamax = sqrtf(maxf(1.0f - sin_thetamax2, 0.0f));
    float cos_theta = (cos_thetamax - 1.0f) * u.x + 1.0f;
    float sin_theta2 = 1.0f - cos_theta * cos_theta;
    if (sin_thetamax2 < 0.00068523f) { sin_theta2 = sin_thetamax2 * u.x; cos_theta = sqrtf(1.0f - sin_theta2); }
    const float cos_alpha = sin_theta2 * inv_sin_thetamax + cos_theta * sqrtf(maxf(1.0f - sin_theta2 * inv_sin_thetamax * inv_sin_thetamax, 0.0f));
    const float sin_alpha = sqrtf(maxf(1.0f - cos_alpha * cos_alpha, 0.0f));
    const float phi = u.y * 2.0f * kPi;
    float sn, cs; dm_sincosf(phi, sn, cs);
    const V3 nworld = (-wcx) * sin_alpha * cs + (-wcy) * sin_alpha * sn + (-wc) * cos_alpha;   // spherical_direction_basis
    const V3 pworld = pcenter + V3(nworld.x, nworld.y, nworld.z) * S.radius;
    it.p = pworld; it.p_error = vabs(pworld) * gammaf(5); it.n = V3(0.0f, 0.0f, 0.0f);
    pdf = 1.0f / (2.0f * kPi * (1.0f - cos_thetamax));
    return it;
}

// Light::sample_li. Returns Li; fills wi, pdf and the far end of the visibility segment.
template <bool SPH> PT_DEV RGB light_sample_li(const DeviceScene &s, uint32_t li, const IData &ref, P2 u, V3 &wi, float &pdf, IData &p1) {
    p1.p = V3(); p1.p_error = V3(); p1.n = V3();
    {   // a triangle area light (diffuse.rs:95-112 + shape.rs:40-58 + triangle.rs:556-584), everything from its record
        LightTri t;
        if (load_light_tri(s, li, t, false)) {
            const float su0 = sqrtf(u.x);
            const float b0 = 1.0f - su0, b1 = u.y * su0;  // uniform_sample_triangle, sampling.rs:250-254
            IData it;
            const float b2 = 1.0f - b0 - b1;
            it.p = t.p0 * b0 + t.p1 * b1 + t.p2 * b2;
            it.n = t.n_sample;   // normalize(cross(p1 - p0, p2 - p0)), flipped by the orientation flags when the mesh has no normals
            if (t.fl & PT_TRI_HAS_N) {
                const uint32_t i0 = s.indices[3 * t.tri], i1 = s.indices[3 * t.tri + 1], i2 = s.indices[3 * t.tri + 2];
                const V3 ns = ld3(s.N, i0) * b0 + ld3(s.N, i1) * b1 + ld3(s.N, i2) * b2;
                it.n = face_forward(it.n, ns);
            }
            const V3 pabs = vabs(t.p0 * b0) + vabs(t.p1 * b1) + vabs(t.p2 * b2);
            it.p_error = pabs * gammaf(6);
            pdf = t.inv_area;   // 1 / area
            V3 w = it.p - ref.p;
            if (length_squared(w) == 0.0f) pdf = 0.0f;
            else {
                w = normalize(w);
                pdf *= distance_squared(ref.p, it.p) / abs_dot(it.n, -w);
                if (__builtin_isinf(pdf)) pdf = 0.0f;
            }
            if (pdf == 0.0f || length_squared(it.p - ref.p) == 0.0f) { pdf = 0.0f; return RGB(0.0f); }
            wi = normalize(it.p - ref.p);
            p1 = it;
            return area_l(t, it.n, -wi);
        }
    }
    const PtLight &L = s.lights[li];
    switch (L.type) {
    case PT_LIGHT_DIFFUSE_AREA: {  // diffuse.rs:95-112 + shape.rs:40-58 + triangle.rs:556-584
        if (SPH && (s.prim_shape[L.prim] >> 30) == PT_SHAPE_SPHERE && s.spheres[s.prim_shape[L.prim] & 0x3fffffffu].kind == PT_QUADRIC_DISK) {
            // Disk::sample (disk.rs:124-139: the whole disk of `radius`, normal from (0, 0, 0.1)) + Shape::sample_interaction (shape.rs:40-52)
            const PtSphere &S = s.spheres[s.prim_shape[L.prim] & 0x3fffffffu];
            const P2 pd = concentric_sample_disk(u);
            const M4 o2w = ldm4g(S.object_to_world), w2o = ldm4g(S.world_to_object);
            IData it;
            it.n = normalize(xf_normal_inv(w2o, V3(0.0f, 0.0f, 0.1f)));
            if (S.reverse_orientation) it.n = it.n * -1.0f;
            it.p = xf_point_abs_err(o2w, V3(pd.x * S.radius, pd.y * S.radius, S.z_min), V3(0.0f, 0.0f, 0.0f), it.p_error);
            pdf = 1.0f / s.light_area[li];
            V3 w = it.p - ref.p;
            if (length_squared(w) == 0.0f) pdf = 0.0f;
            else {
                w = normalize(w);
                pdf *= distance_squared(ref.p, it.p) / abs_dot(it.n, -w);
                if (__builtin_isinf(pdf)) pdf = 0.0f;
            }
            if (pdf == 0.0f || length_squared(it.p - ref.p) == 0.0f) { pdf = 0.0f; return RGB(0.0f); }
            wi = normalize(it.p - ref.p);
            p1 = it;
            return area_l(L, it.n, -wi);
        }
        if (SPH && (s.prim_shape[L.prim] >> 30) == PT_SHAPE_SPHERE) {   // SPH == false: the scene holds no sphere
            IData it = sphere_sample_interaction(s.spheres[s.prim_shape[L.prim] & 0x3fffffffu], ref, u, pdf);
            if (pdf == 0.0f || length_squared(it.p - ref.p) == 0.0f) { pdf = 0.0f; return RGB(0.0f); }
            wi = normalize(it.p - ref.p);
            p1 = it;
            return area_l(L, it.n, -wi);
        }
        pdf = 0.0f; return RGB(0.0f);   // (triangle lights were served from their record above)
    }
    case PT_LIGHT_DISTANT: {  // distant.rs:64-84
        V3 wl(L.dir[0], L.dir[1], L.dir[2]);
        wi = wl; pdf = 1.0f;
        p1.p = ref.p + wl * (2.0f * s.world_radius);
        return RGB(L.L[0], L.L[1], L.L[2]);
    }
    case PT_LIGHT_POINT: {  // point.rs:52-69
        V3 pl(L.pos[0], L.pos[1], L.pos[2]);
        wi = normalize(pl - ref.p); pdf = 1.0f;
        p1.p = pl;
        return RGB(L.L[0], L.L[1], L.L[2]) / distance_squared(pl, ref.p);
    }
    case PT_LIGHT_SPOT: {  // spot.rs:48-56,71-87
        V3 pl(L.pos[0], L.pos[1], L.pos[2]);
        wi = normalize(pl - ref.p); pdf = 1.0f;
        p1.p = pl;
        V3 wl = normalize(xf_vector(ldm4(L.world_to_light), -wi));
        float cos_theta = wl.z, fall;
        if (cos_theta < L.cos_total_width) fall = 0.0f;
        else if (cos_theta >= L.cos_falloff_start) fall = 1.0f;
        else { float delta = (cos_theta - L.cos_total_width) / (L.cos_falloff_start - L.cos_total_width); fall = (delta * delta) * (delta * delta); }
        return RGB(L.L[0], L.L[1], L.L[2]) * fall / distance_squared(pl, ref.p);
    }
    case PT_LIGHT_INFINITE: {  // infinite.rs:140-177
        float map_pdf = 0.0f;
        P2 uv = env_sample_continuous(s, u, map_pdf);
        if (map_pdf == 0.0f) { pdf = 0.0f; return RGB(0.0f); }
        float theta = uv.y * kPi, phi = uv.x * 2.0f * kPi;
        float cos_t, sin_t, sin_p, cos_p;
        dm_sincosf(theta, sin_t, cos_t);
        dm_sincosf(phi, sin_p, cos_p);
        V3 v(sin_t * cos_p, sin_t * sin_p, cos_t);
        wi = xf_vector(ldm4(L.light_to_world), v);
        pdf = map_pdf / (2.0f * kPi * kPi * sin_t);
        if (sin_t == 0.0f) pdf = 0.0f;
        p1.p = ref.p + wi * (2.0f * s.world_radius);
        return env_lookup(s, uv);
    }
    default: pdf = 0.0f; return RGB(0.0f);
    }
}

// Light::pdf_li (area: Shape::pdf_wi re-intersects the light's own triangle, shape.rs:63-82)
template <bool SPH> PT_DEV float light_pdf_li(const DeviceScene &s, uint32_t li, const IData &ref, V3 wi) {
    {   // Shape::pdf_wi of a triangle light (shape.rs:63-82): intersect the light's own triangle, without a shape (no orientation from
        // shading normals: interaction.n is the geometric normal, flipped by the triangle's flags -- triangle.rs:266-392 with `s` = None)
        LightTri t;
        if (load_light_tri(s, li, t, true)) {
            V3 o; spawn_ray(ref, wi, o);
            float th, b0, b1, b2;
            if (!tri_hit_params(t.p0, t.p1, t.p2, o, wi, PT_INF, th, b0, b1, b2)) return 0.0f;
            if (t.degenerate) return 0.0f;   // tri_partials fails for this triangle (a property of its vertices and uvs)
            const V3 ip = t.p0 * b0 + t.p1 * b1 + t.p2 * b2;
            const V3 in = t.n_pdf;          // normalize(cross(p0 - p2, p1 - p2)), flipped by the orientation flags
            float pdf = distance_squared(ref.p, ip) / (dot(in, -wi) * t.area);
            if (__builtin_isinf(pdf)) pdf = 0.0f;
            return pdf;
        }
    }
    const PtLight &L = s.lights[li];
    if (SPH && L.type == PT_LIGHT_DIFFUSE_AREA && (s.prim_shape[L.prim] >> 30) == PT_SHAPE_SPHERE && s.spheres[s.prim_shape[L.prim] & 0x3fffffffu].kind == PT_QUADRIC_DISK) {
        // Shape::pdf_wi (shape.rs:59-73): intersect without a shape (no orientation flip), signed cosine
        const PtSphere &S = s.spheres[s.prim_shape[L.prim] & 0x3fffffffu];
        V3 o; spawn_ray(ref, wi, o);
        SurfaceInteraction il;
        if (!sphere_fill_interaction(S, o, wi, il, false)) return 0.0f;
        float pdf = distance_squared(ref.p, il.p) / (dot(il.n, -wi) * s.light_area[li]);
        if (__builtin_isinf(pdf)) pdf = 0.0f;
        return pdf;
    }
    if (SPH && L.type == PT_LIGHT_DIFFUSE_AREA && (s.prim_shape[L.prim] >> 30) == PT_SHAPE_SPHERE) {  // Sphere::pdf_wi (sphere.rs:380-395)
        const PtSphere &S = s.spheres[s.prim_shape[L.prim] & 0x3fffffffu];
        const V3 pcenter = xf_point(ldm4g(S.object_to_world), V3(0.0f, 0.0f, 0.0f));
        const V3 porigin = offset_ray_origin(ref.p, ref.p_error, ref.n, pcenter - ref.p);
        if (distance_squared(porigin, pcenter) <= S.radius * S.radius) {  // shape_pdfwi (shape.rs:117-136)
            V3 o; spawn_ray(ref, wi, o);
            SurfaceInteraction il;
            if (!sphere_fill_interaction(S, o, wi, il)) return 0.0f;
            float pdf = distance_squared(ref.p, il.p) / (dot(il.n, -wi) * s.light_area[li]);
            if (__builtin_isinf(pdf)) pdf = 0.0f;
            return pdf;
        }
        const float sin_thetamax2 = S.radius * S.radius / distance_squared(ref.p, pcenter);
        const float cos_thetamax = sqrtf(maxf(1.0f - sin_thetamax2, 0.0f));
        return 1.0f / (2.0f * kPi * (1.0f - cos_thetamax));
    }
    if (L.type == PT_LIGHT_DIFFUSE_AREA) return 0.0f;   // (triangle lights were served from their record above)
    if (L.type == PT_LIGHT_INFINITE) {  // infinite.rs:128-138
        V3 w = xf_vector(ldm4(L.world_to_light), wi);
        float theta = spherical_theta(w), phi = spherical_phi(w);
        float sin_t = dm_sinf(theta);
        if (sin_t == 0.0f) return 0.0f;
        return env_pdf(s, P2(phi * kInv2Pi, theta * kInvPi)) / (2.0f * kPi * kPi * sin_t);
    }
    return 0.0f;
}

// ---- light-selection distributions (lightdistrib.rs) -------------------------------------------------
struct LightGrid {
    int strategy;            // PT_LS_* after the `one light => uniform` rule (lightdistrib.rs:21)
    uint32_t nvox[3];
    uint32_t n_lights;
    const float *func;       // [ncell][n_lights]     (uniform/power: ncell == 1)
    const float *cdf;        // [ncell][n_lights+1]
    const float *func_int;   // [ncell]
    // Voxels filled on first touch (PT_LS_SPATIAL with many lights; include/mi355pt.h: PtLightStrategy): cell -> its block
    // {func_int, -, -, -, func[n_lights], cdf[n_lights + 1]}, or the scene's all-zero block while the voxel has not been computed. Every
    // vertex's voxel is computed before the vertex is shaded (k_light_touch), so a lookup that lands on the zero block is a bug: it is
    // counted in *missing and the render fails.
    const unsigned long long *cell_ptr;   // NULL: the dense arrays above
    unsigned long long zero_block;
    uint32_t *missing;
};
PT_DEV size_t light_grid_cell(const LightGrid &g, const DeviceScene &s, V3 p) {  // lightdistrib.rs:233-247
    float o[3] = {p.x - s.wb_min[0], p.y - s.wb_min[1], p.z - s.wb_min[2]};  // Bounds3::offset, bounds.rs:372-390
    uint32_t pi[3];
    for (int i = 0; i < 3; ++i) {
        if (s.wb_max[i] > s.wb_min[i]) o[i] /= s.wb_max[i] - s.wb_min[i];
        // clamp(`as i64`, 0, nvox - 1): the saturating cast (NaN -> 0) followed by the clamp, without the 64-bit conversion
        const float x = o[i] * (float)g.nvox[i];
        const uint32_t hi = g.nvox[i] - 1u;
        const uint32_t v = (x > 0.0f) ? (uint32_t)(int32_t)fminf(x, 2147483520.0f) : 0u;
        pi[i] = v > hi ? hi : v;
    }
    return ((size_t)pi[2] * g.nvox[1] + pi[1]) * g.nvox[0] + pi[0];
}
PT_DEV Dist1D light_distribution_lookup(const LightGrid &g, const DeviceScene &s, V3 p) {
    size_t cell = 0;
    if (g.strategy == PT_LS_SPATIAL) cell = light_grid_cell(g, s, p);
    if (g.cell_ptr) {
        const unsigned long long b = g.cell_ptr[cell];
        if (b == g.zero_block) atomicAdd(g.missing, 1u);
        const float *blk = reinterpret_cast<const float *>(b);
        Dist1D d{blk + 4, blk + 4 + g.n_lights, blk[0], (int)g.n_lights};
        return d;
    }
    Dist1D d{g.func + cell * g.n_lights, g.cdf + cell * (g.n_lights + 1), g.func_int[cell], (int)g.n_lights};
    return d;
}

}  // namespace ptd
