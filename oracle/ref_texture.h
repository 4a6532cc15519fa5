// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_math.h header).
// ref_texture.h: texture evaluation restated.
//   core/texture.rs:112-270      UVMapping2D / SphericalMapping2D / CylindricalMapping2D / PlannarMapping2D / IdentityMapping3D
//   textures/constant.rs, scaled.rs:30-33, mix.rs:29-35, biler.rs:27-36, uv.rs:21-33, checkerboard.rs:27-100
//   textures/imagemap.rs:167-176 (evaluate -> MIPMap::lookup2)
//   core/mipmap.rs:202-258 (lookup, lookup2), :296-327 (texel, triangle), :293-367 (ewa)
//   core/interaction.rs:269-342  compute_differentials ; core/transform.rs:174-186 solve_linearsystem_2x2
// log2 goes through the shared deterministic ln (dm_logf's double core) like every other transcendental.
#pragma once
#include "ref_scene.h"
#include "../include/pt_noise_perm.h"

namespace ref {

struct TexCtx {   // what Texture::evaluate reads from the SurfaceInteraction
    V3 p; P2 uv;
    V3 dpdx, dpdy;
    Float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0;
};

inline Float dm_log2f(Float x) { return (Float)((double)dm_logf(x) * 1.4426950408889634); }   // f32::log2 through the shared ln

struct ImagePyramid {
    int width = 0, height = 0, n_levels = 0, channels = 3;
    std::vector<size_t> offset;   // float offset of each level
    std::vector<Float> texels;
    int ures(int l) const { return std::max(1, width >> l); }
    int vres(int l) const { return std::max(1, height >> l); }
};

inline RGB mip_texel(const ImagePyramid &im, int wrap, int level, int64_t s, int64_t t) {  // mipmap.rs:296-312
    const int u = im.ures(level), v = im.vres(level);
    if (wrap == PT_WRAP_REPEAT) { s %= u; if (s < 0) s += u; t %= v; if (t < 0) t += v; }
    else if (s < 0 || s >= u || t < 0 || t >= v) return RGB(0.0f);
    const Float *p = im.texels.data() + im.offset[level] + ((size_t)t * u + s) * im.channels;
    return im.channels == 1 ? RGB(p[0]) : RGB(p[0], p[1], p[2]);
}
inline RGB mip_triangle(const ImagePyramid &im, int wrap, int level, P2 st) {  // mipmap.rs:314-327
    level = std::min(std::max(level, 0), im.n_levels - 1);
    Float s = st.x * (Float)im.ures(level) - 0.5f, t = st.y * (Float)im.vres(level) - 0.5f;
    int64_t s0 = f2i_sat(std::floor(s)), t0 = f2i_sat(std::floor(t));
    Float ds = s - (Float)s0, dt = t - (Float)t0;
    RGB tmp1 = mip_texel(im, wrap, level, s0 + 1, t0 + 1) * (ds * dt);
    RGB tmp2 = mip_texel(im, wrap, level, s0 + 1, t0) * (ds * (1.0f - dt));
    RGB tmp3 = mip_texel(im, wrap, level, s0, t0 + 1) * ((1.0f - ds) * dt);
    RGB tmp4 = mip_texel(im, wrap, level, s0, t0) * ((1.0f - ds) * (1.0f - dt));
    return tmp4 + tmp3 + tmp2 + tmp1;
}
inline RGB rgb_lerp(Float t, RGB a, RGB b) { return a * (1.0f - t) + b * t; }   // pbrt.rs lerp: (1 - t) * a + t * b
inline RGB mip_lookup(const ImagePyramid &im, int wrap, P2 st, Float width) {  // mipmap.rs:202-223
    Float level = (Float)(im.n_levels - 1) + dm_log2f(fmax_(width, 1.0e-8f));
    if (level < 0.0f) return mip_triangle(im, wrap, 0, st);
    if (level >= (Float)(im.n_levels - 1)) return mip_texel(im, wrap, im.n_levels - 1, 0, 0);
    Float ilevel = std::floor(level);
    Float delta = level - ilevel;
    return rgb_lerp(delta, mip_triangle(im, wrap, (int)ilevel, st), mip_triangle(im, wrap, (int)ilevel + 1, st));
}
inline RGB mip_ewa(const ImagePyramid &im, int wrap, const Float *lut, int level, P2 st, P2 dst0, P2 dst1) {  // mipmap.rs:293-367
    if (level >= im.n_levels) return mip_texel(im, wrap, im.n_levels - 1, 0, 0);
    st.x = st.x * (Float)im.ures(level) - 0.5f;
    st.y = st.y * (Float)im.vres(level) - 0.5f;
    dst0.x *= (Float)im.ures(level); dst0.y *= (Float)im.vres(level);
    dst1.x *= (Float)im.ures(level); dst1.y *= (Float)im.vres(level);
    Float A = dst0.y * dst0.y + dst1.y * dst1.y + 1.0f;
    Float B = -2.0f * (dst0.x * dst0.y + dst1.x * dst1.y);
    Float C = dst0.x * dst0.x + dst1.x * dst1.x + 1.0f;
    Float invf = 1.0f / (A * C - B * B * 0.25f);
    A *= invf; B *= invf; C *= invf;
    Float det = -B * B + 4.0f * A * C;
    Float idet = 1.0f / det;
    Float usqrt = std::sqrt(det * C), vsqrt = std::sqrt(det * A);
    int64_t s0 = f2i_sat(std::ceil(st.x - 2.0f * idet * usqrt)), s1 = f2i_sat(std::floor(st.x + 2.0f * idet * usqrt));
    int64_t t0 = f2i_sat(std::ceil(st.y - 2.0f * idet * vsqrt)), t1 = f2i_sat(std::floor(st.y + 2.0f * idet * vsqrt));
    RGB sum(0.0f); Float sum_wts = 0.0f;
    for (int64_t it = t0; it <= t1; ++it) {
        Float tt = (Float)it - st.y;
        for (int64_t is = s0; is <= s1; ++is) {
            Float ss = (Float)is - st.x;
            Float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
            if (r2 < 1.0f) {
                int index = (int)std::min<uint64_t>(f2u_sat(r2 * 128.0f), 127);
                Float weight = lut[index];
                sum += mip_texel(im, wrap, level, is, it) * weight;
                sum_wts += weight;
            }
        }
    }
    return sum / sum_wts;
}
inline RGB mip_lookup2(const ImagePyramid &im, const PtTexture &T, const Float *lut, P2 st, P2 dst0, P2 dst1) {  // mipmap.rs:225-258
    if (T.trilinear) {
        Float x = fmax_(std::fabs(dst0.x), std::fabs(dst0.y)), y = fmax_(std::fabs(dst1.x), std::fabs(dst1.y));
        return mip_lookup(im, (int)T.wrap, st, fmax_(x, y));
    }
    if (dst0.x * dst0.x + dst0.y * dst0.y < dst1.x * dst1.x + dst1.y * dst1.y) std::swap(dst0, dst1);
    Float majorl = std::sqrt(dst0.x * dst0.x + dst0.y * dst0.y);
    Float minorl = std::sqrt(dst1.x * dst1.x + dst1.y * dst1.y);
    if (minorl * T.max_anisotropy < majorl && minorl > 0.0f) {
        Float scale = majorl / (minorl * T.max_anisotropy);
        dst1.x *= scale; dst1.y *= scale;
        minorl *= scale;
    }
    if (minorl == 0.0f) return mip_triangle(im, (int)T.wrap, 0, st);
    Float lod = fmax_((Float)im.n_levels - 1.0f + dm_log2f(minorl), 0.0f);
    Float flo = std::floor(lod);
    int ilod = (int)f2u_sat(flo);
    return rgb_lerp(lod - (Float)ilod, mip_ewa(im, (int)T.wrap, lut, ilod, st, dst0, dst1), mip_ewa(im, (int)T.wrap, lut, ilod + 1, st, dst0, dst1));
}

// ---- Perlin noise (core/texture.rs:311-438). Quirks kept: `x.floor() as usize` saturates negative cells to 0 (so dx can be
// negative), turbulence adds `o + |noise|` per octave, fbm uses ln(x) * 1.442695 while turbulence uses f32::log2.
static const uint8_t kNoisePerm[512] = {PT_NOISE_PERM_VALUES};
inline Float noise_grad(uint64_t x, uint64_t y, uint64_t z, Float dx, Float dy, Float dz) {
    uint32_t h = kNoisePerm[kNoisePerm[kNoisePerm[x] + y] + z] & 15u;
    Float u = (h < 8 || h == 12 || h == 13) ? dx : dy;
    Float v = (h < 4 || h == 12 || h == 13) ? dy : dz;
    return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
inline Float noise_weight(Float t) { Float t3 = t * t * t, t4 = t3 * t; return 6.0f * t4 * t - 15.0f * t4 + 10.0f * t3; }
inline Float flerp(Float t, Float a, Float b) { return (1.0f - t) * a + t * b; }
inline Float noise3(Float x, Float y, Float z) {
    uint64_t ix = f2u_sat(std::floor(x)), iy = f2u_sat(std::floor(y)), iz = f2u_sat(std::floor(z));
    Float dx = x - (Float)ix, dy = y - (Float)iy, dz = z - (Float)iz;
    ix &= 255; iy &= 255; iz &= 255;
    Float w000 = noise_grad(ix, iy, iz, dx, dy, dz), w100 = noise_grad(ix + 1, iy, iz, dx - 1.0f, dy, dz);
    Float w010 = noise_grad(ix, iy + 1, iz, dx, dy - 1.0f, dz), w110 = noise_grad(ix + 1, iy + 1, iz, dx - 1.0f, dy - 1.0f, dz);
    Float w001 = noise_grad(ix, iy, iz + 1, dx, dy, dz - 1.0f), w101 = noise_grad(ix + 1, iy, iz + 1, dx - 1.0f, dy, dz - 1.0f);
    Float w011 = noise_grad(ix, iy + 1, iz + 1, dx, dy - 1.0f, dz - 1.0f), w111 = noise_grad(ix + 1, iy + 1, iz + 1, dx - 1.0f, dy - 1.0f, dz - 1.0f);
    Float wx = noise_weight(dx), wy = noise_weight(dy), wz = noise_weight(dz);
    Float x00 = flerp(wx, w000, w100), x10 = flerp(wx, w010, w110), x01 = flerp(wx, w001, w101), x11 = flerp(wx, w011, w111);
    Float y0 = flerp(wy, x00, x10), y1 = flerp(wy, x01, x11);
    return flerp(wz, y0, y1);
}
inline Float smooth_step(Float mn, Float mx, Float value) { Float v = clampv((value - mn) / (mx - mn), 0.0f, 1.0f); return v * v * (-2.0f * v + 3.0f); }
inline Float noise_fbm(V3 p, V3 dpdx, V3 dpdy, Float omega, uint32_t max_octaves) {
    Float len2 = fmax_(length_squared(dpdx), length_squared(dpdy));
    Float n = clampv(-1.0f - 0.5f * (dm_logf(len2) * 1.442695040888963387f), 0.0f, (Float)max_octaves);
    uint64_t nint = f2u_sat(std::floor(n));
    Float sum = 0.0f, lambda = 1.0f, o = 1.0f;
    for (uint64_t i = 0; i < nint; ++i) { V3 q = p * lambda; sum += o * noise3(q.x, q.y, q.z); lambda *= 1.99f; o *= omega; }
    Float npartial = n - (Float)nint;
    V3 q = p * lambda;
    sum += o * smooth_step(0.3f, 0.7f, npartial) * noise3(q.x, q.y, q.z);
    return sum;
}
inline Float noise_turbulence(V3 p, V3 dpdx, V3 dpdy, Float omega, uint32_t max_octaves) {
    Float len2 = fmax_(length_squared(dpdx), length_squared(dpdy));
    Float n = clampv(-1.0f - 0.5f * dm_log2f(len2), 0.0f, (Float)max_octaves);
    uint64_t nint = f2u_sat(std::floor(n));
    Float sum = 0.0f, lambda = 1.0f, o = 1.0f;
    for (uint64_t i = 0; i < nint; ++i) { V3 q = p * lambda; sum += o + std::fabs(noise3(q.x, q.y, q.z)); lambda *= 1.99f; o *= omega; }
    Float npartial = n - (Float)nint;
    V3 q = p * lambda;
    sum += o + flerp(smooth_step(0.3f, 0.7f, npartial), 0.2f, std::fabs(noise3(q.x, q.y, q.z)));
    for (uint64_t i = nint; i < max_octaves; ++i) { sum += o * 0.2f; o *= omega; }
    return sum;
}

struct TextureSet {
    std::vector<PtTexture> tex;
    std::vector<ImagePyramid> images;
    std::vector<Float> ewa_lut;

    P2 sphere_map(const M4 &w2t, V3 p) const {  // texture.rs:165-175
        V3 vec = normalize(xf_point(w2t, p) - V3(0.0f, 0.0f, 0.0f));
        Float theta = spherical_theta(vec), phi = spherical_phi(vec);
        return P2(theta * INV_PI, phi * INV2_PI);
    }
    P2 cylinder_map(const M4 &w2t, V3 p) const {  // texture.rs:213-220
        V3 vec = normalize(xf_point(w2t, p) - V3(0.0f, 0.0f, 0.0f));
        return P2(PI + dm_atan2f(vec.y, vec.x) * INV2_PI, vec.z);
    }
    P2 map2d(const PtTexture &T, const TexCtx &c, P2 &dstdx, P2 &dstdy) const {
        switch (T.mapping) {
        case PT_MAP_UV:
            dstdx = P2(T.su * c.dudx, T.sv * c.dvdx); dstdy = P2(T.su * c.dudy, T.sv * c.dvdy);
            return P2(T.su * c.uv.x + T.du, T.sv * c.uv.y + T.dv);
        case PT_MAP_PLANAR: {
            V3 vs(T.vs[0], T.vs[1], T.vs[2]), vt(T.vt[0], T.vt[1], T.vt[2]);
            dstdx = P2(dot(c.dpdx, vs), dot(c.dpdx, vt)); dstdy = P2(dot(c.dpdy, vs), dot(c.dpdy, vt));
            return P2(T.du + dot(c.p, vs), T.dv + dot(c.p, vt));
        }
        default: {  // spherical / cylindrical, texture.rs:178-201,223-244
            M4 w2t = m4_from(T.world_to_texture);
            auto f = [&](V3 p) { return T.mapping == PT_MAP_SPHERICAL ? sphere_map(w2t, p) : cylinder_map(w2t, p); };
            P2 st = f(c.p);
            const Float delta = 0.1f;
            P2 sx = f(c.p + c.dpdx * delta), sy = f(c.p + c.dpdy * delta);
            dstdx = P2((sx.x - st.x) / delta, (sx.y - st.y) / delta);
            dstdy = P2((sy.x - st.x) / delta, (sy.y - st.y) / delta);
            if (dstdx.y > 0.5f) dstdx.y = 1.0f - dstdx.y; else if (dstdx.y < -0.5f) dstdx.y = -(dstdx.y + 1.0f);
            if (dstdy.y > 0.5f) dstdy.y = 1.0f - dstdy.y; else if (dstdy.y < -0.5f) dstdy.y = -(dstdy.y + 1.0f);
            return st;
        }
        }
    }
    static bool even_sum(Float a, Float b) { return (f2i_sat(std::floor(a)) + f2i_sat(std::floor(b))) % 2 == 0; }
    RGB eval(int id, const TexCtx &c) const {
        const PtTexture &T = tex[id];
        switch (T.type) {
        case PT_TEX_CONSTANT: return RGB(T.value[0], T.value[1], T.value[2]);
        case PT_TEX_SCALE: return eval(T.child[0], c) * eval(T.child[1], c);
        case PT_TEX_MIX: {
            RGB t1 = eval(T.child[0], c), t2 = eval(T.child[1], c);
            Float amt = eval(T.child[2], c).c[0];
            return t1 * (1.0f - amt) + t2 * amt;
        }
        case PT_TEX_CHECKERBOARD2D: {
            P2 dstdx, dstdy;
            P2 st = map2d(T, c, dstdx, dstdy);
            if (!T.aa_closedform) return even_sum(st.x, st.y) ? eval(T.child[0], c) : eval(T.child[1], c);
            Float ds = fmax_(std::fabs(dstdx.x), std::fabs(dstdy.x)), dt = fmax_(std::fabs(dstdx.y), std::fabs(dstdy.y));
            Float s0 = st.x - ds, s1 = st.x + ds, t0 = st.y - dt, t1 = st.y + dt;
            if (std::floor(s0) == std::floor(s1) && std::floor(t0) == std::floor(t1))
                return even_sum(st.x, st.y) ? eval(T.child[0], c) : eval(T.child[1], c);
            auto bump = [](Float x) { return std::floor(x / 2.0f) + 2.0f * fmax_(x / 2.0f - std::floor(x / 2.0f) - 0.5f, 0.0f); };
            Float sint = (bump(s1) - bump(s0)) / (2.0f * ds), tint = (bump(t1) - bump(t0)) / (2.0f * dt);
            Float area2 = sint * tint - 2.0f * sint * tint;   // as written in checkerboard.rs:63
            if (ds > 1.0f || dt > 1.0f) area2 = 0.5f;
            return eval(T.child[0], c) * (1.0f - area2) + eval(T.child[1], c) * area2;
        }
        case PT_TEX_CHECKERBOARD3D: {
            V3 p = xf_point(m4_from(T.world_to_texture), c.p);
            bool even = (f2i_sat(std::floor(p.x)) + f2i_sat(std::floor(p.y)) + f2i_sat(std::floor(p.z))) % 2 == 0;
            return even ? eval(T.child[0], c) : eval(T.child[1], c);
        }
        case PT_TEX_IMAGEMAP: {
            P2 dstdx, dstdy;
            P2 st = map2d(T, c, dstdx, dstdy);
            return mip_lookup2(images[T.image], T, ewa_lut.data(), st, dstdx, dstdy);
        }
        case PT_TEX_UV: {
            P2 dstdx, dstdy;
            P2 st = map2d(T, c, dstdx, dstdy);
            return RGB(st.x - std::floor(st.x), st.y - std::floor(st.y), 0.0f);
        }
        case PT_TEX_FBM: case PT_TEX_WRINKLED: case PT_TEX_WINDY: case PT_TEX_MARBLE: {  // IdentityMapping3D (texture.rs:281-297)
            M4 w2t = m4_from(T.world_to_texture);
            V3 dpdx = xf_vector(w2t, c.dpdx), dpdy = xf_vector(w2t, c.dpdy), p = xf_point(w2t, c.p);
            if (T.type == PT_TEX_FBM) return RGB(noise_fbm(p, dpdx, dpdy, T.omega, T.octaves));
            if (T.type == PT_TEX_WRINKLED) return RGB(noise_turbulence(p, dpdx, dpdy, T.omega, T.octaves));
            if (T.type == PT_TEX_WINDY) {
                Float wstrength = noise_fbm(p * 0.1f, dpdx * 0.1f, dpdy * 0.1f, 0.5f, 3);
                Float wheight = noise_fbm(p, dpdx, dpdy, 0.5f, 6);
                return RGB(std::fabs(wstrength) * wheight);
            }
            static const Float Cm[9][3] = {{0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.5f, 0.5f, 0.5f}, {0.6f, 0.59f, 0.58f},
                                           {0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.2f, 0.2f, 0.33f}, {0.58f, 0.58f, 0.6f}};
            p = p * T.marble_scale;
            Float fb = noise_fbm(p, dpdx * T.marble_scale, dpdy * T.marble_scale, T.omega, T.octaves);
            Float marble = p.y + T.variation * fb;
            Float t = 0.5f + 0.5f * dm_sinf(marble);
            uint64_t first = std::min<uint64_t>(5, f2u_sat(std::floor(t * 6.0f)));
            RGB c0(Cm[first][0], Cm[first][1], Cm[first][2]), c1(Cm[first + 1][0], Cm[first + 1][1], Cm[first + 1][2]);
            RGB c2(Cm[first + 2][0], Cm[first + 2][1], Cm[first + 2][2]), c3(Cm[first + 3][0], Cm[first + 3][1], Cm[first + 3][2]);
            RGB s0 = c0 * (1.0f - t) + c1 * t, s1 = c1 * (1.0f - t) + c2 * t, s2 = c2 * (1.0f - t) + c3 * t;
            s0 = s0 * (1.0f - t) + s1 * t; s1 = s1 * (1.0f - t) + s2 * t;
            return (s0 * (1.0f - t) + s1 * t) * 1.5f;
        }
        case PT_TEX_DOTS: {
            P2 dstdx, dstdy;
            P2 st = map2d(T, c, dstdx, dstdy);
            uint64_t scell = f2u_sat(std::floor(st.x + 0.5f)), tcell = f2u_sat(std::floor(st.y + 0.5f));
            if (noise3((Float)scell + 0.5f, (Float)tcell + 0.5f, 0.5f) > 0.0f) {
                const Float radius = 0.35f, max_shift = 0.5f - radius;
                Float scenter = (Float)scell + max_shift * noise3((Float)scell + 1.5f, (Float)tcell + 2.8f, 0.5f);
                Float tcenter = (Float)tcell + max_shift * noise3((Float)scell + 4.5f, (Float)tcell + 9.8f, 0.5f);
                Float ds = st.x - scenter, dt = st.y - tcenter;
                if (ds * ds + dt * dt < radius * radius) return eval(T.child[1], c);
            }
            return eval(T.child[0], c);
        }
        case PT_TEX_BILERP: {
            P2 dstdx, dstdy;
            P2 st = map2d(T, c, dstdx, dstdy);
            RGB v00(T.v00[0], T.v00[1], T.v00[2]), v01(T.v01[0], T.v01[1], T.v01[2]), v10(T.v10[0], T.v10[1], T.v10[2]), v11(T.v11[0], T.v11[1], T.v11[2]);
            return v00 * (1.0f - st.y) * (1.0f - st.x) + v01 * (1.0f - st.x) * st.y + v10 * (1.0f - st.y) * st.x + v11 * st.y * st.x;
        }
        }
        return RGB(0.0f);
    }
};

// SurfaceInteraction::compute_differentials (interaction.rs:269-342). `has_diff` false (every ray after the camera ray):
// all differentials are zero.
struct RayDiff { bool has = false; V3 rx_o, rx_d, ry_o, ry_d; };
inline bool solve_2x2(const Float a[2][2], const Float b[2], Float &x0, Float &x1) {  // transform.rs:174-186
    Float det = a[0][0] * a[1][1] - a[0][1] * a[1][0];
    if (std::fabs(det) < 1.0e-10f) return false;
    x0 = (a[1][1] * b[0] - a[0][1] * b[1]) / det;
    x1 = (a[0][0] * b[1] - a[1][0] * b[0]) / det;
    return !(std::isnan(x0) || std::isnan(x1));
}
inline TexCtx compute_differentials(const SurfaceInteraction &si, const RayDiff &rd) {
    TexCtx c; c.p = si.p; c.uv = si.uv; c.dpdx = V3(0, 0, 0); c.dpdy = V3(0, 0, 0);
    if (!rd.has) return c;
    Float d = dot(si.n, si.p);
    Float tx = -(dot(si.n, rd.rx_o) - d) / dot(si.n, rd.rx_d);
    if (std::isinf(tx) || std::isnan(tx)) return c;
    V3 px = rd.rx_o + rd.rx_d * tx;
    Float ty = -(dot(si.n, rd.ry_o) - d) / dot(si.n, rd.ry_d);
    if (std::isinf(ty) || std::isnan(ty)) return c;
    V3 py = rd.ry_o + rd.ry_d * ty;
    c.dpdx = px - si.p; c.dpdy = py - si.p;
    int dim[2];
    if (std::fabs(si.n.x) > std::fabs(si.n.y) && std::fabs(si.n.x) > std::fabs(si.n.z)) { dim[0] = 1; dim[1] = 2; }
    else if (std::fabs(si.n.y) > std::fabs(si.n.z)) { dim[0] = 0; dim[1] = 2; }
    else { dim[0] = 0; dim[1] = 1; }
    const Float A[2][2] = {{si.dpdu[dim[0]], si.dpdv[dim[0]]}, {si.dpdu[dim[1]], si.dpdv[dim[1]]}};
    const Float Bx[2] = {px[dim[0]] - si.p[dim[0]], px[dim[1]] - si.p[dim[1]]};
    const Float By[2] = {py[dim[0]] - si.p[dim[0]], py[dim[1]] - si.p[dim[1]]};
    if (!solve_2x2(A, Bx, c.dudx, c.dvdx)) { c.dudx = 0.0f; c.dvdx = 0.0f; }
    if (!solve_2x2(A, By, c.dudy, c.dvdy)) { c.dudy = 0.0f; c.dvdy = 0.0f; }
    return c;
}

}  // namespace ref
