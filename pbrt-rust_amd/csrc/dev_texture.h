// dev_texture.h -- texture evaluation on the device.
//   core/texture.rs:112-270 (2-D / 3-D mappings), textures/{constant,scaled,mix,biler,uv,checkerboard,imagemap}.rs,
//   core/mipmap.rs:202-367 (lookup, lookup2, texel, triangle, ewa), core/interaction.rs:269-342 (compute_differentials),
//   core/transform.rs:174-186 (solve_linearsystem_2x2).
// The reference evaluates a texture tree recursively; here the host flattens every root into a postfix program
// (children before parent) that runs over a small value stack -- same arithmetic, no recursion. A checkerboard or mix
// evaluates all of its children and selects/blends afterwards (textures are pure functions, so the value is identical).
#pragma once
#include "dev_scene.h"
#include "../../include/pt_noise_perm.h"

namespace ptd {

constexpr int kTexStack = 6;     // value-stack depth of a flattened texture program (validated by the host)

struct TexCtx {                  // what Texture::evaluate reads from the SurfaceInteraction
    V3 p; P2 uv;
    V3 dpdx, dpdy;
    float dudx, dvdx, dudy, dvdy;
};
struct RayDiff { bool has; V3 rx_o, rx_d, ry_o, ry_d; };

PT_DEV float dm_log2f(float x) { return (float)((double)dm_logf(x) * 1.4426950408889634); }   // f32::log2 through the shared ln

PT_DEV int img_ures(const DevImage &im, int l) { int v = (int)im.width >> l; return v < 1 ? 1 : v; }
PT_DEV int img_vres(const DevImage &im, int l) { int v = (int)im.height >> l; return v < 1 ? 1 : v; }

PT_DEV RGB mip_texel(const DevImage &im, uint32_t wrap, int level, int64_t s, int64_t t) {  // mipmap.rs:296-312
    const int u = img_ures(im, level), v = img_vres(im, level);
    if (wrap == PT_WRAP_REPEAT) {   // u, v are powers of two: mod_ is a mask (two's complement)
        s &= (int64_t)(u - 1); t &= (int64_t)(v - 1);
    } else if (s < 0 || s >= u || t < 0 || t >= v) return RGB(0.0f);
    const float *p = im.texels + im.level_offset[level] + ((size_t)t * u + (size_t)s) * im.channels;
    return im.channels == 1 ? RGB(p[0]) : RGB(p[0], p[1], p[2]);
}
PT_DEV RGB mip_triangle(const DevImage &im, uint32_t wrap, int level, P2 st) {  // mipmap.rs:314-327
    level = level < 0 ? 0 : (level > (int)im.n_levels - 1 ? (int)im.n_levels - 1 : level);
    const float s = st.x * (float)img_ures(im, level) - 0.5f, t = st.y * (float)img_vres(im, level) - 0.5f;
    const int64_t s0 = f2i_sat(floorf(s)), t0 = f2i_sat(floorf(t));
    const float ds = s - (float)s0, dt = t - (float)t0;
    const RGB tmp1 = mip_texel(im, wrap, level, s0 + 1, t0 + 1) * (ds * dt);
    const RGB tmp2 = mip_texel(im, wrap, level, s0 + 1, t0) * (ds * (1.0f - dt));
    const RGB tmp3 = mip_texel(im, wrap, level, s0, t0 + 1) * ((1.0f - ds) * dt);
    const RGB tmp4 = mip_texel(im, wrap, level, s0, t0) * ((1.0f - ds) * (1.0f - dt));
    return tmp4 + tmp3 + tmp2 + tmp1;
}
PT_DEV RGB rgb_lerp(float t, RGB a, RGB b) { return a * (1.0f - t) + b * t; }
PT_DEV RGB mip_lookup(const DevImage &im, uint32_t wrap, P2 st, float width) {  // mipmap.rs:202-223
    const float level = (float)((int)im.n_levels - 1) + dm_log2f(maxf(width, 1.0e-8f));
    if (level < 0.0f) return mip_triangle(im, wrap, 0, st);
    if (level >= (float)((int)im.n_levels - 1)) return mip_texel(im, wrap, (int)im.n_levels - 1, 0, 0);
    const float ilevel = floorf(level);
    const float delta = level - ilevel;
    return rgb_lerp(delta, mip_triangle(im, wrap, (int)ilevel, st), mip_triangle(im, wrap, (int)ilevel + 1, st));
}
PT_DEV RGB mip_ewa(const DevImage &im, uint32_t wrap, const float *lut, int level, P2 st, P2 dst0, P2 dst1) {  // mipmap.rs:293-367
    if (level >= (int)im.n_levels) return mip_texel(im, wrap, (int)im.n_levels - 1, 0, 0);
    st.x = st.x * (float)img_ures(im, level) - 0.5f;
    st.y = st.y * (float)img_vres(im, level) - 0.5f;
    dst0.x *= (float)img_ures(im, level); dst0.y *= (float)img_vres(im, level);
    dst1.x *= (float)img_ures(im, level); dst1.y *= (float)img_vres(im, level);
    float A = dst0.y * dst0.y + dst1.y * dst1.y + 1.0f;
    float B = -2.0f * (dst0.x * dst0.y + dst1.x * dst1.y);
    float C = dst0.x * dst0.x + dst1.x * dst1.x + 1.0f;
    const float invf = 1.0f / (A * C - B * B * 0.25f);
    A *= invf; B *= invf; C *= invf;
    const float det = -B * B + 4.0f * A * C;
    const float idet = 1.0f / det;
    const float usqrt = sqrtf(det * C), vsqrt = sqrtf(det * A);
    const int64_t s0 = f2i_sat(ceilf(st.x - 2.0f * idet * usqrt)), s1 = f2i_sat(floorf(st.x + 2.0f * idet * usqrt));
    const int64_t t0 = f2i_sat(ceilf(st.y - 2.0f * idet * vsqrt)), t1 = f2i_sat(floorf(st.y + 2.0f * idet * vsqrt));
    RGB sum(0.0f); float sum_wts = 0.0f;
    // the footprint is bounded: the minor axis is >= 1 texel at this level and the eccentricity <= max_anisotropy
    if (s1 - s0 > 4096 || t1 - t0 > 4096) return RGB(0.0f) / sum_wts;   // NaN footprint: same NaN the reference's empty sum gives
    for (int64_t it = t0; it <= t1; ++it) {
        const float tt = (float)it - st.y;
        for (int64_t is = s0; is <= s1; ++is) {
            const float ss = (float)is - st.x;
            const float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
            if (r2 < 1.0f) {
                const uint32_t idx = min(f2u32_sat(r2 * 128.0f), 127u);
                const float weight = lut[idx];
                sum = sum + mip_texel(im, wrap, level, is, it) * weight;
                sum_wts += weight;
            }
        }
    }
    return sum / sum_wts;
}
PT_DEV RGB mip_lookup2(const DevImage &im, const PtTexture &T, const float *lut, P2 st, P2 dst0, P2 dst1) {  // mipmap.rs:225-258
    if (T.trilinear) {
        const float x = maxf(fabsf(dst0.x), fabsf(dst0.y)), y = maxf(fabsf(dst1.x), fabsf(dst1.y));
        return mip_lookup(im, T.wrap, st, maxf(x, y));
    }
    if (dst0.x * dst0.x + dst0.y * dst0.y < dst1.x * dst1.x + dst1.y * dst1.y) { const P2 tmp = dst0; dst0 = dst1; dst1 = tmp; }
    const float majorl = sqrtf(dst0.x * dst0.x + dst0.y * dst0.y);
    float minorl = sqrtf(dst1.x * dst1.x + dst1.y * dst1.y);
    if (minorl * T.max_anisotropy < majorl && minorl > 0.0f) {
        const float scale = majorl / (minorl * T.max_anisotropy);
        dst1.x *= scale; dst1.y *= scale;
        minorl *= scale;
    }
    if (minorl == 0.0f) return mip_triangle(im, T.wrap, 0, st);
    const float lod = maxf((float)im.n_levels - 1.0f + dm_log2f(minorl), 0.0f);
    const int ilod = (int)f2u32_sat(floorf(lod));
    return rgb_lerp(lod - (float)ilod, mip_ewa(im, T.wrap, lut, ilod, st, dst0, dst1), mip_ewa(im, T.wrap, lut, ilod + 1, st, dst0, dst1));
}

PT_DEV P2 tex_sphere_map(const M4 &w2t, V3 p) {  // texture.rs:165-175
    const V3 vec = normalize(xf_point(w2t, p) - V3(0.0f, 0.0f, 0.0f));
    const float theta = spherical_theta(vec), phi = spherical_phi(vec);
    return P2(theta * kInvPi, phi * kInv2Pi);
}
PT_DEV P2 tex_cylinder_map(const M4 &w2t, V3 p) {  // texture.rs:213-220
    const V3 vec = normalize(xf_point(w2t, p) - V3(0.0f, 0.0f, 0.0f));
    return P2(kPi + dm_atan2f(vec.y, vec.x) * kInv2Pi, vec.z);
}
PT_DEV P2 tex_map2d(const PtTexture &T, const TexCtx &c, P2 &dstdx, P2 &dstdy) {
    if (T.mapping == PT_MAP_UV) {
        dstdx = P2(T.su * c.dudx, T.sv * c.dvdx); dstdy = P2(T.su * c.dudy, T.sv * c.dvdy);
        return P2(T.su * c.uv.x + T.du, T.sv * c.uv.y + T.dv);
    }
    if (T.mapping == PT_MAP_PLANAR) {
        const V3 vs(T.vs[0], T.vs[1], T.vs[2]), vt(T.vt[0], T.vt[1], T.vt[2]);
        dstdx = P2(dot(c.dpdx, vs), dot(c.dpdx, vt)); dstdy = P2(dot(c.dpdy, vs), dot(c.dpdy, vt));
        return P2(T.du + dot(c.p, vs), T.dv + dot(c.p, vt));
    }
    M4 w2t; for (int i = 0; i < 16; ++i) w2t.m[i] = T.world_to_texture[i];
    const bool sph = T.mapping == PT_MAP_SPHERICAL;
    const P2 st = sph ? tex_sphere_map(w2t, c.p) : tex_cylinder_map(w2t, c.p);
    const float delta = 0.1f;
    const V3 px = c.p + c.dpdx * delta, py = c.p + c.dpdy * delta;
    const P2 sx = sph ? tex_sphere_map(w2t, px) : tex_cylinder_map(w2t, px);
    const P2 sy = sph ? tex_sphere_map(w2t, py) : tex_cylinder_map(w2t, py);
    dstdx = P2((sx.x - st.x) / delta, (sx.y - st.y) / delta);
    dstdy = P2((sy.x - st.x) / delta, (sy.y - st.y) / delta);
    if (dstdx.y > 0.5f) dstdx.y = 1.0f - dstdx.y; else if (dstdx.y < -0.5f) dstdx.y = -(dstdx.y + 1.0f);
    if (dstdy.y > 0.5f) dstdy.y = 1.0f - dstdy.y; else if (dstdy.y < -0.5f) dstdy.y = -(dstdy.y + 1.0f);
    return st;
}
// ---- Perlin noise (core/texture.rs:311-438); quirks as in the oracle: saturating `floor() as usize`, turbulence's `o + |n|`,
// fbm's ln(x) * 1.442695 vs turbulence's f32::log2.
__device__ const uint8_t kNoisePerm[512] = {PT_NOISE_PERM_VALUES};
PT_DEV float noise_grad(uint32_t x, uint32_t y, uint32_t z, float dx, float dy, float dz) {
    const uint32_t h = kNoisePerm[kNoisePerm[kNoisePerm[x] + y] + z] & 15u;
    const float u = (h < 8u || h == 12u || h == 13u) ? dx : dy;
    const float v = (h < 4u || h == 12u || h == 13u) ? dy : dz;
    return ((h & 1u) ? -u : u) + ((h & 2u) ? -v : v);
}
PT_DEV float noise_weight(float t) { const float t3 = t * t * t, t4 = t3 * t; return 6.0f * t4 * t - 15.0f * t4 + 10.0f * t3; }
PT_DEV float flerp(float t, float a, float b) { return (1.0f - t) * a + t * b; }
PT_DEV uint64_t f2u64_sat(float f) { if (!(f > 0.0f)) return 0ull; if (f >= 18446744073709551616.0f) return ~0ull; return (uint64_t)f; }
__device__ __noinline__ float noise3(float x, float y, float z) {
    const uint64_t ux = f2u64_sat(floorf(x)), uy = f2u64_sat(floorf(y)), uz = f2u64_sat(floorf(z));
    const float dx = x - (float)ux, dy = y - (float)uy, dz = z - (float)uz;
    const uint32_t ix = (uint32_t)(ux & 255ull), iy = (uint32_t)(uy & 255ull), iz = (uint32_t)(uz & 255ull);
    const float w000 = noise_grad(ix, iy, iz, dx, dy, dz), w100 = noise_grad(ix + 1, iy, iz, dx - 1.0f, dy, dz);
    const float w010 = noise_grad(ix, iy + 1, iz, dx, dy - 1.0f, dz), w110 = noise_grad(ix + 1, iy + 1, iz, dx - 1.0f, dy - 1.0f, dz);
    const float w001 = noise_grad(ix, iy, iz + 1, dx, dy, dz - 1.0f), w101 = noise_grad(ix + 1, iy, iz + 1, dx - 1.0f, dy, dz - 1.0f);
    const float w011 = noise_grad(ix, iy + 1, iz + 1, dx, dy - 1.0f, dz - 1.0f), w111 = noise_grad(ix + 1, iy + 1, iz + 1, dx - 1.0f, dy - 1.0f, dz - 1.0f);
    const float wx = noise_weight(dx), wy = noise_weight(dy), wz = noise_weight(dz);
    const float x00 = flerp(wx, w000, w100), x10 = flerp(wx, w010, w110), x01 = flerp(wx, w001, w101), x11 = flerp(wx, w011, w111);
    const float y0 = flerp(wy, x00, x10), y1 = flerp(wy, x01, x11);
    return flerp(wz, y0, y1);
}
PT_DEV float pclamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }   // pbrt.rs clamp (NaN passes through)
PT_DEV float smooth_step(float mn, float mx, float value) { const float v = pclamp((value - mn) / (mx - mn), 0.0f, 1.0f); return v * v * (-2.0f * v + 3.0f); }
PT_DEV float noise_fbm(V3 p, V3 dpdx, V3 dpdy, float omega, uint32_t max_octaves) {
    const float len2 = maxf(length_squared(dpdx), length_squared(dpdy));
    const float n = pclamp(-1.0f - 0.5f * (dm_logf(len2) * 1.442695040888963387f), 0.0f, (float)max_octaves);
    const uint64_t nint = f2u64_sat(floorf(n));
    float sum = 0.0f, lambda = 1.0f, o = 1.0f;
    for (uint64_t i = 0; i < nint; ++i) { const V3 q = p * lambda; sum += o * noise3(q.x, q.y, q.z); lambda *= 1.99f; o *= omega; }
    const float npartial = n - (float)nint;
    const V3 q = p * lambda;
    sum += o * smooth_step(0.3f, 0.7f, npartial) * noise3(q.x, q.y, q.z);
    return sum;
}
PT_DEV float noise_turbulence(V3 p, V3 dpdx, V3 dpdy, float omega, uint32_t max_octaves) {
    const float len2 = maxf(length_squared(dpdx), length_squared(dpdy));
    const float n = pclamp(-1.0f - 0.5f * dm_log2f(len2), 0.0f, (float)max_octaves);
    const uint64_t nint = f2u64_sat(floorf(n));
    float sum = 0.0f, lambda = 1.0f, o = 1.0f;
    for (uint64_t i = 0; i < nint; ++i) { const V3 q = p * lambda; sum += o + fabsf(noise3(q.x, q.y, q.z)); lambda *= 1.99f; o *= omega; }
    const float npartial = n - (float)nint;
    const V3 q = p * lambda;
    sum += o + flerp(smooth_step(0.3f, 0.7f, npartial), 0.2f, fabsf(noise3(q.x, q.y, q.z)));
    for (uint64_t i = nint; i < (uint64_t)max_octaves; ++i) { sum += o * 0.2f; o *= omega; }
    return sum;
}

PT_DEV bool tex_even_sum(float a, float b) { return ((f2i_sat(floorf(a)) + f2i_sat(floorf(b))) % 2) == 0; }   // isize % 2 == 0

// Evaluate texture `root` (an index into s.textures) at the interaction. A real function (not inlined): build_bsdf
// calls it from a dozen parameter sites of the largest kernels.
__device__ __noinline__ RGB tex_eval(const DeviceScene &s, int root, const TexCtx &c) {
    RGB stk[kTexStack];
    int sp = 0;
    const uint32_t begin = s.tex_prog_offset[root], end = s.tex_prog_offset[root + 1];
    for (uint32_t pc = begin; pc < end; ++pc) {
        const PtTexture &T = s.textures[s.tex_prog[pc]];
        RGB v(0.0f);
        switch (T.type) {
        case PT_TEX_CONSTANT: v = RGB(T.value[0], T.value[1], T.value[2]); break;
        case PT_TEX_SCALE: { const RGB b = stk[--sp], a = stk[--sp]; v = a * b; break; }
        case PT_TEX_MIX: { const float amt = stk[--sp].r; const RGB t2 = stk[--sp], t1 = stk[--sp]; v = t1 * (1.0f - amt) + t2 * amt; break; }
        case PT_TEX_CHECKERBOARD2D: {
            const RGB t2 = stk[--sp], t1 = stk[--sp];
            P2 dstdx, dstdy;
            const P2 st = tex_map2d(T, c, dstdx, dstdy);
            bool point = !T.aa_closedform;
            float area2 = 0.0f;
            if (!point) {
                const float ds = maxf(fabsf(dstdx.x), fabsf(dstdy.x)), dt = maxf(fabsf(dstdx.y), fabsf(dstdy.y));
                const float s0 = st.x - ds, s1 = st.x + ds, t0 = st.y - dt, t1_ = st.y + dt;
                if (floorf(s0) == floorf(s1) && floorf(t0) == floorf(t1_)) point = true;
                else {
                    auto bump = [](float x) { return floorf(x / 2.0f) + 2.0f * maxf(x / 2.0f - floorf(x / 2.0f) - 0.5f, 0.0f); };
                    const float sint = (bump(s1) - bump(s0)) / (2.0f * ds), tint = (bump(t1_) - bump(t0)) / (2.0f * dt);
                    area2 = sint * tint - 2.0f * sint * tint;   // as written in checkerboard.rs:63
                    if (ds > 1.0f || dt > 1.0f) area2 = 0.5f;
                }
            }
            v = point ? (tex_even_sum(st.x, st.y) ? t1 : t2) : (t1 * (1.0f - area2) + t2 * area2);
            break;
        }
        case PT_TEX_CHECKERBOARD3D: {
            const RGB t2 = stk[--sp], t1 = stk[--sp];
            M4 w2t; for (int i = 0; i < 16; ++i) w2t.m[i] = T.world_to_texture[i];
            const V3 p = xf_point(w2t, c.p);
            const bool even = ((f2i_sat(floorf(p.x)) + f2i_sat(floorf(p.y)) + f2i_sat(floorf(p.z))) % 2) == 0;
            v = even ? t1 : t2;
            break;
        }
        case PT_TEX_IMAGEMAP: {
            P2 dstdx, dstdy;
            const P2 st = tex_map2d(T, c, dstdx, dstdy);
            v = mip_lookup2(s.images[T.image], T, s.ewa_lut, st, dstdx, dstdy);
            break;
        }
        case PT_TEX_UV: {
            P2 dstdx, dstdy;
            const P2 st = tex_map2d(T, c, dstdx, dstdy);
            v = RGB(st.x - floorf(st.x), st.y - floorf(st.y), 0.0f);
            break;
        }
        case PT_TEX_FBM: case PT_TEX_WRINKLED: case PT_TEX_WINDY: case PT_TEX_MARBLE: {  // IdentityMapping3D (texture.rs:281-297)
            M4 w2t; for (int i = 0; i < 16; ++i) w2t.m[i] = T.world_to_texture[i];
            const V3 dpdx = xf_vector(w2t, c.dpdx), dpdy = xf_vector(w2t, c.dpdy);
            V3 p = xf_point(w2t, c.p);
            if (T.type == PT_TEX_FBM) v = RGB(noise_fbm(p, dpdx, dpdy, T.omega, T.octaves));
            else if (T.type == PT_TEX_WRINKLED) v = RGB(noise_turbulence(p, dpdx, dpdy, T.omega, T.octaves));
            else if (T.type == PT_TEX_WINDY) {
                const float wstrength = noise_fbm(p * 0.1f, dpdx * 0.1f, dpdy * 0.1f, 0.5f, 3u);
                const float wheight = noise_fbm(p, dpdx, dpdy, 0.5f, 6u);
                v = RGB(fabsf(wstrength) * wheight);
            } else {
                p = p * T.marble_scale;
                const float fb = noise_fbm(p, dpdx * T.marble_scale, dpdy * T.marble_scale, T.omega, T.octaves);
                const float marble = p.y + T.variation * fb;
                const float t = 0.5f + 0.5f * dm_sinf(marble);
                const uint64_t f6 = f2u64_sat(floorf(t * 6.0f));
                const int first = (int)(f6 < 5ull ? f6 : 5ull);
                auto Cm = [](int i) -> RGB {   // marble.rs:10-14
                    return (i == 3) ? RGB(0.5f, 0.5f, 0.5f) : (i == 4) ? RGB(0.6f, 0.59f, 0.58f) : (i == 7) ? RGB(0.2f, 0.2f, 0.33f) : RGB(0.58f, 0.58f, 0.6f);
                };
                const RGB c0 = Cm(first), c1 = Cm(first + 1), c2 = Cm(first + 2), c3 = Cm(first + 3);
                RGB s0 = c0 * (1.0f - t) + c1 * t, s1 = c1 * (1.0f - t) + c2 * t;
                const RGB s2 = c2 * (1.0f - t) + c3 * t;
                s0 = s0 * (1.0f - t) + s1 * t; s1 = s1 * (1.0f - t) + s2 * t;
                v = (s0 * (1.0f - t) + s1 * t) * 1.5f;
            }
            break;
        }
        case PT_TEX_DOTS: {
            const RGB inside = stk[--sp], outside = stk[--sp];
            P2 dstdx, dstdy;
            const P2 st = tex_map2d(T, c, dstdx, dstdy);
            const uint64_t scell = f2u64_sat(floorf(st.x + 0.5f)), tcell = f2u64_sat(floorf(st.y + 0.5f));
            v = outside;
            if (noise3((float)scell + 0.5f, (float)tcell + 0.5f, 0.5f) > 0.0f) {
                const float radius = 0.35f, max_shift = 0.5f - radius;
                const float scenter = (float)scell + max_shift * noise3((float)scell + 1.5f, (float)tcell + 2.8f, 0.5f);
                const float tcenter = (float)tcell + max_shift * noise3((float)scell + 4.5f, (float)tcell + 9.8f, 0.5f);
                const float ds = st.x - scenter, dt = st.y - tcenter;
                if (ds * ds + dt * dt < radius * radius) v = inside;
            }
            break;
        }
        default: {  // PT_TEX_BILERP
            P2 dstdx, dstdy;
            const P2 st = tex_map2d(T, c, dstdx, dstdy);
            const RGB v00(T.v00[0], T.v00[1], T.v00[2]), v01(T.v01[0], T.v01[1], T.v01[2]), v10(T.v10[0], T.v10[1], T.v10[2]), v11(T.v11[0], T.v11[1], T.v11[2]);
            v = v00 * (1.0f - st.y) * (1.0f - st.x) + v01 * (1.0f - st.x) * st.y + v10 * (1.0f - st.y) * st.x + v11 * st.y * st.x;
            break;
        }
        }
        stk[sp++] = v;
    }
    return stk[0];
}

// SurfaceInteraction::compute_differentials (interaction.rs:269-342); all zero without ray differentials.
PT_DEV bool solve_2x2(const float a[2][2], const float b[2], float &x0, float &x1) {  // transform.rs:174-186
    const float det = a[0][0] * a[1][1] - a[0][1] * a[1][0];
    if (fabsf(det) < 1.0e-10f) return false;
    x0 = (a[1][1] * b[0] - a[0][1] * b[1]) / det;
    x1 = (a[0][0] * b[1] - a[1][0] * b[0]) / det;
    return !(x0 != x0 || x1 != x1);
}
PT_DEV float v3c(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
PT_DEV TexCtx compute_differentials(const SurfaceInteraction &si, const RayDiff &rd) {
    TexCtx c; c.p = si.p; c.uv = si.uv; c.dpdx = V3(0.0f, 0.0f, 0.0f); c.dpdy = V3(0.0f, 0.0f, 0.0f);
    c.dudx = c.dvdx = c.dudy = c.dvdy = 0.0f;
    if (!rd.has) return c;
    const float d = dot(si.n, si.p);
    const float tx = -(dot(si.n, rd.rx_o) - d) / dot(si.n, rd.rx_d);
    if (__builtin_isinf(tx) || tx != tx) return c;
    const V3 px = rd.rx_o + rd.rx_d * tx;
    const float ty = -(dot(si.n, rd.ry_o) - d) / dot(si.n, rd.ry_d);
    if (__builtin_isinf(ty) || ty != ty) return c;
    const V3 py = rd.ry_o + rd.ry_d * ty;
    c.dpdx = px - si.p; c.dpdy = py - si.p;
    int d0, d1;
    if (fabsf(si.n.x) > fabsf(si.n.y) && fabsf(si.n.x) > fabsf(si.n.z)) { d0 = 1; d1 = 2; }
    else if (fabsf(si.n.y) > fabsf(si.n.z)) { d0 = 0; d1 = 2; }
    else { d0 = 0; d1 = 1; }
    const float A[2][2] = {{v3c(si.dpdu, d0), v3c(si.dpdv, d0)}, {v3c(si.dpdu, d1), v3c(si.dpdv, d1)}};
    const float Bx[2] = {v3c(px, d0) - v3c(si.p, d0), v3c(px, d1) - v3c(si.p, d1)};
    const float By[2] = {v3c(py, d0) - v3c(si.p, d0), v3c(py, d1) - v3c(si.p, d1)};
    if (!solve_2x2(A, Bx, c.dudx, c.dvdx)) { c.dudx = 0.0f; c.dvdx = 0.0f; }
    if (!solve_2x2(A, By, c.dudy, c.dvdy)) { c.dudy = 0.0f; c.dvdy = 0.0f; }
    return c;
}

}  // namespace ptd
