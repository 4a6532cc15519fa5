// dev_math.h -- device-side scalar/vector arithmetic for the gfx950 kernels.
// Semantics follow the reference bit for bit (f32, no FMA contraction, IEEE div/sqrt):
//   core/pbrt.rs:23-34,80-112,136-144,172-208; core/geometry/vector.rs:226-369,481-559;
//   core/geometry/geometry.rs:6-54; core/transform.rs:413-459,496-577; core/spectrum.rs:123-127,484-502.
// Compile with -ffp-contract=off (Rust never fuses a*b+c).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__
#define PT_HD __host__ __device__ __forceinline__
#ifdef PT_NOINLINE_HEAVY   // code-size experiment: heavy helpers as real functions
#define PT_HDX __host__ __device__ __noinline__
#define PT_DEVX __device__ __noinline__
#else
#define PT_HDX PT_HD
#define PT_DEVX PT_DEV
#endif
// The scalar f64 transcendentals (sin/cos, atan2, ln, exp: numbers in, numbers out, no memory arguments) are REAL functions since round 3:
// inlined at each of their ~10 call sites they made up 20 % of the matte shade kernel's code and, more to the point, 16 of its 202 VGPRs
// (k_shade<1, 0, 1>: 84 -> 67 KB, 202 -> 186 VGPRs; the specular-only kernel 145 -> 127 = four waves per SIMD). By itself that changes
// nothing (the kernel's instruction cache hit rate was 99.98 % already: SQC_ICACHE counters, profiles/r3); it is what lets the matte kernel
// run THREE waves per SIMD with 32 bytes of scratch instead of 76 (kern_shade.h): 93.4 -> 83.7 ms per C2 step. -DPT_INLINE_MATH restores the old form.
#ifdef PT_INLINE_MATH
#define PT_HDM PT_HDX
#else
#define PT_HDM __host__ __device__ __noinline__ inline
#endif

namespace ptd {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kPiOver2 = 1.57079632679489661923f;
constexpr float kPiOver4 = 0.78539816339744830961f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kInv2Pi = 0.15915494309189533577f;
constexpr float kShadowEps = 0.0001f;
constexpr float kMachEps = 5.9604644775390625e-08f;  // f32::EPSILON * 0.5
constexpr float kOneMinusEps = 0x1.fffffep-1f;
#define PT_INF __builtin_huge_valf()

PT_HD float gammaf(int n) { return ((float)n * kMachEps) / (1.0f - (float)n * kMachEps); }

PT_HD uint32_t f2bits(float f) { return __builtin_bit_cast(uint32_t, f); }
PT_HD float bits2f(uint32_t u) { return __builtin_bit_cast(float, u); }

PT_HD float next_float_up(float v) {  // pbrt.rs:80-95
    if (__builtin_isinf(v) && v > 0.0f) return v;
    float i = v;
    if (i == -0.0f) i = 0.0f;
    uint32_t ui = f2bits(i);
    if (i >= 0.0f) ui += 1; else ui -= 1;
    return bits2f(ui);
}
PT_HD float next_float_down(float v) {  // pbrt.rs:97-112
    if (__builtin_isinf(v) && v < 0.0f) return v;
    float i = v;
    if (i == 0.0f) i = -0.0f;
    uint32_t ui = f2bits(i);
    if (i > 0.0f) ui -= 1; else ui += 1;
    return bits2f(ui);
}
PT_HD float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }  // NaN passes through
PT_HD float lerpf(float t, float x, float y) { return x * (1.0f - t) + y * t; }
PT_HD float maxf(float a, float b) { return fmaxf(a, b); }  // f32::max: non-NaN operand wins
PT_HD float minf(float a, float b) { return fminf(a, b); }
// Rust float->int `as` casts saturate, NaN -> 0.
PT_HD int64_t f2i_sat(float f) {
    if (f != f) return 0;
    if (f >= 9.2233720368547758e18f) return INT64_MAX;
    if (f <= -9.2233720368547758e18f) return INT64_MIN;
    return (int64_t)f;
}
PT_HD uint32_t f2u32_sat(float f) {
    if (!(f > 0.0f)) return 0;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

// ---- deterministic transcendentals (DESIGN.md "deterministic math"): double Taylor + single rounding ----
constexpr double kDmPio2Hi = 0x1.921fb54442d18p+0;
constexpr double kDmPio2Lo = 0x1.1a62633145c07p-54;
constexpr double kDmPi = 0x1.921fb54442d18p+1;
constexpr double kDm2OverPi = 0x1.45f306dc9c883p-1;
constexpr double kDmLn2 = 0x1.62e42fefa39efp-1;

PT_HD double dm_sin_k(double r) {
    double r2 = r * r;
    double p = 1.0 / 355687428096000.0;
    p = p * r2 - 1.0 / 1307674368000.0;
    p = p * r2 + 1.0 / 6227020800.0;
    p = p * r2 - 1.0 / 39916800.0;
    p = p * r2 + 1.0 / 362880.0;
    p = p * r2 - 1.0 / 5040.0;
    p = p * r2 + 1.0 / 120.0;
    p = p * r2 - 1.0 / 6.0;
    return r + r * (r2 * p);
}
PT_HD double dm_cos_k(double r) {
    double r2 = r * r;
    double p = 1.0 / 6402373705728000.0;
    p = p * r2 - 1.0 / 20922789888000.0;
    p = p * r2 + 1.0 / 87178291200.0;
    p = p * r2 - 1.0 / 479001600.0;
    p = p * r2 + 1.0 / 3628800.0;
    p = p * r2 - 1.0 / 40320.0;
    p = p * r2 + 1.0 / 720.0;
    p = p * r2 - 1.0 / 24.0;
    p = p * r2 + 0.5;
    return 1.0 - r2 * p;
}
// sin and cos of one argument share the reduction.
struct DmSC { float s, c; };
PT_HD void dm_sincosf_impl(float xf, float &s, float &c);
PT_HDM DmSC dm_sincosf2(float xf) { DmSC r; dm_sincosf_impl(xf, r.s, r.c); return r; }
PT_HD void dm_sincosf(float xf, float &s, float &c) { DmSC r = dm_sincosf2(xf); s = r.s; c = r.c; }
PT_HD void dm_sincosf_impl(float xf, float &s, float &c) {
    double x = xf;
    if (!(__builtin_fabs(x) < 1.0e9)) { s = c = __builtin_nanf(""); return; }
    double kd = __builtin_floor(x * kDm2OverPi + 0.5);
    double r = (x - kd * kDmPio2Hi) - kd * kDmPio2Lo;
    int q = (int)((long long)kd & 3);
    double sk = dm_sin_k(r), ck = dm_cos_k(r);
    double sv = (q == 0) ? sk : (q == 1) ? ck : (q == 2) ? -sk : -ck;
    double cv = (q == 0) ? ck : (q == 1) ? -sk : (q == 2) ? -ck : sk;
    s = (float)sv; c = (float)cv;
}
PT_HD float dm_sinf(float x) { float s, c; dm_sincosf(x, s, c); return s; }
PT_HD float dm_cosf(float x) { float s, c; dm_sincosf(x, s, c); return c; }

PT_HD double dm_atan_tab(int k) {
    switch (k) {
    case 0: return 0x0.0p+0;
    case 1: return 0x1.fd5ba9aac2f6ep-4;
    case 2: return 0x1.f5b75f92c80ddp-3;
    case 3: return 0x1.6f61941e4def1p-2;
    case 4: return 0x1.dac670561bb4fp-2;
    case 5: return 0x1.1e00babdefeb4p-1;
    case 6: return 0x1.4978fa3269ee1p-1;
    case 7: return 0x1.700a7c5784634p-1;
    default: return 0x1.921fb54442d18p-1;
    }
}
PT_HD double dm_atan01(double z) {
    int k = (int)(z * 8.0 + 0.5);
    double c = (double)k / 8.0;
    double t = (z - c) / (1.0 + z * c);
    double t2 = t * t;
    double p = 1.0 / 17.0;
    p = p * t2 - 1.0 / 15.0;
    p = p * t2 + 1.0 / 13.0;
    p = p * t2 - 1.0 / 11.0;
    p = p * t2 + 1.0 / 9.0;
    p = p * t2 - 1.0 / 7.0;
    p = p * t2 + 1.0 / 5.0;
    p = p * t2 - 1.0 / 3.0;
    return dm_atan_tab(k) + (t + t * (t2 * p));
}
PT_HD double dm_atan_pos(double z) { return (z > 1.0) ? kDmPio2Hi - dm_atan01(1.0 / z) : dm_atan01(z); }
PT_HDM double dm_atan2d(double y, double x) {
    if (x != x || y != y) return __builtin_nan("");
    double ay = __builtin_fabs(y), ax = __builtin_fabs(x);
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ax == 0.0) a = kDmPio2Hi;
    else if (__builtin_isinf(ax) && __builtin_isinf(ay)) a = kDmPio2Hi * 0.5;
    else a = dm_atan_pos(ay / ax);
    if (__builtin_signbit(x)) a = kDmPi - a;
    return __builtin_signbit(y) ? -a : a;
}
PT_HD float dm_atan2f(float y, float x) { return (float)dm_atan2d((double)y, (double)x); }
PT_HD float dm_acosf(float xf) {
    double x = xf;
    if (!(x >= -1.0 && x <= 1.0)) return __builtin_nanf("");
    double s = __builtin_sqrt((1.0 - x) * (1.0 + x));
    return (float)dm_atan2d(s, x);
}
// ln of a positive finite double (the core of dm_logf, also used by dm_powf)
PT_HDM double dm_logd_pos(double x) {
    uint64_t bits = __builtin_bit_cast(uint64_t, x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __builtin_bit_cast(double, bits);
    if (m > 0x1.6a09e667f3bcdp+0) { m = m * 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = 2.0 / 21.0;
    p = p * s2 + 2.0 / 19.0;
    p = p * s2 + 2.0 / 17.0;
    p = p * s2 + 2.0 / 15.0;
    p = p * s2 + 2.0 / 13.0;
    p = p * s2 + 2.0 / 11.0;
    p = p * s2 + 2.0 / 9.0;
    p = p * s2 + 2.0 / 7.0;
    p = p * s2 + 2.0 / 5.0;
    p = p * s2 + 2.0 / 3.0;
    p = p * s2 + 2.0;
    return (double)e * kDmLn2 + s * p;
}
PT_HD float dm_logf(float xf) {
    if (xf != xf || xf < 0.0f) return __builtin_nanf("");
    if (xf == 0.0f) return -PT_INF;
    if (__builtin_isinf(xf)) return xf;
    return (float)dm_logd_pos((double)xf);
}
// e^y for |y| <= 700 in f64: y = k ln2 + r, Taylor series of e^r (|r| <= 0.35, degree 14), scaled by 2^k. Shared with the
// oracle (ref_math.h) so that f32::powf / f32::exp call sites (materials/disney.rs) are bit-identical on both sides.
PT_HDM double dm_expd(double y) {
    if (y > 700.0) y = 700.0;
    if (y < -700.0) y = -700.0;
    const double kf = __builtin_floor(y * (1.0 / kDmLn2) + 0.5);
    const double r = y - kf * kDmLn2;
    double p = 1.0 / 87178291200.0;
    p = p * r + 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    const uint64_t bits = (uint64_t)((int64_t)kf + 1023) << 52;
    return p * __builtin_bit_cast(double, bits);
}
PT_HD float dm_powf(float a, float b) {   // a > 0 (f32::powf call sites on the path have a positive base)
    if (!(a > 0.0f)) return (a == 0.0f) ? (b == 0.0f ? 1.0f : 0.0f) : __builtin_nanf("");
    return (float)dm_expd((double)b * dm_logd_pos((double)a));
}


// ---- vectors ---------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
    PT_HD V3() : x(0), y(0), z(0) {}
    PT_HD V3(float a, float b, float c) : x(a), y(b), z(c) {}
    PT_HD float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
PT_HD V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
PT_HD V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
PT_HD V3 operator/(V3 a, float s) { float d = 1.0f / s; return V3(a.x * d, a.y * d, a.z * d); }  // vector.rs:481-495
PT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_HD float abs_dot(V3 a, V3 b) { return fabsf(dot(a, b)); }
PT_HD float length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
PT_HD float length(V3 a) { return sqrtf(length_squared(a)); }
PT_HD V3 normalize(V3 a) { return a / length(a); }
PT_HD V3 vabs(V3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
PT_HD V3 cross(V3 a, V3 b) {  // vector.rs:339-352: f64 products, one rounding
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return V3((float)((ay * bz) - (az * by)), (float)((az * bx) - (ax * bz)), (float)((ax * by) - (ay * bx)));
}
PT_HD int max_dimension(V3 v) { return (v.x > v.y) ? ((v.x > v.z) ? 0 : 2) : ((v.y > v.z) ? 1 : 2); }
PT_HD float max_component(V3 v) { return maxf(v.x, maxf(v.y, v.z)); }
PT_HD V3 face_forward(V3 n, V3 v) { return (dot(n, v) < 0.0f) ? -n : n; }
PT_HD void coordinate_system(V3 v1, V3 &v2, V3 &v3) {  // vector.rs:551-559
    if (fabsf(v1.x) > fabsf(v1.y)) v2 = V3(-v1.z, 0.0f, v1.x) / sqrtf(v1.x * v1.x + v1.z * v1.z);
    else v2 = V3(0.0f, v1.z, -v1.y) / sqrtf(v1.y * v1.y + v1.z * v1.z);
    v3 = cross(v1, v2);
}
PT_HD float distance_squared(V3 a, V3 b) { return length_squared(a - b); }

struct P2 { float x, y; PT_HD P2() : x(0), y(0) {} PT_HD P2(float a, float b) : x(a), y(b) {} };

// ---- RGB spectrum (pbrt_macros/src/lib.rs:113-668) ----------------------------------------------
struct RGB {
    float r, g, b;
    PT_HD RGB() : r(0), g(0), b(0) {}
    PT_HD explicit RGB(float v) : r(v), g(v), b(v) {}
    PT_HD RGB(float a, float b_, float c) : r(a), g(b_), b(c) {}
    PT_HD bool is_black() const { return r == 0.0f && g == 0.0f && b == 0.0f; }
    PT_HD float y() const { return 0.212671f * r + 0.715160f * g + 0.072169f * b; }
    PT_HD float max_component_value() const { return maxf(maxf(r, g), b); }
    PT_HD bool has_nans() const { return r != r || g != g || b != b; }
    PT_HD RGB clamps(float lo, float hi) const { return RGB(clampf(r, lo, hi), clampf(g, lo, hi), clampf(b, lo, hi)); }
};
PT_HD RGB operator+(RGB a, RGB b) { return RGB(a.r + b.r, a.g + b.g, a.b + b.b); }
PT_HD RGB operator-(RGB a, RGB b) { return RGB(a.r - b.r, a.g - b.g, a.b - b.b); }
PT_HD RGB operator*(RGB a, RGB b) { return RGB(a.r * b.r, a.g * b.g, a.b * b.b); }
PT_HD RGB operator*(RGB a, float s) { return RGB(a.r * s, a.g * s, a.b * s); }
PT_HD RGB operator/(RGB a, float s) { return RGB(a.r / s, a.g / s, a.b / s); }
PT_HD RGB operator/(RGB a, RGB b) { return RGB(a.r / b.r, a.g / b.g, a.b / b.b); }
PT_HD RGB sqrt_rgb(RGB a) { return RGB(sqrtf(a.r), sqrtf(a.g), sqrtf(a.b)); }
PT_HD void xyz_to_rgb(const float xyz[3], float rgb[3]) {  // spectrum.rs:484-492
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}
PT_HD void rgb_to_xyz(const float rgb[3], float xyz[3]) {  // spectrum.rs:494-502
    xyz[0] = 0.412453f * rgb[0] + 0.357580f * rgb[1] + 0.180423f * rgb[2];
    xyz[1] = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];
    xyz[2] = 0.019334f * rgb[0] + 0.119193f * rgb[1] + 0.950227f * rgb[2];
}

// ---- transforms (row-major float[16] == Matrix4x4.m[r][c]) -----------------------------------------
struct M4 { float m[16]; };
PT_HD V3 xf_point(const M4 &t, V3 p) {  // transform.rs:413-432
    float x = p.x, y = p.y, z = p.z;
    float xp = x * t.m[0] + y * t.m[1] + z * t.m[2] + t.m[3];
    float yp = x * t.m[4] + y * t.m[5] + z * t.m[6] + t.m[7];
    float zp = x * t.m[8] + y * t.m[9] + z * t.m[10] + t.m[11];
    float wp = x * t.m[12] + y * t.m[13] + z * t.m[14] + t.m[15];
    if (wp == 1.0f) return V3(xp, yp, zp);
    return V3(xp, yp, zp) / wp;
}
PT_HD V3 xf_point_err(const M4 &t, V3 p, V3 &perr) {  // transform.rs:434-459
    float x = p.x, y = p.y, z = p.z;
    float xp = x * t.m[0] + y * t.m[1] + z * t.m[2] + t.m[3];
    float yp = x * t.m[4] + y * t.m[5] + z * t.m[6] + t.m[7];
    float zp = x * t.m[8] + y * t.m[9] + z * t.m[10] + t.m[11];
    float wp = x * t.m[12] + y * t.m[13] + z * t.m[14] + t.m[15];
    float xs = fabsf(x * t.m[0]) + fabsf(y * t.m[1]) + fabsf(z * t.m[2]) + fabsf(t.m[3]);
    float ys = fabsf(x * t.m[4]) + fabsf(y * t.m[5]) + fabsf(z * t.m[6]) + fabsf(t.m[7]);
    float zs = fabsf(x * t.m[8]) + fabsf(y * t.m[9]) + fabsf(z * t.m[10]) + fabsf(t.m[11]);
    perr = V3(xs, ys, zs) * gammaf(3);
    if (wp == 1.0f) return V3(xp, yp, zp);
    return V3(xp, yp, zp) / wp;
}
PT_HD V3 xf_vector(const M4 &t, V3 v) {  // transform.rs:496-508
    float x = v.x, y = v.y, z = v.z;
    return V3(x * t.m[0] + y * t.m[1] + z * t.m[2], x * t.m[4] + y * t.m[5] + z * t.m[6], x * t.m[8] + y * t.m[9] + z * t.m[10]);
}

// geometry.rs:6-24
PT_HDX V3 offset_ray_origin(V3 p, V3 perr, V3 n, V3 w) {
    float d = dot(vabs(n), perr);
    V3 offset = n * d;
    if (dot(w, n) < 0.0f) offset = -offset;
    V3 po = p + offset;
    if (offset.x > 0.0f) po.x = next_float_up(po.x); else if (offset.x < 0.0f) po.x = next_float_down(po.x);
    if (offset.y > 0.0f) po.y = next_float_up(po.y); else if (offset.y < 0.0f) po.y = next_float_down(po.y);
    if (offset.z > 0.0f) po.z = next_float_up(po.z); else if (offset.z < 0.0f) po.z = next_float_down(po.z);
    return po;
}
PT_HD float spherical_theta(V3 v) { return dm_acosf(clampf(v.z, -1.0f, 1.0f)); }  // geometry.rs:39-42
PT_HD float spherical_phi(V3 v) { float p = dm_atan2f(v.y, v.x); return (p < 0.0f) ? p + 2.0f * kPi : p; }

// sampling.rs:153-176
PT_HD P2 concentric_sample_disk(P2 u) {
    float ox = u.x * 2.0f - 1.0f, oy = u.y * 2.0f - 1.0f;
    if (ox == 0.0f && oy == 0.0f) return P2(0.0f, 0.0f);
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) { r = ox; theta = kPiOver4 * (oy / ox); }
    else { r = oy; theta = kPiOver2 - kPiOver4 * (ox / oy); }
    float s, c; dm_sincosf(theta, s, c);
    return P2(c * r, s * r);
}

}  // namespace ptd
