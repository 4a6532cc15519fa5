// ORACLE -- TEST INFRASTRUCTURE ONLY. CPU restatement of pbrt-rust's per-sample render path.
// Nothing under oracle/ is linked, imported or executed by the product path.
// Parity status: the reference (Rust nightly + un-vendored crates) cannot be built here, so this
// restatement is pinned against the reference's own known-answer tests (tests/test_oracle_kats.py)
// but END-TO-END RADIANCE IS "parity unpinned" (SURVEY.md section 8c).
//
// ref_math.h: scalar/vector helpers.
//   core/pbrt.rs:23-34 (constants), :80-112 (next_float_up/down), :136-144 (lerp), :172-182 (clamp),
//   :184-204 (find_interval), :206-208 (gamma); core/geometry/vector.rs; core/geometry/normal.rs;
//   core/spectrum.rs:78-127,484-502; pbrt_macros/src/lib.rs:113-668 (spectrum ops).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <limits>

namespace ref {

typedef float Float;
static const Float PI = 3.14159265358979323846f;
static const Float PI_OVER2 = 1.57079632679489661923f;
static const Float PI_OVER4 = 0.78539816339744830961f;
static const Float INV_PI = 0.31830988618379067154f;
static const Float INV2_PI = 0.15915494309189533577f;
static const Float INV4_PI = 0.07957747154594766788f;
static const Float INF = std::numeric_limits<float>::infinity();
static const Float SHADOW_EPSILON = 0.0001f;
static const Float MACHINE_EPSILON = std::numeric_limits<float>::epsilon() * 0.5f;
static const Float ONE_MINUS_EPSILON = 0x1.fffffep-1f;  // core/rng.rs:4-5

inline Float gamma(int n) { return ((Float)n * MACHINE_EPSILON) / (1.0f - (Float)n * MACHINE_EPSILON); }

inline uint32_t float_to_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float bits_to_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// core/pbrt.rs:80-95
inline float next_float_up(float v) {
    if (std::isinf(v) && v > 0.0f) return v;
    float i = v;
    if (i == -0.0f) i = 0.0f;
    uint32_t ui = float_to_bits(i);
    if (i >= 0.0f) ui += 1; else ui -= 1;
    return bits_to_float(ui);
}
// core/pbrt.rs:97-112
inline float next_float_down(float v) {
    if (std::isinf(v) && v < 0.0f) return v;
    float i = v;
    if (i == 0.0f) i = -0.0f;
    uint32_t ui = float_to_bits(i);
    if (i > 0.0f) ui -= 1; else ui += 1;
    return bits_to_float(ui);
}

template <class T> inline T clampv(T val, T low, T high) {  // core/pbrt.rs:172-182 (NaN passes through)
    if (val < low) return low;
    else if (val > high) return high;
    return val;
}
inline Float lerp(Float t, Float x, Float y) { return x * (1.0f - t) + y * t; }
// Rust f32::max/min: the non-NaN operand wins (== fmaxf/fminf).
inline Float fmax_(Float a, Float b) { return std::fmax(a, b); }
inline Float fmin_(Float a, Float b) { return std::fmin(a, b); }

// Rust `as usize`/`as isize` from f32 saturate; NaN -> 0.
inline int64_t f2i_sat(Float f) {
    if (f != f) return 0;
    if (f >= 9.2233720368547758e18f) return INT64_MAX;
    if (f <= -9.2233720368547758e18f) return INT64_MIN;
    return (int64_t)f;
}
inline uint64_t f2u_sat(Float f) {
    if (!(f > 0.0f)) return 0;  // NaN and negatives -> 0
    if (f >= 1.8446744073709552e19f) return UINT64_MAX;
    return (uint64_t)f;
}

// core/pbrt.rs:184-204
template <class Pred> inline int find_interval(int size, Pred pred) {
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (pred(middle)) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    return clampv(first - 1, 0, size - 2);
}

// ---- deterministic transcendentals ---------------------------------------------------------
// The reference calls Rust's f32::sin/cos/acos/atan2/ln (platform libm, not pinned). To make the
// CPU oracle and the HIP kernels agree BIT FOR BIT, both evaluate the same double-precision
// Taylor/argument-reduction scheme with IEEE +,-,*,/ and sqrt only (no FMA contraction) and round
// once to f32. Against glibc's sinf/cosf/... the result differs by at most 1 ulp in rare cases
// (measured in tests/test_oracle_kats.py). Scheme (DESIGN.md "deterministic math"):
//   sin/cos : k = floor(x*2/pi + 0.5), r = (x - k*PIO2_HI) - k*PIO2_LO, Taylor to r^17 / r^18
//   atan    : z>1 -> pi/2 - atan(1/z); c = round(8z)/8, t = (z-c)/(1+z*c), table + Taylor to t^17
//   acos(x) = atan2(sqrt((1-x)(1+x)), x)
//   ln      : x = m*2^e, m in [sqrt(.5), sqrt(2)), s = (m-1)/(m+1), 2*(s + s^3/3 + ... + s^21/21) + e*ln2
static const double DM_PIO2_HI = 0x1.921fb54442d18p+0;
static const double DM_PIO2_LO = 0x1.1a62633145c07p-54;
static const double DM_PI = 0x1.921fb54442d18p+1;
static const double DM_2_OVER_PI = 0x1.45f306dc9c883p-1;
static const double DM_LN2 = 0x1.62e42fefa39efp-1;

inline double dm_sin_k(double r) {
    double r2 = r * r;
    double p = 1.0 / 355687428096000.0;                // 1/17!
    p = p * r2 - 1.0 / 1307674368000.0;                // 1/15!
    p = p * r2 + 1.0 / 6227020800.0;                   // 1/13!
    p = p * r2 - 1.0 / 39916800.0;                     // 1/11!
    p = p * r2 + 1.0 / 362880.0;                       // 1/9!
    p = p * r2 - 1.0 / 5040.0;                         // 1/7!
    p = p * r2 + 1.0 / 120.0;                          // 1/5!
    p = p * r2 - 1.0 / 6.0;                            // 1/3!
    return r + r * (r2 * p);
}
inline double dm_cos_k(double r) {
    double r2 = r * r;
    double p = 1.0 / 6402373705728000.0;               // 1/18!
    p = p * r2 - 1.0 / 20922789888000.0;               // 1/16!
    p = p * r2 + 1.0 / 87178291200.0;                  // 1/14!
    p = p * r2 - 1.0 / 479001600.0;                    // 1/12!
    p = p * r2 + 1.0 / 3628800.0;                      // 1/10!
    p = p * r2 - 1.0 / 40320.0;                        // 1/8!
    p = p * r2 + 1.0 / 720.0;                          // 1/6!
    p = p * r2 - 1.0 / 24.0;                           // 1/4!
    p = p * r2 + 0.5;
    return 1.0 - r2 * p;
}
inline void dm_reduce(double x, double &r, int &q) {
    double kd = std::floor(x * DM_2_OVER_PI + 0.5);
    r = (x - kd * DM_PIO2_HI) - kd * DM_PIO2_LO;
    q = (int)((long long)kd & 3);
}
inline float dm_sinf(float xf) {
    double x = xf;
    if (!(std::fabs(x) < 1.0e9)) return std::numeric_limits<float>::quiet_NaN();
    double r; int q; dm_reduce(x, r, q);
    double v = (q == 0) ? dm_sin_k(r) : (q == 1) ? dm_cos_k(r) : (q == 2) ? -dm_sin_k(r) : -dm_cos_k(r);
    return (float)v;
}
inline float dm_cosf(float xf) {
    double x = xf;
    if (!(std::fabs(x) < 1.0e9)) return std::numeric_limits<float>::quiet_NaN();
    double r; int q; dm_reduce(x, r, q);
    double v = (q == 0) ? dm_cos_k(r) : (q == 1) ? -dm_sin_k(r) : (q == 2) ? -dm_cos_k(r) : dm_sin_k(r);
    return (float)v;
}
static const double DM_ATAN_TAB[9] = {
    0x0.0p+0,               0x1.fd5ba9aac2f6ep-4, 0x1.f5b75f92c80ddp-3, 0x1.6f61941e4def1p-2, 0x1.dac670561bb4fp-2,
    0x1.1e00babdefeb4p-1, 0x1.4978fa3269ee1p-1, 0x1.700a7c5784634p-1, 0x1.921fb54442d18p-1};
inline double dm_atan01(double z) {  // 0 <= z <= 1
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
    return DM_ATAN_TAB[k] + (t + t * (t2 * p));
}
inline double dm_atan_pos(double z) {  // z >= 0 (inf allowed)
    if (z > 1.0) return DM_PIO2_HI - dm_atan01(1.0 / z);
    return dm_atan01(z);
}
inline double dm_atan2d(double y, double x) {
    if (x != x || y != y) return std::numeric_limits<double>::quiet_NaN();
    double ay = std::fabs(y), ax = std::fabs(x);
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ax == 0.0) a = DM_PIO2_HI;
    else if (std::isinf(ax) && std::isinf(ay)) a = DM_PIO2_HI * 0.5;
    else a = dm_atan_pos(ay / ax);
    if (std::signbit(x)) a = DM_PI - a;
    return std::signbit(y) ? -a : a;
}
inline float dm_atan2f(float y, float x) { return (float)dm_atan2d((double)y, (double)x); }
inline float dm_acosf(float xf) {
    double x = xf;
    if (!(x >= -1.0 && x <= 1.0)) return std::numeric_limits<float>::quiet_NaN();
    double s = std::sqrt((1.0 - x) * (1.0 + x));
    return (float)dm_atan2d(s, x);
}
inline double dm_logd_pos(double x) {
    uint64_t bits; std::memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    bits = (bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m; std::memcpy(&m, &bits, 8);  // [1,2)
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
    return (double)e * DM_LN2 + s * p;
}
inline float dm_logf(float xf) {
    if (xf != xf || xf < 0.0f) return std::numeric_limits<float>::quiet_NaN();
    if (xf == 0.0f) return -INF;
    if (std::isinf(xf)) return xf;
    return (float)dm_logd_pos((double)xf);
}
// e^y in f64 (same scheme as dev_math.h: k ln2 + r, degree-14 Taylor series, 2^k scale) -> f32::powf / exp stand-ins
inline double dm_expd(double y) {
    if (y > 700.0) y = 700.0;
    if (y < -700.0) y = -700.0;
    const double kf = std::floor(y * (1.0 / DM_LN2) + 0.5);
    const double r = y - kf * DM_LN2;
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
    double sc; std::memcpy(&sc, &bits, 8);
    return p * sc;
}
inline float dm_powf(float a, float b) {
    if (!(a > 0.0f)) return (a == 0.0f) ? (b == 0.0f ? 1.0f : 0.0f) : std::numeric_limits<float>::quiet_NaN();
    return (float)dm_expd((double)b * dm_logd_pos((double)a));
}


// ---- vectors ---------------------------------------------------------------------------------
struct V3 {
    Float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(Float a, Float b, Float c) : x(a), y(b), z(c) {}
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, Float s) { return V3(a.x * s, a.y * s, a.z * s); }
// vector.rs:481-495: `v / s` is `v * (1/s)`
inline V3 operator/(V3 a, Float s) { Float d = 1.0f / s; return V3(a.x * d, a.y * d, a.z * d); }
inline Float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Float abs_dot(V3 a, V3 b) { return std::fabs(dot(a, b)); }
inline Float length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline Float length(V3 a) { return std::sqrt(length_squared(a)); }
inline V3 normalize(V3 a) { return a / length(a); }
inline V3 vabs(V3 a) { return V3(std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)); }
// vector.rs:339-352: cross product evaluated in f64, rounded once
inline V3 cross(V3 a, V3 b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return V3((Float)((ay * bz) - (az * by)), (Float)((az * bx) - (ax * bz)), (Float)((ax * by) - (ay * bx)));
}
inline int max_dimension(V3 v) { return (v.x > v.y) ? ((v.x > v.z) ? 0 : 2) : ((v.y > v.z) ? 1 : 2); }
inline Float max_component(V3 v) { return fmax_(v.x, fmax_(v.y, v.z)); }
inline V3 permute(V3 v, int x, int y, int z) { return V3(v[x], v[y], v[z]); }
inline V3 face_forward(V3 n, V3 v) { return (dot(n, v) < 0.0f) ? -n : n; }
// vector.rs:551-559
inline void coordinate_system(V3 v1, V3 &v2, V3 &v3) {
    if (std::fabs(v1.x) > std::fabs(v1.y)) v2 = V3(-v1.z, 0.0f, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
    else v2 = V3(0.0f, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
    v3 = cross(v1, v2);
}
inline Float distance_squared(V3 a, V3 b) { return length_squared(a - b); }

struct P2 { Float x, y; P2() : x(0), y(0) {} P2(Float a, Float b) : x(a), y(b) {} Float operator[](int i) const { return i == 0 ? x : y; } };

// ---- RGB spectrum ---------------------------------------------------------------------------
struct RGB {
    Float c[3];
    RGB() { c[0] = c[1] = c[2] = 0; }
    explicit RGB(Float v) { c[0] = c[1] = c[2] = v; }
    RGB(Float r, Float g, Float b) { c[0] = r; c[1] = g; c[2] = b; }
    bool is_black() const { return c[0] == 0.0f && c[1] == 0.0f && c[2] == 0.0f; }
    Float y() const { return 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]; }
    Float max_component_value() const { return fmax_(fmax_(c[0], c[1]), c[2]); }
    bool has_nans() const { return c[0] != c[0] || c[1] != c[1] || c[2] != c[2]; }
    RGB clamps(Float lo, Float hi) const { return RGB(clampv(c[0], lo, hi), clampv(c[1], lo, hi), clampv(c[2], lo, hi)); }
};
inline RGB operator+(RGB a, RGB b) { return RGB(a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2]); }
inline RGB operator-(RGB a, RGB b) { return RGB(a.c[0] - b.c[0], a.c[1] - b.c[1], a.c[2] - b.c[2]); }
inline RGB operator*(RGB a, RGB b) { return RGB(a.c[0] * b.c[0], a.c[1] * b.c[1], a.c[2] * b.c[2]); }
inline RGB operator*(RGB a, Float s) { return RGB(a.c[0] * s, a.c[1] * s, a.c[2] * s); }
inline RGB operator/(RGB a, Float s) { return RGB(a.c[0] / s, a.c[1] / s, a.c[2] / s); }  // true division
inline RGB operator/(RGB a, RGB b) { return RGB(a.c[0] / b.c[0], a.c[1] / b.c[1], a.c[2] / b.c[2]); }
inline RGB &operator+=(RGB &a, RGB b) { a = a + b; return a; }
inline RGB &operator*=(RGB &a, RGB b) { a = a * b; return a; }
inline RGB sqrt_rgb(RGB a) { return RGB(std::sqrt(a.c[0]), std::sqrt(a.c[1]), std::sqrt(a.c[2])); }

// core/spectrum.rs:484-502
inline void xyz_to_rgb(const Float xyz[3], Float rgb[3]) {
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}
inline void rgb_to_xyz(const Float rgb[3], Float xyz[3]) {
    xyz[0] = 0.412453f * rgb[0] + 0.357580f * rgb[1] + 0.180423f * rgb[2];
    xyz[1] = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];
    xyz[2] = 0.019334f * rgb[0] + 0.119193f * rgb[1] + 0.950227f * rgb[2];
}

// ---- 4x4 transforms (row major m[r][c], core/transform.rs) --------------------------------------
struct M4 { Float m[4][4]; };
inline M4 m4_from(const float *p) { M4 r; std::memcpy(r.m, p, 64); return r; }
// transform.rs:413-432
inline V3 xf_point(const M4 &t, V3 p) {
    Float x = p.x, y = p.y, z = p.z;
    Float xp = x * t.m[0][0] + y * t.m[0][1] + z * t.m[0][2] + t.m[0][3];
    Float yp = x * t.m[1][0] + y * t.m[1][1] + z * t.m[1][2] + t.m[1][3];
    Float zp = x * t.m[2][0] + y * t.m[2][1] + z * t.m[2][2] + t.m[2][3];
    Float wp = x * t.m[3][0] + y * t.m[3][1] + z * t.m[3][2] + t.m[3][3];
    if (wp == 1.0f) return V3(xp, yp, zp);
    return V3(xp, yp, zp) / wp;
}
// transform.rs:434-459
inline V3 xf_point_err(const M4 &t, V3 p, V3 &perr) {
    Float x = p.x, y = p.y, z = p.z;
    Float xp = x * t.m[0][0] + y * t.m[0][1] + z * t.m[0][2] + t.m[0][3];
    Float yp = x * t.m[1][0] + y * t.m[1][1] + z * t.m[1][2] + t.m[1][3];
    Float zp = x * t.m[2][0] + y * t.m[2][1] + z * t.m[2][2] + t.m[2][3];
    Float wp = x * t.m[3][0] + y * t.m[3][1] + z * t.m[3][2] + t.m[3][3];
    Float xs = std::fabs(x * t.m[0][0]) + std::fabs(y * t.m[0][1]) + std::fabs(z * t.m[0][2]) + std::fabs(t.m[0][3]);
    Float ys = std::fabs(x * t.m[1][0]) + std::fabs(y * t.m[1][1]) + std::fabs(z * t.m[1][2]) + std::fabs(t.m[1][3]);
    Float zs = std::fabs(x * t.m[2][0]) + std::fabs(y * t.m[2][1]) + std::fabs(z * t.m[2][2]) + std::fabs(t.m[2][3]);
    perr = V3(xs, ys, zs) * gamma(3);
    if (wp == 1.0f) return V3(xp, yp, zp);
    return V3(xp, yp, zp) / wp;
}
// transform.rs:496-508
inline V3 xf_vector(const M4 &t, V3 v) {
    Float x = v.x, y = v.y, z = v.z;
    return V3(x * t.m[0][0] + y * t.m[0][1] + z * t.m[0][2],
              x * t.m[1][0] + y * t.m[1][1] + z * t.m[1][2],
              x * t.m[2][0] + y * t.m[2][1] + z * t.m[2][2]);
}
// transform.rs:529-541: n' = transpose(m_inv) * n ; caller passes the INVERSE matrix
inline V3 xf_normal_inv(const M4 &minv, V3 n) {
    Float x = n.x, y = n.y, z = n.z;
    return V3(x * minv.m[0][0] + y * minv.m[1][0] + z * minv.m[2][0],
              x * minv.m[0][1] + y * minv.m[1][1] + z * minv.m[2][1],
              x * minv.m[0][2] + y * minv.m[1][2] + z * minv.m[2][2]);
}

// transform.rs:461-494
inline V3 xf_point_abs_err(const M4 &t, V3 p, V3 perr, V3 &abs_err) {
    Float x = p.x, y = p.y, z = p.z;
    Float xp = x * t.m[0][0] + y * t.m[0][1] + z * t.m[0][2] + t.m[0][3];
    Float yp = x * t.m[1][0] + y * t.m[1][1] + z * t.m[1][2] + t.m[1][3];
    Float zp = x * t.m[2][0] + y * t.m[2][1] + z * t.m[2][2] + t.m[2][3];
    Float wp = x * t.m[3][0] + y * t.m[3][1] + z * t.m[3][2] + t.m[3][3];
    Float g = gamma(3);
    abs_err.x = (g + 1.0f) * (std::fabs(t.m[0][0]) * perr.x + std::fabs(t.m[0][1]) * perr.y + std::fabs(t.m[0][2]) * perr.z) +
                g * (std::fabs(t.m[0][0] * x) + std::fabs(t.m[0][1] * y) + std::fabs(t.m[0][2] * z) + std::fabs(t.m[0][3]));
    abs_err.y = (g + 1.0f) * (std::fabs(t.m[1][0]) * perr.x + std::fabs(t.m[1][1]) * perr.y + std::fabs(t.m[1][2]) * perr.z) +
                g * (std::fabs(t.m[1][0] * x) + std::fabs(t.m[1][1] * y) + std::fabs(t.m[1][2] * z) + std::fabs(t.m[1][3]));
    abs_err.z = (g + 1.0f) * (std::fabs(t.m[2][0]) * perr.x + std::fabs(t.m[2][1]) * perr.y + std::fabs(t.m[2][2]) * perr.z) +
                g * (std::fabs(t.m[2][0] * x) + std::fabs(t.m[2][1] * y) + std::fabs(t.m[2][2] * z) + std::fabs(t.m[2][3]));
    if (wp == 1.0f) return V3(xp, yp, zp);
    return V3(xp, yp, zp) / wp;
}
inline bool m4_is_identity(const M4 &t) {  // transform.rs:248-253
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) if (t.m[r][c] != ((r == c) ? 1.0f : 0.0f)) return false;
    return true;
}

struct Ray {
    V3 o, d;
    Float t_max;
    Float time;
    uint32_t medium = 0xFFFFFFFFu;   // Ray::medium (index into Scene::media, PT_NONE = vacuum); read by volpath only
    Ray() : t_max(INF), time(0) {}
    Ray(V3 o_, V3 d_, Float t = INF, Float tm = 0.0f) : o(o_), d(d_), t_max(t), time(tm) {}
};
// transform.rs:543-577 (differentials are not carried: constant textures only)
inline Ray xf_ray(const M4 &t, const Ray &r) {
    V3 oerr;
    V3 o = xf_point_err(t, r.o, oerr);
    V3 d = xf_vector(t, r.d);
    Float l2 = length_squared(d);
    Float t_max = r.t_max;
    if (l2 > 0.0f) {
        Float dt = dot(vabs(d), oerr) / l2;
        o = o + d * dt;
        t_max -= dt;
    }
    return Ray(o, d, t_max, r.time);
}

// core/geometry/geometry.rs:6-24
inline V3 offset_ray_origin(V3 p, V3 perr, V3 n, V3 w) {
    Float d = dot(vabs(n), perr);
    V3 offset = n * d;
    if (dot(w, n) < 0.0f) offset = -offset;
    V3 po = p + offset;
    for (int i = 0; i < 3; ++i) {
        if (offset[i] > 0.0f) po[i] = next_float_up(po[i]);
        else if (offset[i] < 0.0f) po[i] = next_float_down(po[i]);
    }
    return po;
}

// core/geometry/geometry.rs:39-54
inline Float spherical_theta(V3 v) { return dm_acosf(clampv(v.z, -1.0f, 1.0f)); }
inline Float spherical_phi(V3 v) {
    Float p = dm_atan2f(v.y, v.x);
    return (p < 0.0f) ? p + 2.0f * PI : p;
}

// core/rng.rs:25-76 PCG32 (test seeds / synthetic scenes only)
struct RNG {
    uint64_t state, inc;
    RNG() : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) {}
    explicit RNG(uint64_t seq) { set_sequence(seq); }
    void set_sequence(uint64_t seq) {
        state = 0; inc = (seq << 1) | 1;
        uniform_u32(); state += 0x853c49e6748fea9bULL; uniform_u32();
    }
    uint32_t uniform_u32() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    uint32_t uniform_u32_bounded(uint32_t b) {
        uint32_t threshold = (~b + 1u) % b;
        for (;;) { uint32_t r = uniform_u32(); if (r >= threshold) return r % b; }
    }
    Float uniform_float() { return fmin_(ONE_MINUS_EPSILON, (Float)uniform_u32() * 0x1.0p-32f); }
};

}  // namespace ref
