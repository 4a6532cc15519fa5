// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_math.h header).
// ref_efloat.h: core/efloat.rs -- the running-error-bound float of Sphere::intersect (shapes/sphere.rs:59-152), shared by
// ref_sphere.cpp and the restated reference tests of ref_kats.cpp (tests/fp.rs:125-226).
#pragma once
#include "ref_math.h"

namespace ref {

// core/efloat.rs
struct EFloat {
    Float v, low, high;
    EFloat() : v(0), low(0), high(0) {}
    EFloat(Float v_, Float err) : v(v_) {
        if (err == 0.0f) { low = v; high = v; }
        else { low = next_float_down(v - err); high = next_float_up(v + err); }
    }
    explicit EFloat(Float f) : v(f), low(f), high(f) {}
};
inline EFloat operator+(EFloat a, EFloat b) { EFloat r; r.v = a.v + b.v; r.low = next_float_down(a.low + b.low); r.high = next_float_up(a.high + b.high); return r; }
inline EFloat operator-(EFloat a, EFloat b) { EFloat r; r.v = a.v - b.v; r.low = next_float_down(a.low - b.high); r.high = next_float_up(a.high - b.low); return r; }
inline EFloat operator*(EFloat a, EFloat b) {
    EFloat r; r.v = a.v * b.v;
    Float p[4] = {a.low * b.low, a.high * b.low, a.low * b.high, a.high * b.high};
    r.low = next_float_down(fmin_(fmin_(p[0], p[1]), fmin_(p[2], p[3])));
    r.high = next_float_up(fmax_(fmax_(p[0], p[1]), fmax_(p[2], p[3])));
    return r;
}
inline EFloat operator/(EFloat a, EFloat b) {  // efloat.rs:124-146: the straddle test looks at the NUMERATOR (as written there)
    EFloat r; r.v = a.v / b.v;
    if (a.low < 0.0f && a.high > 0.0f) { r.low = -INF; r.high = INF; }
    else {
        Float d[4] = {a.low / b.low, a.high / b.low, a.low / b.high, a.high / b.high};
        r.low = next_float_down(fmin_(fmin_(d[0], d[1]), fmin_(d[2], d[3])));
        r.high = next_float_up(fmax_(fmax_(d[0], d[1]), fmax_(d[2], d[3])));
    }
    return r;
}
inline bool quadratic(EFloat a, EFloat b, EFloat c, EFloat &t0, EFloat &t1) {  // efloat.rs:211-231
    double discrim = (double)b.v * (double)b.v - 4.0 * (double)a.v * (double)c.v;
    if (discrim < 0.0) return false;
    double root = std::sqrt(discrim);
    EFloat frd((Float)root, (Float)((double)MACHINE_EPSILON * root));
    EFloat q = (b.v < 0.0f) ? (EFloat(-0.5f) * (b - frd)) : (EFloat(-0.5f) * (b + frd));  // Mul<Float>: from(f) * self
    t0 = q / a; t1 = c / q;
    if (t0.v > t1.v) std::swap(t0, t1);
    return true;
}

// efloat.rs:37-68 (used by the reference's tests only; restated so that tests/fp.rs:125-157 can be pinned)
inline EFloat efloat_sqrt(EFloat a) { EFloat r; r.v = std::sqrt(a.v); r.low = next_float_down(std::sqrt(a.low)); r.high = next_float_up(std::sqrt(a.high)); return r; }
inline EFloat efloat_abs(EFloat a) {
    if (a.low >= 0.0f) return a;
    EFloat r;
    if (a.high <= 0.0f) { r.v = -a.v; r.low = -a.high; r.high = -a.low; }
    else { r.v = std::fabs(a.v); r.low = 0.0f; r.high = fmax_(-a.low, a.high); }
    return r;
}

}  // namespace ref
