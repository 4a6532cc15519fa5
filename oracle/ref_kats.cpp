// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_math.h header).
// ref_kats.cpp: the reference's own test loops for code this oracle restates, run in C++ for speed and reported to
// tests/test_oracle_kats.py:  tests/fp.rs:125-226 (EFloat abs / sqrt / add / sub / mul / div containment, 10^6 seeds each),
// tests/fp.rs:46-57 (float_bits), tests/bitops.rs:7-64 (log2_int, round_up_pow2), tests/sampling.rs:55-97 (generator matrices, Gray-code samples).
#include "ref_efloat.h"
#include "ref_sampler.h"
#include <cmath>

namespace {
using namespace ref;
struct Rng {   // core/rng.rs:10-75, RNG::new(sequence_index)
    uint64_t state = 0x853c49e6748fea9bull, inc = 0xda3e39cb94b95bdbull;
    Rng() {}
    explicit Rng(uint64_t seq) { state = 0; inc = (seq << 1) | 1; u32(); state += 0x853c49e6748fea9bull; u32(); }
    uint32_t u32() { uint64_t old = state; state = old * 0x5851f42d4c957f2dull + inc; uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27), rot = (uint32_t)(old >> 59); return (xs >> rot) | (xs << ((~rot + 1u) & 31)); }
    uint32_t below(uint32_t b) { uint32_t threshold = (~b + 1u) % b; for (;;) { uint32_t r = u32(); if (r >= threshold) return r % b; } }
    Float f() { return fmin_(ONE_MINUS_EPSILON, (Float)u32() * 0x1.0p-32f); }
};
Float bits_to_float(uint32_t b) { Float f; std::memcpy(&f, &b, 4); return f; }
uint32_t float_to_bits(Float f) { uint32_t b; std::memcpy(&b, &f, 4); return b; }
EFloat get_efloat(Rng &rng, Float min_exp = -6.0f, Float max_exp = 6.0f) {   // tests/fp.rs:73-98
    const Float t = rng.f();
    const Float logu = min_exp * (1.0f - t) + max_exp * t;   // lerp (pbrt.rs:136-144)
    const Float val = std::pow(10.0f, logu);
    Float err = 0.0f;
    switch (rng.below(4)) {
    case 1: { const uint32_t ulp = rng.below(1024); err = std::fabs(bits_to_float(float_to_bits(val) + ulp) - val); break; }
    case 2: { const uint32_t ulp = rng.below(1024 * 1024); err = std::fabs(bits_to_float(float_to_bits(val) + ulp) - val); break; }
    case 3: err = (4.0f * rng.f()) * std::fabs(val); break;
    default: break;
    }
    const Float sign = rng.f() < 0.5f ? -1.0f : 1.0f;
    return EFloat(sign * val, err);
}
double get_precise(const EFloat &ef, Rng &rng) {   // tests/fp.rs:100-114
    switch (rng.below(3)) {
    case 0: return (double)ef.low;
    case 1: return (double)ef.high;
    default: {
        const Float t = rng.f();
        double p = (1.0 - (double)t) * (double)ef.low + (double)t * (double)ef.high;
        if (p > (double)ef.high) p = (double)ef.high;
        if (p < (double)ef.low) p = (double)ef.low;
        return p;
    }
    }
}
inline Float abs_err(const EFloat &e) { return next_float_up(fmax_(std::fabs(e.high - e.v), std::fabs(e.v - e.low))); }   // efloat.rs get_absolute_error
}  // namespace

extern "C" {
// op: 0 abs, 1 sqrt, 2 add, 3 sub, 4 mul, 5 div (tests/fp.rs:125-226). Returns the number of containment violations over
// trials 0..iters-1 (the reference runs 1 000 000); *n_tested = trials not skipped by the test's own preconditions.
int orc_test_efloat(int op, int iters, int *n_tested) {
    int failures = 0, tested = 0;
    for (int trial = 0; trial < iters; ++trial) {
        Rng rng((uint64_t)trial);
        if (op <= 1) {
            const EFloat ef = get_efloat(rng);
            const double precise = get_precise(ef, rng);
            if (op == 1 && abs_err(ef) > 0.25f * std::fabs(ef.low)) continue;
            const EFloat r = op == 0 ? efloat_abs(ef) : efloat_sqrt(efloat_abs(ef));
            const double pr = op == 0 ? std::fabs(precise) : std::sqrt(std::fabs(precise));
            ++tested;
            if (!(pr >= (double)r.low && pr <= (double)r.high)) ++failures;
        } else {
            const EFloat e0 = get_efloat(rng), e1 = get_efloat(rng);
            const double p0 = get_precise(e0, rng), p1 = get_precise(e1, rng);
            if (op == 5 && (e1.low * e1.high < 0.0f || abs_err(e1) > 0.25f * std::fabs(e1.low))) continue;
            const EFloat r = op == 2 ? e0 + e1 : op == 3 ? e0 - e1 : op == 4 ? e0 * e1 : e0 / e1;
            const double pr = op == 2 ? p0 + p1 : op == 3 ? p0 - p1 : op == 4 ? p0 * p1 : p0 / p1;
            ++tested;
            if (!(pr >= (double)r.low && pr <= (double)r.high)) ++failures;
        }
    }
    if (n_tested) *n_tested = tested;
    return failures;
}

// tests/fp.rs:46-57 float_bits: RNG::new(1), `iters` (the reference: 100 000) draws ui; f = bits_to_float(ui); NaNs skipped; float_to_bits(f) == ui -- on
// ref_math.h's float_to_bits / bits_to_float (pbrt.rs:57-78), the pair under next_float_up / next_float_down and offset_ray_origin.
// Returns the number of mismatches; *n_tested = the draws that were not NaN.
int orc_test_float_bits(int iters, int *n_tested) {
    Rng rng(1);
    int failures = 0, tested = 0;
    for (int i = 0; i < iters; ++i) {
        const uint32_t ui = rng.u32();
        const Float f = ref::bits_to_float(ui);
        if (f != f) continue;
        ++tested;
        if (ref::float_to_bits(f) != ui) ++failures;
    }
    if (n_tested) *n_tested = tested;
    return failures;
}

// tests/sampling.rs:24-53 scrambled_radical_inverse_test: for dim < n_dims (the reference: 128), RNG::new(dim), base = PRIMES[dim], the permutation base-1 .. 0 shuffled by
// `shuffle(&mut perm, len, 1, &mut rng)` (sampling.rs:178-186), and the seven indices of the test. The Rust file compares scrambled_radical_inverse with a hand-rolled
// digit loop through a `relative_eq!` whose result it DROPS -- and the hand-rolled loop is broken (`val *= ..` on a zero, `n *= inv_base as u32`): it computes no reference
// value at all. The twin compares the oracle's scrambled_radical_inverse_base (ref_sampler.h, lowdiscrepancy.rs:469-484: what HaltonSampler::sample_dimension calls) with
// the value the radical inverse HAS -- sum_i perm[d_i] b^-(i+1) over the index's digits plus perm[0] for every digit beyond them (perm[0] b^-k / (b - 1)), in exact
// rational arithmetic carried in long double -- to the test's epsilon 1e-5 (relative). Returns the number of violations; *worst = largest relative deviation.
int orc_test_scrambled_radical_inverse(int n_dims, double *worst) {
    const HaltonTables &T = halton_tables();
    static const uint32_t indices[7] = {0u, 1u, 2u, 1151u, 32351u, 4363211u, 681122u};
    int failures = 0; double w = 0.0;
    for (int dim = 0; dim < n_dims; ++dim) {
        Rng rng((uint64_t)dim);
        const uint32_t base = T.primes[dim];
        std::vector<uint16_t> perm(base);
        for (uint32_t i = 0; i < base; ++i) perm[i] = (uint16_t)(base - 1 - i);
        for (uint32_t i = 0; i < base; ++i) { const uint32_t other = i + rng.below(base - i); std::swap(perm[i], perm[other]); }
        for (uint32_t index : indices) {
            long double val = 0.0L, scale = 1.0L / (long double)base; uint32_t n = index;
            while (n > 0) { val += (long double)perm[n % base] * scale; scale /= (long double)base; n /= base; }
            val += (long double)perm[0] * scale * (long double)base / ((long double)base - 1.0L);   // perm[0] * sum_{j >= k+1} b^-j
            const double got = (double)scrambled_radical_inverse_base(base, perm.data(), index);
            const double want = (double)(val < 0.99999994L ? val : 0.99999994L);   // min(.., ONE_MINUS_EPSILON) as the function clamps
            const double rel = std::fabs(got - want) / std::fmax(std::fmax(std::fabs(got), std::fabs(want)), 1e-30);
            if (rel > w) w = rel;
            if (!(rel <= 1.0e-5)) ++failures;
        }
    }
    if (worst) *worst = w;
    return failures;
}

// tests/bitops.rs:7-64 on the oracle's log2_int / round_up_pow2_32 (ref_sampler.h: SobolSampler::new and the MIPMap resampler use
// them); the i64 variants of the reference are the same bit tricks on 64 bits and are restated here only for the test.
int orc_test_bitops(void) {
    int failures = 0;
    auto log2_int64 = [](int64_t v) { return (int64_t)(63 - __builtin_clzll((uint64_t)v)); };
    auto round_up_pow2_64 = [](int64_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v |= v >> 32; return v + 1; };
    for (int i = 0; i < 32; ++i) { const uint32_t ui = 1u << i; failures += log2_int(ui) != i; failures += log2_int64((int64_t)ui) != i; }
    for (int i = 1; i < 32; ++i) { const uint32_t ui = 1u << i; failures += log2_int(ui + 1u) != i; failures += log2_int64((int64_t)ui + 1) != i; }
    for (int i = 0; i < 64; ++i) failures += log2_int64((int64_t)(1ull << i)) != i;
    for (int i = 1; i < 64; ++i) failures += log2_int64((int64_t)(1ull << i) + 1) != i;
    failures += round_up_pow2_32(7) != 8u;
    for (int32_t i = 1; i < (1 << 24); ++i) {
        const bool p2 = i > 0 && !((i & (i - 1)) > 0);
        if (p2) failures += round_up_pow2_32(i) != (uint32_t)i; else failures += round_up_pow2_32(i) != (1u << (log2_int((uint32_t)i) + 1));
        if (p2) failures += round_up_pow2_64(i) != i; else failures += round_up_pow2_64(i) != ((int64_t)1 << (log2_int64(i) + 1));
    }
    for (int i = 0; i < 30; ++i) {
        const int32_t v = 1 << i;
        failures += round_up_pow2_32(v) != (uint32_t)v;
        if (v > 2) failures += round_up_pow2_32(v - 1) != (uint32_t)v;
        failures += round_up_pow2_32(v + 1) != (uint32_t)(2 * v);
    }
    return failures;
}

// tests/sampling.rs:55-83 generator_matrix + :85-97 gray_code_sample_test on multiply_generator (lowdiscrepancy.rs:428-440, the
// column-XOR loop behind sobol_sample_float); reverse_bits32 / sample_generator_matrix / gray_code_sample1d restated next to it.
int orc_test_generator_matrix(void) {
    auto reverse_bits32 = [](uint32_t n) {   // lowdiscrepancy.rs:382-390
        n = (n << 16) | (n >> 16);
        n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
        n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
        n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
        n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
        return n;
    };
    auto sample_generator_matrix = [](const uint32_t *C, uint32_t a, uint32_t scramble) { return fmin_((Float)(multiply_generator(C, a) ^ scramble) * 0x1.0p-32f, ONE_MINUS_EPSILON); };
    int failures = 0;
    uint32_t c[32], crev[32];
    for (int i = 0; i < 32; ++i) { c[i] = 1u << i; crev[i] = reverse_bits32(c[i]); }
    for (uint32_t a = 0; a < 128; ++a) {
        failures += multiply_generator(c, a) != a;
        failures += radical_inverse_base(2, a) != (Float)reverse_bits32(multiply_generator(c, a)) * 2.3283064365386963e-10f;
        failures += radical_inverse_base(2, a) != sample_generator_matrix(crev, a, 0);
    }
    Rng rng;   // RNG::default()
    for (int i = 0; i < 32; ++i) { c[i] = rng.u32(); crev[i] = reverse_bits32(c[i]); }
    for (uint32_t a = 0; a < 1024; ++a) failures += reverse_bits32(multiply_generator(c, a)) != multiply_generator(crev, a);
    // gray_code_sample_test: the 64 Gray-code samples of the identity matrix are the 64 values multiply_generator produces
    for (int i = 0; i < 32; ++i) c[i] = 1u << i;
    Float v[64]; uint32_t acc = 0;
    for (int i = 0; i < 64; ++i) { v[i] = fmin_((Float)acc * 0x1.0p-32f, ONE_MINUS_EPSILON); acc ^= c[__builtin_ctz((uint32_t)(i + 1))]; }   // lowdiscrepancy.rs:444-451
    for (uint32_t a = 0; a < 64; ++a) {
        const Float u = (Float)multiply_generator(c, a) * 2.3283064365386963e-10f;
        bool found = false; for (int i = 0; i < 64; ++i) found = found || v[i] == u;
        failures += !found;
    }
    return failures;
}
}  // extern "C"
