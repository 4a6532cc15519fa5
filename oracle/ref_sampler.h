// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_math.h header).
// ref_sampler.h: SobolSampler + GlobalSampler "base class" macros.
//   samplers/sobol.rs:14-117; core/sampler.rs:12,170-180,256-354; core/lowdiscrepancy.rs:512-569.
// Tables are DATA extracted by tools/extract_sobol_tables.py (core/sobolmatrices.rs).
#pragma once
#include "ref_math.h"
#include "../include/mi355pt.h"
#include <algorithm>
#include <cstdio>
#include <vector>
#include <string>
#include <stdexcept>

namespace ref {

struct SobolTables {
    std::vector<uint32_t> m32;   // 1024*52
    std::vector<uint64_t> vdc;   // 25*52
    std::vector<uint64_t> vdc_inv;  // 26*52
    bool load(const char *path) {
        FILE *f = std::fopen(path, "rb");
        if (!f) return false;
        char magic[8];
        m32.resize(1024 * 52); vdc.resize(25 * 52); vdc_inv.resize(26 * 52);
        bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "PTSOBOL1", 8) == 0 &&
                  std::fread(m32.data(), 4, m32.size(), f) == m32.size() &&
                  std::fread(vdc.data(), 8, vdc.size(), f) == vdc.size() &&
                  std::fread(vdc_inv.data(), 8, vdc_inv.size(), f) == vdc_inv.size();
        std::fclose(f);
        return ok;
    }
};
inline SobolTables &sobol_tables() { static SobolTables t; return t; }

static const int NUM_SOBOL_DIMENSIONS = 1024;
static const int SOBOL_MATRIX_SIZE = 52;

inline uint32_t round_up_pow2_32(int32_t v) {  // core/pbrt.rs round_up_pow2_32
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return (uint32_t)(v + 1);
}
inline int log2_int(uint32_t v) { return 31 - __builtin_clz(v); }

// core/lowdiscrepancy.rs:512-543
inline uint64_t sobol_interval_to_index(uint32_t m, uint64_t frame, int64_t px, int64_t py) {
    if (m == 0) return 0;
    const SobolTables &T = sobol_tables();
    uint32_t m2 = m << 1;
    uint64_t index = frame << m2;
    uint64_t delta = 0;
    int c = 0;
    while (frame != 0) {
        if (frame & 1) delta ^= T.vdc[(m - 1) * 52 + c];
        c += 1; frame >>= 1;
    }
    uint64_t b = ((uint64_t)(((uint32_t)px) << m) | (uint64_t)py) ^ delta;
    c = 0;
    while (b != 0) {
        if (b & 1) index ^= T.vdc_inv[(m - 1) * 52 + c];
        c += 1; b >>= 1;
    }
    return index;
}

// core/lowdiscrepancy.rs:428-440 multiply_generator: the XOR of the generator-matrix columns selected by the bits of `a` (the loop
// sobol_sample_float, :549-569, runs over its 52-column matrices; pinned by tests/sampling.rs:55-83 through ref_kats.cpp)
inline uint32_t multiply_generator(const uint32_t *C, uint64_t a) {
    uint32_t v = 0;
    int i = 0;
    while (a != 0) {
        if (a & 1) v ^= C[i];
        i += 1; a >>= 1;
    }
    return v;
}
// core/lowdiscrepancy.rs:549-569
inline Float sobol_sample_float(uint64_t a, int dimension, uint32_t scramble) {
    const SobolTables &T = sobol_tables();
    const uint32_t v = scramble ^ multiply_generator(T.m32.data() + dimension * SOBOL_MATRIX_SIZE, a);
    return fmin_((Float)v * 0x1.0p-32f, ONE_MINUS_EPSILON);
}

// ---- HaltonSampler (samplers/halton.rs) -- parity unpinned by the reference's tests except radical_inverse(0, a)
// == reverse_bits32(a) * 2^-32 for a < 1024 (tests/sampling.rs:16-21), which tests/test_halton.py checks.
struct HaltonTables {   // PRIMES / PRIME_SUMS (lowdiscrepancy.rs:9-192) and compute_radical_inverse_permutations (:359-378)
    std::vector<uint32_t> primes, sums;
    std::vector<uint16_t> perm;
    struct Pcg32 {      // rng.rs:17-58
        uint64_t state = 0x853c49e6748fea9bull, inc = 0xda3e39cb94b95bdbull;
        uint32_t uniform_int32() {
            uint64_t old = state;
            state = old * 0x5851f42d4c957f2dull + inc;
            uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
            uint32_t rot = (uint32_t)(old >> 59);
            return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
        }
        uint32_t uniform_int32_2(uint32_t b) {
            uint32_t threshold = (~b + 1u) % b;
            for (;;) { uint32_t r = uniform_int32(); if (r >= threshold) return r % b; }
        }
    };
    HaltonTables() {
        for (uint32_t c = 2; primes.size() < 1000; ++c) {
            bool is_prime = true;
            for (uint32_t d = 2; d * d <= c; ++d) if (c % d == 0) { is_prime = false; break; }
            if (is_prime) primes.push_back(c);
        }
        uint32_t total = 0;
        for (uint32_t q : primes) { sums.push_back(total); total += q; }
        perm.resize(total);
        Pcg32 rng;   // RNG::default()
        uint32_t p = 0;
        for (size_t i = 0; i < primes.size(); ++i) {
            for (uint32_t j = 0; j < primes[i]; ++j) perm[p + j] = (uint16_t)j;
            for (uint32_t j = 0; j < primes[i]; ++j) {   // shuffle(.., count = PRIMES[i], 1, rng) (sampling.rs:178-186)
                uint32_t other = j + rng.uniform_int32_2(primes[i] - j);
                std::swap(perm[p + j], perm[p + other]);
            }
            p += primes[i];
        }
    }
};
inline const HaltonTables &halton_tables() { static HaltonTables t; return t; }

inline int64_t mod_i(int64_t a, int64_t b) { int64_t r = a - (a / b) * b; return r < 0 ? r + b : r; }   // pbrt.rs mod_
inline void extended_gcd(int64_t a, int64_t b, int64_t &x, int64_t &y) {   // halton.rs:19-29
    if (b == 0) { x = 1; y = 0; return; }
    int64_t d = a / b, xp, yp;
    extended_gcd(b, a % b, xp, yp);
    x = yp; y = xp - d * yp;
}
inline int64_t multiplicative_inverse(int64_t a, int64_t n) { int64_t x, y; extended_gcd(a, n, x, y); return mod_i(x, n); }   // halton.rs:31-35
inline uint64_t inverse_radical_inverse(uint64_t base, uint64_t inverse, uint64_t ndigits) {   // lowdiscrepancy.rs:416-426
    uint64_t index = 0;
    for (uint64_t i = 0; i < ndigits; ++i) { uint64_t digit = inverse % base; inverse /= base; index = index * base + digit; }
    return index;
}
inline uint64_t reverse_bits64_h(uint64_t n) { uint64_t r = 0; for (int i = 0; i < 64; ++i) { r = (r << 1) | (n & 1); n >>= 1; } return r; }
inline Float radical_inverse_base(uint64_t base, uint64_t n) {   // radical_inverse_specialized (lowdiscrepancy.rs:399-414)
    Float inv_base = 1.0f / (Float)base, inv_base_n = 1.0f;
    uint64_t rev = 0;
    while (n != 0) { uint64_t next = n / base, digit = n - next * base; rev = rev * base + digit; inv_base_n *= inv_base; n = next; }
    return fmin_((Float)rev * inv_base_n, ONE_MINUS_EPSILON);
}
inline Float scrambled_radical_inverse_base(uint64_t base, const uint16_t *perm, uint64_t a) {   // lowdiscrepancy.rs:469-484
    Float inv_base = 1.0f / (Float)base, inv_base_n = 1.0f;
    uint64_t rev = 0;
    while (a != 0) { uint64_t next = a / base, digit = a - next * base; rev = rev * base + perm[digit]; inv_base_n *= inv_base; a = next; }
    Float res = inv_base_n * ((Float)rev + inv_base * (Float)perm[0] / (1.0f - inv_base));
    return fmin_(res, ONE_MINUS_EPSILON);
}

struct CameraSample { P2 pfilm; P2 plens; Float time; };

struct SobolSampler {
    int64_t sb_min[2], sb_max[2];   // sample_bounds
    int32_t resolution, log2_resolution;
    uint64_t spp;
    // per-pixel state
    int64_t cur_pixel[2];
    uint64_t cur_sample;
    int dimension;
    uint64_t interval_sample_index;
    int array_end_dim;  // == ARRAY_START_DIM (5): the path integrator requests no arrays
    bool dim_overflow;
    // HaltonSampler state (halton.rs:54-60); `halton` selects it: the GlobalSampler bookkeeping below is shared
    bool halton = false, at_center = false;
    int64_t base_scales[2] = {1, 1}, base_exponents[2] = {0, 0}, mult_inverse[2] = {0, 0};
    uint64_t sample_stride = 1;

    SobolSampler(uint64_t spp_, const int32_t sb[4], uint32_t sampler_type = PT_SAMPLER_SOBOL, bool sample_at_pixel_center = false) {
        if (sampler_type == PT_SAMPLER_HALTON) {   // HaltonSampler::new (halton.rs:62-110)
            halton = true; at_center = sample_at_pixel_center;
            const int64_t res[2] = {sb[2] - sb[0], sb[3] - sb[1]};
            for (int i = 0; i < 2; ++i) {
                int64_t base = i == 0 ? 2 : 3, scale = 1, e = 0;
                while (scale < std::min<int64_t>(res[i], 128)) { scale *= base; e += 1; }
                base_scales[i] = scale; base_exponents[i] = e;
            }
            sample_stride = (uint64_t)(base_scales[0] * base_scales[1]);
            mult_inverse[0] = multiplicative_inverse(base_scales[1], base_scales[0]);
            mult_inverse[1] = multiplicative_inverse(base_scales[0], base_scales[1]);
        }
        sb_min[0] = sb[0]; sb_min[1] = sb[1]; sb_max[0] = sb[2]; sb_max[1] = sb[3];
        int32_t dx = sb[2] - sb[0], dy = sb[3] - sb[1];
        resolution = (int32_t)round_up_pow2_32(std::max(dx, dy));   // sobol.rs:42-44
        log2_resolution = log2_int((uint32_t)resolution);
        spp = spp_; cur_pixel[0] = cur_pixel[1] = 0; cur_sample = 0; dimension = 0;
        interval_sample_index = 0; array_end_dim = 5; dim_overflow = false;
    }
    uint64_t get_index_for_sample(uint64_t n) const {  // sobol.rs:61-66 / halton.rs:122-155
        if (halton) {
            uint64_t offset = 0;
            if (sample_stride > 1) {
                const int64_t pm[2] = {mod_i(cur_pixel[0], 128), mod_i(cur_pixel[1], 128)};
                for (int i = 0; i < 2; ++i) {
                    uint64_t dim_offset = inverse_radical_inverse(i == 0 ? 2 : 3, (uint64_t)pm[i], (uint64_t)base_exponents[i]);
                    offset += dim_offset * (sample_stride / (uint64_t)base_scales[i]) * (uint64_t)mult_inverse[i];
                }
                offset %= sample_stride;
            }
            return offset + n * sample_stride;
        }
        return sobol_interval_to_index((uint32_t)log2_resolution, n, cur_pixel[0] - sb_min[0], cur_pixel[1] - sb_min[1]);
    }
    Float sample_dimension(uint64_t index, int dim) {  // sobol.rs:68-86 / halton.rs:157-165
        if (halton) {
            if (at_center && (dim == 0 || dim == 1)) return 0.5f;
            if (dim >= 1000) { dim_overflow = true; return 0.0f; }  // permutation_for_dimension panics (halton.rs:113-119)
            if (dim == 0) return (Float)reverse_bits64_h(index >> (uint64_t)base_exponents[0]) * 0x1.0p-64f;   // pbrt_macros:101, no clamp
            if (dim == 1) return radical_inverse_base(3, index / (uint64_t)base_scales[1]);
            const HaltonTables &T = halton_tables();
            return scrambled_radical_inverse_base(T.primes[dim], T.perm.data() + T.sums[dim], index);
        }
        if (dim >= NUM_SOBOL_DIMENSIONS) { dim_overflow = true; return 0.0f; }  // reference panics
        Float s = sobol_sample_float(index, dim, 0);
        if (dim == 0 || dim == 1) {
            s = s * (Float)resolution + (Float)sb_min[dim];
            s = clampv(s - (Float)cur_pixel[dim], 0.0f, ONE_MINUS_EPSILON);
        }
        return s;
    }
    void start_pixel(int64_t x, int64_t y) {  // sampler.rs:268-308
        cur_pixel[0] = x; cur_pixel[1] = y; cur_sample = 0; dimension = 0;
        interval_sample_index = get_index_for_sample(0);
    }
    bool start_next_sample() {  // sampler.rs:256-265
        dimension = 0;
        interval_sample_index = get_index_for_sample(cur_sample + 1);
        cur_sample += 1;
        return cur_sample < spp;
    }
    bool set_sample_number(uint64_t n) {
        dimension = 0;
        interval_sample_index = get_index_for_sample(n);
        cur_sample = n;
        return cur_sample < spp;
    }
    Float get_1d() {  // sampler.rs:322-333
        if (dimension >= 5 && dimension < array_end_dim) dimension = array_end_dim;
        Float r = sample_dimension(interval_sample_index, dimension);
        dimension += 1;
        return r;
    }
    P2 get_2d() {  // sampler.rs:336-354 (y evaluated first, assigned to dim+1)
        if (dimension + 1 >= 5 && dimension < array_end_dim) dimension = array_end_dim;
        Float y = sample_dimension(interval_sample_index, dimension + 1);
        Float x = sample_dimension(interval_sample_index, dimension);
        dimension += 2;
        return P2(x, y);
    }
    CameraSample get_camera_sample(int64_t px, int64_t py) {  // sampler.rs:170-180
        CameraSample cs;
        P2 u = get_2d();
        cs.pfilm = P2((Float)px + u.x, (Float)py + u.y);
        cs.time = get_1d();
        cs.plens = get_2d();
        return cs;
    }
};

// core/lowdiscrepancy.rs:399-414 + pbrt_macros/src/lib.rs:92-111 (bases 2,3,5,7,11)
inline uint64_t reverse_bits64(uint64_t n) {
    uint64_t r = 0;
    for (int i = 0; i < 64; ++i) { r = (r << 1) | (n & 1); n >>= 1; }
    return r;
}
inline Float radical_inverse(int base_index, uint64_t n) {
    static const int PRIMES[5] = {2, 3, 5, 7, 11};
    if (base_index == 0) return (Float)reverse_bits64(n) * 0x1.0p-64f;  // no clamp (App. A #28)
    uint64_t base = (uint64_t)PRIMES[base_index];
    Float inv_base = 1.0f / (Float)base;
    uint64_t rev = 0;
    Float inv_base_n = 1.0f;
    while (n != 0) {
        uint64_t next = n / base;
        uint64_t digit = n - next * base;
        rev = rev * base + digit;
        inv_base_n *= inv_base;
        n = next;
    }
    return fmin_((Float)rev * inv_base_n, ONE_MINUS_EPSILON);
}

}  // namespace ref
