// ORACLE -- TEST INFRASTRUCTURE ONLY (see ref_math.h header).
// ref_sampler.h: SobolSampler + GlobalSampler "base class" macros.
//   samplers/sobol.rs:14-117; core/sampler.rs:12,170-180,256-354; core/lowdiscrepancy.rs:512-569.
// Tables are DATA extracted by tools/extract_sobol_tables.py (core/sobolmatrices.rs).
#pragma once
#include "ref_math.h"
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

// core/lowdiscrepancy.rs:549-569
inline Float sobol_sample_float(uint64_t a, int dimension, uint32_t scramble) {
    const SobolTables &T = sobol_tables();
    uint32_t v = scramble;
    int i = dimension * SOBOL_MATRIX_SIZE;
    while (a != 0) {
        if (a & 1) v ^= T.m32[i];
        i += 1; a >>= 1;
    }
    return fmin_((Float)v * 0x1.0p-32f, ONE_MINUS_EPSILON);
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

    SobolSampler(uint64_t spp_, const int32_t sb[4]) {
        sb_min[0] = sb[0]; sb_min[1] = sb[1]; sb_max[0] = sb[2]; sb_max[1] = sb[3];
        int32_t dx = sb[2] - sb[0], dy = sb[3] - sb[1];
        resolution = (int32_t)round_up_pow2_32(std::max(dx, dy));   // sobol.rs:42-44
        log2_resolution = log2_int((uint32_t)resolution);
        spp = spp_; cur_pixel[0] = cur_pixel[1] = 0; cur_sample = 0; dimension = 0;
        interval_sample_index = 0; array_end_dim = 5; dim_overflow = false;
    }
    uint64_t get_index_for_sample(uint64_t n) const {  // sobol.rs:61-66
        return sobol_interval_to_index((uint32_t)log2_resolution, n, cur_pixel[0] - sb_min[0], cur_pixel[1] - sb_min[1]);
    }
    Float sample_dimension(uint64_t index, int dim) {  // sobol.rs:68-86
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
