// dev_sampler.h -- Sobol' sampler on device.
//   samplers/sobol.rs:61-86; core/sampler.rs:170-180,322-354 (dimension bookkeeping);
//   core/lowdiscrepancy.rs:512-569 (interval-to-index, sample_float), :399-414 + pbrt_macros:92-111 (radical inverse).
// Tables are DATA (core/sobolmatrices.rs) uploaded from data/sobol_tables.bin.
#pragma once
#include "dev_math.h"

namespace ptd {

struct SobolTables {
    const uint32_t *m32;      // [1024][52]
    const uint64_t *vdc;      // [25][52]
    const uint64_t *vdc_inv;  // [26][52]
};

struct SobolParams {
    int32_t sb_min[2];        // sample_bounds.p_min
    int32_t resolution;       // round_up_pow2(max extent), sobol.rs:42
    int32_t log2_resolution;
};

// lowdiscrepancy.rs:512-543
PT_DEV uint64_t sobol_interval_to_index(const SobolTables &T, uint32_t m, uint64_t frame, uint32_t px, uint32_t py) {
    if (m == 0) return 0;
    const uint64_t *M = T.vdc + (m - 1) * 52, *MI = T.vdc_inv + (m - 1) * 52;
    uint64_t index = frame << (m << 1);
    uint64_t delta = 0;
    for (int c = 0; frame != 0; ++c, frame >>= 1)
        if (frame & 1) delta ^= M[c];
    uint64_t b = ((uint64_t)(px << m) | (uint64_t)py) ^ delta;
    for (int c = 0; b != 0; ++c, b >>= 1)
        if (b & 1) index ^= MI[c];
    return index;
}

// lowdiscrepancy.rs:549-569 with scramble = 0
PT_DEV float sobol_sample_float(const uint32_t *m32, uint64_t a, uint32_t dim) {
    uint32_t v = 0;
    const uint32_t *row = m32 + dim * 52;
    for (int i = 0; a != 0; ++i, a >>= 1)
        if (a & 1) v ^= row[i];
    return minf((float)v * 0x1.0p-32f, kOneMinusEps);
}

// Per-path sampler state: the global Sobol' index of this (pixel, sample) and the running dimension.
struct Sampler {
    uint64_t index;
    uint32_t dim;
    const uint32_t *m32;
    bool overflow;
    PT_DEV float sample_dimension(uint32_t d) {  // sobol.rs:68-86 for dim >= 2
        if (d >= 1024) { overflow = true; return 0.0f; }  // the reference panics here
        return sobol_sample_float(m32, index, d);
    }
    PT_DEV float get_1d() { float r = sample_dimension(dim); dim += 1; return r; }  // sampler.rs:322-333 (array_end_dim == 5)
    PT_DEV P2 get_2d() {                                                             // sampler.rs:336-354
        float y = sample_dimension(dim + 1);
        float x = sample_dimension(dim);
        dim += 2;
        return P2(x, y);
    }
};

// Film-plane dimensions 0/1 are remapped to the pixel (sobol.rs:77-81).
PT_DEV float sobol_pixel_dim(const uint32_t *m32, const SobolParams &sp, uint64_t index, int d, int32_t pixel) {
    float s = sobol_sample_float(m32, index, (uint32_t)d);
    s = s * (float)sp.resolution + (float)sp.sb_min[d];
    return clampf(s - (float)pixel, 0.0f, kOneMinusEps);
}

PT_DEV uint64_t reverse_bits64(uint64_t n) { return __brevll(n); }
PT_DEV float radical_inverse(int base_index, uint64_t n) {  // bases 2,3,5,7,11
    if (base_index == 0) return (float)reverse_bits64(n) * 0x1.0p-64f;
    const uint64_t base = (base_index == 1) ? 3 : (base_index == 2) ? 5 : (base_index == 3) ? 7 : 11;
    float inv_base = 1.0f / (float)base;
    uint64_t rev = 0;
    float inv_base_n = 1.0f;
    while (n != 0) {
        uint64_t next = n / base;
        uint64_t digit = n - next * base;
        rev = rev * base + digit;
        inv_base_n *= inv_base;
        n = next;
    }
    return minf((float)rev * inv_base_n, kOneMinusEps);
}

}  // namespace ptd
