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
// `M`/`MI` = VD_C_SOBOL_MATRICES[m-1] / _INV[m-1] (HBM or an LDS copy). Set-bit iteration as above.
PT_DEV uint64_t sobol_interval_to_index(const uint64_t *M, const uint64_t *MI, uint32_t m, uint64_t frame, uint32_t px, uint32_t py) {
    if (m == 0) return 0;
    uint64_t index = frame << (m << 1);
    uint64_t delta = 0;
    while (frame != 0) { const int c = __builtin_ctzll(frame); frame &= frame - 1; delta ^= M[c]; }
    uint64_t b = ((uint64_t)(px << m) | (uint64_t)py) ^ delta;
    while (b != 0) { const int c = __builtin_ctzll(b); b &= b - 1; index ^= MI[c]; }
    return index;
}
PT_DEV uint64_t sobol_interval_to_index(const SobolTables &T, uint32_t m, uint64_t frame, uint32_t px, uint32_t py) {
    if (m == 0) return 0;
    return sobol_interval_to_index(T.vdc + (m - 1) * 52, T.vdc_inv + (m - 1) * 52, m, frame, px, py);
}

// lowdiscrepancy.rs:549-569 with scramble = 0. The reference walks every bit of the index; XOR is
// order-independent, so only the set bits are visited here (count-trailing-zeros loop) -- identical value.
PT_DEV uint32_t sobol_bits(const uint32_t *row, uint64_t a) {
    uint32_t v = 0;
    while (a != 0) { const int i = __builtin_ctzll(a); a &= a - 1; v ^= row[i]; }
    return v;
}
PT_DEV float sobol_to_float(uint32_t v) { return minf((float)v * 0x1.0p-32f, kOneMinusEps); }
PT_DEV float sobol_sample_float(const uint32_t *m32, uint64_t a, uint32_t dim) { return sobol_to_float(sobol_bits(m32 + dim * 52, a)); }

constexpr uint32_t kSobolLdsDims = 64;                   // generator matrices of dimensions 0..63 staged in LDS (13 KB)
constexpr uint32_t kSobolLdsWords = kSobolLdsDims * 52;
// Cooperative copy of the first kSobolLdsDims rows into LDS; caller __syncthreads() afterwards.
PT_DEV void sobol_stage_lds(uint32_t *lds, const uint32_t *m32, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid; i < kSobolLdsWords; i += nthreads) lds[i] = m32[i];
}

// Per-path sampler state: the global Sobol' index of this (pixel, sample) and the running dimension.
struct Sampler {
    uint64_t index;
    uint32_t dim;
    const uint32_t *m32;      // full table in HBM
    const uint32_t *lds;      // first kSobolLdsDims rows in LDS
    bool overflow;
    uint32_t base;            // window of 8 consecutive dimensions evaluated in one pass over the index bits
    uint32_t w0, w1, w2, w3, w4, w5, w6, w7;
    // A path vertex consumes at most 8 dimensions (1 light choice + 2 + 2 + 2 BSDF + 1 roulette, path.rs /
    // integrator.rs:91-101); evaluating them together shares the bit loop and issues 8 independent LDS reads per bit.
    PT_DEV void load_window() {
        base = dim;
        w0 = w1 = w2 = w3 = w4 = w5 = w6 = w7 = 0;
#ifdef PT_ABL_SOBOL   // timing ablation only: a cheap hash instead of the generator-matrix products (results differ)
        { uint32_t h = (uint32_t)index * 0x9E3779B9u ^ (uint32_t)(index >> 32) ^ (base * 0x85EBCA6Bu);
          w0 = h; w1 = h * 3u; w2 = h * 5u; w3 = h * 7u; w4 = h * 11u; w5 = h * 13u; w6 = h * 17u; w7 = h * 19u; return; }
#endif
        if (base + 8 <= kSobolLdsDims) {
            const uint32_t *row = lds + base * 52;
            uint64_t a = index;
            while (a != 0) {
                const int i = __builtin_ctzll(a); a &= a - 1;
                w0 ^= row[i]; w1 ^= row[52 + i]; w2 ^= row[104 + i]; w3 ^= row[156 + i];
                w4 ^= row[208 + i]; w5 ^= row[260 + i]; w6 ^= row[312 + i]; w7 ^= row[364 + i];
            }
        } else base = 0xffffffffu;  // beyond the staged rows: evaluate on demand from HBM
    }
    PT_DEV float sample_dimension(uint32_t d) {  // sobol.rs:68-86 for dim >= 2
        if (d >= 1024) { overflow = true; return 0.0f; }  // the reference panics here
        const uint32_t k = d - base;
        if (base != 0xffffffffu && k < 8u) {
            const uint32_t v = k == 0 ? w0 : k == 1 ? w1 : k == 2 ? w2 : k == 3 ? w3 : k == 4 ? w4 : k == 5 ? w5 : k == 6 ? w6 : w7;
            return sobol_to_float(v);
        }
        return sobol_sample_float(d < kSobolLdsDims ? lds : m32, index, d);
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
