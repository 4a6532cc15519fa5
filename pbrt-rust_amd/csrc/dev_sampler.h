// dev_sampler.h -- Sobol' and Halton samplers on device (both are GlobalSamplers: one global index per (pixel, sample),
// one value per dimension; only get_index_for_sample / sample_dimension differ).
//   samplers/halton.rs:19-35,62-110,112-165; core/lowdiscrepancy.rs:359-426,469-484 (Halton)
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
    // Halton: the first 1000 primes, their running sums and the digit permutations (compute_radical_inverse_permutations
    // with the default RNG, lowdiscrepancy.rs:359-378), built once on the host by host_device.hip
    const uint32_t *prime;     // [1000]
    const uint32_t *prime_sum; // [1000]
    const uint16_t *perm;      // [sum of the first 1000 primes]
    // m32 folded by index nibble (built by host_device.hip from m32): nib[j][d][n] = XOR of m32[d][4 j + b] over the set bits b of n, j < kSobolNibbles.
    // A Sobol' value is then one 16-entry table look-up per index nibble instead of one step per set index bit -- the same XORs, regrouped.
    const uint32_t *nib;       // [kSobolNibbles][1024][16]
};
constexpr uint32_t kSobolNibbles = 10;                       // index bits 0..39 through the nibble tables, the rest bit by bit from m32
constexpr uint32_t kSobolNibWords = kSobolNibbles * 16;      // per dimension (LDS budget: 640 B a dimension)

struct HaltonParams {          // HaltonSampler::new (halton.rs:62-110)
    uint32_t enabled;          // PtRenderParams.sampler_type == PT_SAMPLER_HALTON
    uint32_t base_scale[2], base_exp[2];
    uint32_t stride;           // sample_stride = base_scale[0] * base_scale[1]
    uint32_t mult_inv[2];
    uint32_t at_center;        // "samplepixelcenter"
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

// One dimension through the nibble tables: `t` = the dimension's first table (LDS or HBM), the next nibble's `jstride` words on;
// `row` = the dimension's m32 row for index bits >= 40.
PT_DEV uint32_t sobol_bits_nib(const uint32_t *t, uint32_t jstride, const uint32_t *row, uint64_t a) {
    const uint32_t lo = (uint32_t)a, hi = (uint32_t)(a >> 32);
    uint32_t v = t[lo & 15u] ^ t[jstride + ((lo >> 4) & 15u)] ^ t[2 * jstride + ((lo >> 8) & 15u)] ^ t[3 * jstride + ((lo >> 12) & 15u)]
               ^ t[4 * jstride + ((lo >> 16) & 15u)] ^ t[5 * jstride + ((lo >> 20) & 15u)] ^ t[6 * jstride + ((lo >> 24) & 15u)] ^ t[7 * jstride + (lo >> 28)];
    if (hi != 0) {
        v ^= t[8 * jstride + (hi & 15u)] ^ t[9 * jstride + ((hi >> 4) & 15u)];
        for (uint32_t r = hi >> 8; r != 0; r &= r - 1) v ^= row[40 + __builtin_ctz(r)];
    }
    return v;
}
// The same from the HBM tables as a real function: the one-dimension-at-a-time path of Sampler::sample_dimension (dimensions outside the
// vertex's window: ratio / delta tracking in grid media, the tail of very deep paths) is rare, and inlined at every get_1d / get_2d it
// only spreads the hot code of the large shade kernels over more instruction-cache lines.
__device__ __noinline__ inline uint32_t sobol_bits_far(const uint32_t *nib, const uint32_t *m32, uint32_t d, uint64_t a) {
    return sobol_bits_nib(nib + d * 16u, 1024u * 16u, m32 + d * 52u, a);
}
// Cooperative copy of the first `dims` dimensions' nibble tables into LDS, [nibble][dimension][16] (a 16-entry table spans 16
// banks, so the per-lane look-ups never conflict); caller __syncthreads() afterwards.
PT_DEV void sobol_stage_lds(uint32_t *lds, const uint32_t *nib, uint32_t dims, uint32_t tid, uint32_t nthreads) {
    const uint4 *src = reinterpret_cast<const uint4 *>(nib); uint4 *dst = reinterpret_cast<uint4 *>(lds);
    for (uint32_t i = tid; i < kSobolNibbles * dims * 4u; i += nthreads) { const uint32_t j = i / (dims * 4u), r = i - j * (dims * 4u); dst[i] = src[j * (1024u * 4u) + r]; }
}

constexpr uint32_t kHaltonMaxDims = 1000;  // PRIME_TABLE_SIZE: permutation_for_dimension panics beyond (halton.rs:113-119)

// inverse_radical_inverse::<BASE> (lowdiscrepancy.rs:416-426)
PT_DEV uint64_t halton_inverse_radical_inverse(uint32_t base, uint64_t inverse, uint32_t ndigits) {
    uint64_t index = 0;
    for (uint32_t i = 0; i < ndigits; ++i) { const uint64_t digit = inverse % base; inverse /= base; index = index * base + digit; }
    return index;
}
// HaltonSampler::get_index_for_sample (halton.rs:122-155); mod_ is the non-negative remainder (pbrt.rs)
PT_DEV uint64_t halton_index_for_sample(const HaltonParams &hp, int32_t px, int32_t py, uint64_t sample_num) {
    uint64_t offset = 0;
    if (hp.stride > 1) {
        const int32_t pm[2] = {((px % 128) + 128) % 128, ((py % 128) + 128) % 128};
        for (int i = 0; i < 2; ++i) {
            const uint64_t dim_offset = halton_inverse_radical_inverse(i == 0 ? 2u : 3u, (uint64_t)pm[i], hp.base_exp[i]);
            offset += dim_offset * (uint64_t)(hp.stride / hp.base_scale[i]) * (uint64_t)hp.mult_inv[i];
        }
        offset %= (uint64_t)hp.stride;
    }
    return offset + sample_num * (uint64_t)hp.stride;
}
// radical_inverse_specialized::<BASE> (lowdiscrepancy.rs:399-414); 32-bit digits loop when the value fits (same digits)
PT_DEV float halton_radical_inverse(uint32_t base, uint64_t n) {
    const float inv_base = 1.0f / (float)base;
    uint64_t rev = 0; float inv_base_n = 1.0f;
    if ((n >> 32) == 0) { uint32_t a = (uint32_t)n; while (a != 0) { const uint32_t next = a / base; rev = rev * base + (a - next * base); inv_base_n *= inv_base; a = next; } }
    else while (n != 0) { const uint64_t next = n / base; rev = rev * base + (n - next * base); inv_base_n *= inv_base; n = next; }
    return minf((float)rev * inv_base_n, kOneMinusEps);
}
// scranmbled_radical_inverse_specialized::<BASE> (lowdiscrepancy.rs:469-484)
PT_DEV float halton_scrambled_radical_inverse(uint32_t base, const uint16_t *perm, uint64_t n) {
    const float inv_base = 1.0f / (float)base;
    uint64_t rev = 0; float inv_base_n = 1.0f;
    if ((n >> 32) == 0) { uint32_t a = (uint32_t)n; while (a != 0) { const uint32_t next = a / base; rev = rev * base + perm[a - next * base]; inv_base_n *= inv_base; a = next; } }
    else while (n != 0) { const uint64_t next = n / base; rev = rev * base + perm[n - next * base]; inv_base_n *= inv_base; n = next; }
    return minf(inv_base_n * ((float)rev + inv_base * (float)perm[0] / (1.0f - inv_base)), kOneMinusEps);
}
// HaltonSampler::sample_dimension (halton.rs:157-165); dim < kHaltonMaxDims
PT_DEV float halton_sample_dimension(const SobolTables &T, const HaltonParams &hp, uint64_t index, uint32_t dim) {
    if (hp.at_center && dim < 2) return 0.5f;
    if (dim == 0) return (float)__brevll(index >> hp.base_exp[0]) * 0x1.0p-64f;   // radical_inverse(0, ..): no clamp (pbrt_macros:101)
    if (dim == 1) return halton_radical_inverse(3u, index / hp.base_scale[1]);
    return halton_scrambled_radical_inverse(T.prime[dim], T.perm + T.prime_sum[dim], index);
}

// Per-path sampler state: the global sample index of this (pixel, sample) and the running dimension.
struct Sampler {
    uint64_t index;
    uint32_t dim;
    const uint32_t *m32;      // full table in HBM (index bits >= 40)
    const uint32_t *nib;      // nibble tables of every dimension in HBM
    const uint32_t *lds;      // nibble tables of the first lds_dims dimensions in LDS
    uint32_t lds_dims;
    bool overflow;
    bool halton;              // wave-uniform: Halton instead of Sobol' (the window then holds float bits)
    const uint32_t *prime, *prime_sum; const uint16_t *perm;
    uint32_t base;            // window of 8 (or 3) consecutive dimensions evaluated together
    uint32_t wn;              // how many of them
    uint32_t w0, w1, w2, w3, w4, w5, w6, w7;
    // A path vertex consumes at most 8 dimensions (1 light choice + 2 + 2 + 2 BSDF + 1 roulette, path.rs /
    // integrator.rs:91-101); evaluating them together shares the nibble addresses: 8 independent LDS reads per index nibble.
    PT_DEV void load_window() {
        base = dim; wn = 8;
        w0 = w1 = w2 = w3 = w4 = w5 = w6 = w7 = 0;
        if (halton) {
            if (base + 8 > kHaltonMaxDims) { base = 0xffffffffu; return; }
            uint32_t w[8];
            for (int k = 0; k < 8; ++k) w[k] = __float_as_uint(halton_scrambled_radical_inverse(prime[base + k], perm + prime_sum[base + k], index));
            w0 = w[0]; w1 = w[1]; w2 = w[2]; w3 = w[3]; w4 = w[4]; w5 = w[5]; w6 = w[6]; w7 = w[7];
            return;
        }
        if (base + 8 <= lds_dims) nib_window(lds + base * 16u, lds_dims * 16u);
        else if (base + 8 <= 1024u) nib_window(nib + base * 16u, 1024u * 16u);
        else base = 0xffffffffu;  // the last dimensions: one at a time
    }
    // the eight dimensions' tables of one nibble are 16 words apart: one address per index nibble, eight reads at constant offsets from it
    PT_DEV void nib_window(const uint32_t *t, uint32_t jstride) {
        const uint32_t lo = (uint32_t)index, hi = (uint32_t)(index >> 32);
#define PT_NIB2(j, xa, xb) { const uint32_t *qa = t + (j) * jstride + ((xa) & 15u), *qb = t + ((j) + 1) * jstride + ((xb) & 15u); \
                             const uint32_t a0 = qa[0], a1 = qa[16], a2 = qa[32], a3 = qa[48], a4 = qa[64], a5 = qa[80], a6 = qa[96], a7 = qa[112]; \
                             const uint32_t b0 = qb[0], b1 = qb[16], b2 = qb[32], b3 = qb[48], b4 = qb[64], b5 = qb[80], b6 = qb[96], b7 = qb[112]; \
                             w0 ^= a0 ^ b0; w1 ^= a1 ^ b1; w2 ^= a2 ^ b2; w3 ^= a3 ^ b3; w4 ^= a4 ^ b4; w5 ^= a5 ^ b5; w6 ^= a6 ^ b6; w7 ^= a7 ^ b7; }
        PT_NIB2(0, lo, lo >> 4) PT_NIB2(2, lo >> 8, lo >> 12) PT_NIB2(4, lo >> 16, lo >> 20) PT_NIB2(6, lo >> 24, lo >> 28)   // (two bursts of 16 reads instead of one of 32: no difference)
        if (hi != 0) {
            PT_NIB2(8, hi, hi >> 4)
            for (uint32_t r = hi >> 8; r != 0; r &= r - 1) {
                const uint32_t *row = m32 + base * 52 + 40 + __builtin_ctz(r);
                w0 ^= row[0]; w1 ^= row[52]; w2 ^= row[104]; w3 ^= row[156]; w4 ^= row[208]; w5 ^= row[260]; w6 ^= row[312]; w7 ^= row[364];
            }
        }
#undef PT_NIB2
    }
    // A vertex with a specular-only BSDF draws three dimensions (BSDF sample + roulette; no light sampling: path.rs:131) -- Sobol' only
    PT_DEV void load_window3() {
        base = dim; wn = 3;
        w0 = w1 = w2 = w3 = w4 = w5 = w6 = w7 = 0;
        if (halton || base + 3 > lds_dims) { base = 0xffffffffu; return; }
        const uint32_t *t = lds + base * 16u; const uint32_t js = lds_dims * 16u;
        const uint32_t lo = (uint32_t)index, hi = (uint32_t)(index >> 32);
#define PT_NIB3(j, x) { const uint32_t *q = t + (j) * js + ((x) & 15u); w0 ^= q[0]; w1 ^= q[16]; w2 ^= q[32]; }
        PT_NIB3(0, lo) PT_NIB3(1, lo >> 4) PT_NIB3(2, lo >> 8) PT_NIB3(3, lo >> 12) PT_NIB3(4, lo >> 16) PT_NIB3(5, lo >> 20) PT_NIB3(6, lo >> 24) PT_NIB3(7, lo >> 28)
        if (hi != 0) {
            PT_NIB3(8, hi) PT_NIB3(9, hi >> 4)
            for (uint32_t r = hi >> 8; r != 0; r &= r - 1) { const uint32_t *row = m32 + base * 52 + 40 + __builtin_ctz(r); w0 ^= row[0]; w1 ^= row[52]; w2 ^= row[104]; }
        }
#undef PT_NIB3
    }
    PT_DEV float sample_dimension(uint32_t d) {  // sobol.rs:68-86 / halton.rs:157-165 for dim >= 2
        if (d >= (halton ? kHaltonMaxDims : 1024u)) { overflow = true; return 0.0f; }  // the reference panics here
        const uint32_t k = d - base;
        if (base != 0xffffffffu && k < wn) {
            const uint32_t v = k == 0 ? w0 : k == 1 ? w1 : k == 2 ? w2 : k == 3 ? w3 : k == 4 ? w4 : k == 5 ? w5 : k == 6 ? w6 : w7;
            return halton ? __uint_as_float(v) : sobol_to_float(v);
        }
        if (halton) return halton_scrambled_radical_inverse(prime[d], perm + prime_sum[d], index);
        return sobol_to_float(sobol_bits_far(nib, m32, d, index));
    }
    PT_DEV float peek_sobol(uint32_t d) const {   // one Sobol' dimension of the staged ones (d < lds_dims), window or not
        return sobol_to_float(sobol_bits_nib(lds + d * 16u, lds_dims * 16u, m32 + d * 52, index));
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
