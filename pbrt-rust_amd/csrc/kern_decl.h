// kern_decl.h -- declarations of every kernel for the host driver (host_common.h and the .hip files it names). The definitions live in kern_*.h and are
// instantiated by the tu_*.hip translation units (one code object each, compiled in parallel).
#pragma once
#include "kern_common.h"

template <int ANY, int MODE, bool PROBE, int QUADK> __global__ void k_trace(DeviceScene s, TraceJob job);   // ANY: 0 closest hit, 1 any hit, 2 mixed (per-lane kind); QUADK: 0 two-wide (exact) walk, 1 production walk, 2 production walk of a pool beyond 4 GB
template <int MAXL, int MODE, int DIFF> __global__ void k_shade(DeviceScene s, RenderConst rc, SobolTables tabs, LightGrid grid, PathSoA ps, ShadeJob job);
template <bool SPH, bool VOL> __global__ void k_shade_miss(DeviceScene s, RenderConst rc, PathSoA ps, ShadeJob job);
template <bool SPH, bool VOL> __global__ void k_bssrdf(DeviceScene s, RenderConst rc, SobolTables tabs, LightGrid grid, PathSoA ps, BssrdfJob job);
__global__ void k_medium_route(DeviceScene s, RenderConst rc, SobolTables tabs, PathSoA ps, const uint32_t *queue, const uint32_t *count_ptr,
                               uint32_t *class_count, uint32_t *c0, uint32_t *c1, uint32_t *c2, uint32_t *c3, uint32_t *c4, uint32_t *c5, uint32_t *error);
__global__ void k_shade_medium(DeviceScene s, RenderConst rc, SobolTables tabs, LightGrid grid, PathSoA ps, ShadeJob job);
__global__ void k_build_packets(DeviceScene s, const uint32_t *ordered, uint32_t n_refs, TriPacket *out);
__global__ void k_mark_leaf_ends(TriPacket *leaf, const uint32_t *last_index, uint32_t n);
__global__ void k_light_area(DeviceScene s, float *area, float4 *rec);
template <int NQ, int CAP> __global__ void k_route(DeviceScene s, const uint32_t *queue, const uint32_t *count_ptr, PathSoA ps, uint32_t *class_count, RouteJob rj);
__global__ void k_generate(RenderConst rc, SobolTables tabs, PathSoA ps, uint32_t *q_ext, uint32_t *q_ext_count, DevCounters *counters);
__global__ void k_film(RenderConst rc, PathSoA ps, const float *filter_table, float *film_rgbw, DevCounters *counters);
template <bool SPH> __global__ void k_film_final(DeviceScene s, RenderConst rc, PathSoA ps, const float *filter_table, float *film_rgbw, DevCounters *counters);
__global__ void k_film_finish(const float *film_rgbw, float *film_xyzw, uint32_t npix);
__global__ void k_film_sum(FilmSumArgs a, float4 *dst, int accumulate, size_t n_quads);
__global__ void k_light_grid_contrib(DeviceScene s, uint32_t nvx, uint32_t nvy, uint32_t nvz, float *func, const uint32_t *cells, size_t n_cells, size_t stride);
__global__ void k_light_grid_finish(uint32_t n_lights, size_t ncell, float *func, float *cdf, float *func_int, const uint32_t *cells, size_t stride, unsigned long long *cell_ptr);
template <bool SPH> __global__ void k_light_touch(DeviceScene s, LightGrid g, PathSoA ps, const uint32_t *queue, const uint32_t *count_ptr, uint32_t kind, uint32_t max_depth, uint32_t *req_flag, uint32_t *req_list, uint32_t *req_count);
__global__ void k_halton_samples(SobolTables tabs, HaltonParams hp, uint32_t n, const int32_t *pixel_xy, const uint32_t *sample_num,
                                 uint32_t n_dims, float *out, uint64_t *out_index);
__global__ void k_sobol_samples(SobolTables tabs, SobolParams sp, uint32_t n, const int32_t *pixel_xy, const uint32_t *sample_num,
                                uint32_t n_dims, float *out, uint64_t *out_index);
__global__ void k_camera_rays(RenderConst rc, uint32_t n, const float *cs, float *out_o, float *out_d);
__global__ void k_dist1d_sample(const float *func, const float *cdf, float func_int, int n, int discrete, uint32_t n_u, const float *u, float *out_x, float *out_pdf, int32_t *out_off);
