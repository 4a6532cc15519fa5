// kernels.h -- shared declarations between the HIP kernels (kernels.hip) and the host driver (capi.hip).
#pragma once
#include "dev_scene.h"
#include "dev_sampler.h"
#include "dev_light.h"
#include "dev_bssrdf.h"

namespace ptd {

constexpr int kNumClasses = 6;      // material-sorted shade queues: 0 matte, 1 one-lobe, 2 two-lobe, 3 uber + subsurface,
constexpr int kMissClass = 4, kMediumClass = 5;   // 5 = medium vertices of the volumetric integrator (k_shade_medium)
//       // 4 = rays that escaped + resolve-only (dead) paths: a light kernel of their own
#ifndef PT_LDS_STACK
#define PT_LDS_STACK 12
#endif
constexpr int kLdsStack = PT_LDS_STACK;       // traversal stack entries (2 words each) kept in LDS per lane; deeper entries spill to HBM
constexpr int kMaxStack = 64;       // the reference's stack size (accelerators/bvh.rs:722)
constexpr int kTraceBlock = 256;
constexpr int kProbeRing = 8;       // k_trace<.., PROBE>: matching intersections of a BSSRDF probe chain kept per lane (3 x uint4 each)

// path flags (meta >> 24)
enum : uint32_t { PF_SPECULAR = 1u, PF_PEND_SHADOW = 2u, PF_PEND_MIS = 4u, PF_DEAD = 8u, PF_NEE_UNCOUNTED = 16u,
                  PF_CAMERA_RAY = 32u };   // the ray still carries the camera's differentials (cleared at the first shaded vertex)

// SoA path state in HBM; index = path id (pid). Every array has `capacity` entries.
struct PathSoA {
    float *pfilm_x, *pfilm_y;
    float *ox, *oy, *oz, *dx, *dy, *dz;                 // continuation ray (t_max = inf)
    uint32_t *hit_prim; float *hit_b0, *hit_b1, *hit_b2; // closest hit of the continuation ray
    uint32_t *hit_inst;                                  // instance the hit went through (PT_NONE: top level)
    float *beta_r, *beta_g, *beta_b, *L_r, *L_g, *L_b, *etascale;
    uint64_t *sobol_index;
    uint32_t *meta;                                      // dim (bits 0-15) | bounces (16-23) | flags (24-31)
    // pending next-event estimation of the previous vertex
    float *sh_ox, *sh_oy, *sh_oz, *sh_dx, *sh_dy, *sh_dz; // shadow ray (t_max = 1 - eps)
    float *A_r, *A_g, *A_b;                              // f*Li*w/lightpdf if unoccluded
    float *mis_ox, *mis_oy, *mis_oz, *mis_dx, *mis_dy, *mis_dz;
    float *mis_f_r, *mis_f_g, *mis_f_b, *mis_w, *mis_spdf;
    uint32_t *nee_light; float *nee_choice_pdf;
    float *nb_r, *nb_g, *nb_b;                           // beta at NEE time
    uint32_t *mis_prim; float *mis_b0, *mis_b1, *mis_b2; // closest hit of the MIS ray
    uint8_t *occluded;
    // volumetric path integrator only (PtRenderParams.integrator == PT_INTEGRATOR_VOLPATH)
    uint32_t *medium;        // Ray::medium of the continuation ray (PT_NONE = vacuum)
    float *hit_t;            // its hit parameter (ray.t_max after Scene::intersect); then the medium vertex's parameter
    uint32_t *mis_medium; float *mis_t;   // the MIS ray's medium and hit parameter (Scene::intersect_tr)
    uint32_t *sh_prim;       // closest hit of the shadow ray (VisibilityTester::tr intersects, it does not intersect_p)
};
// Per-path subsurface probe state (allocated only for scenes with a subsurface material): the sampled probe segment of
// TabulatedBSSRDF::sample_sp (bssrdf.rs:357-365), the outgoing point's frame, and the chain counters.
struct BssSoA {
    float *start_x, *start_y, *start_z, *target_x, *target_y, *target_z;
    float *po_x, *po_y, *po_z, *ns_x, *ns_y, *ns_z, *ss_x, *ss_y, *ss_z;
    float *u1n;
    uint32_t *mat;   // material id of the BSSRDF (Arc::ptr_eq test of the chain, bssrdf.rs:385-391)
    uint32_t *cnt;   // nfound of the finished chain (written by k_trace<.., PROBE>, read by k_bssrdf)
};
constexpr int kBssSoAArrays = 18;
constexpr int kPathSoAFloatArrays = 56;  // 4-byte arrays in the slab (+ one u64 array + one u8 array)

struct QueueSet {
    uint32_t *ext[2];                 // pids with a continuation ray to trace (ping-pong)
    uint32_t *shade[2][kNumClasses];  // pids to shade, per material class (ping-pong)
    uint32_t *shadow, *mis;           // pids with pending shadow / MIS rays
    uint32_t *probe[2];               // pids walking a BSSRDF probe chain (ping-pong; NULL without subsurface materials)
    // counters (device): layout documented in QCounters
};
struct QCounters {
    uint32_t ext[2];
    uint32_t shade[2][kNumClasses];
    uint32_t shadow, mis;
    uint32_t probe[2];
    uint32_t head[4];                 // persistent-wave work heads for the trace launches
    uint32_t error;                   // PtStatus raised on device (stack / sobol overflow)
    uint32_t pad;
};

struct DevCounters {  // PtCounters mirror, atomically updated once per wave
    unsigned long long camera_rays, intersect_tests, shadow_tests, nodes, tri_tests, sphere_tests;
    unsigned long long zero_num, zero_den, path_len[16], san_nan, san_neg, san_inf, splats, stages;
    unsigned long long shade_items[kNumClasses], shade_bytes[kNumClasses];  // path vertices shaded / path-state + queue bytes moved
    unsigned long long regions[16];            // PT_REGION_PROFILE builds: wave cycles per k_shade region
    unsigned long long bss_items, bss_bytes;   // k_bssrdf: probe steps processed / state bytes moved
    unsigned long long k_nodes[5], k_tris[5], k_rays[5];  // per trace launch kind: 0 extend, 1 extend_mis, 2 shadow, 3 extend_camera, 4 extend_probe (segments)
};

struct RenderConst {
    // pass geometry
    uint32_t n_tile_slots;            // tiles owned by this rank
    uint32_t tile_rank, tile_world;
    uint32_t ntx, nty;                // tile grid of integrator.rs:277-279
    uint32_t spp, s_begin, s_count;   // samples of this pass: [s_begin, s_begin + s_count)
    uint32_t n_pix_slots;             // n_tile_slots * 256
    int32_t sample_bounds[4], pixel_bounds[4], crop[4];
    SobolParams sobol;
    HaltonParams halton;
    M4 raster_to_camera, camera_to_world;
    float lens_radius, focal_distance, shutter_open, shutter_close;
    float dx_camera[3], dy_camera[3];  // perspective.rs:64-70
    float inv_sqrt_spp;                // Ray::scale_differential factor (integrator.rs:340)
    uint32_t max_depth; float rr_threshold;
    uint32_t volpath, camera_medium;   // VolPathIntegrator (volpath.rs) instead of PathIntegrator; the camera's medium
    float filter_radius[2]; float max_sample_luminance;
    uint32_t film_w, film_h;
};

struct TraceJob {
    const uint32_t *queue;   // path ids (NULL => identity)
    const uint32_t *count;   // device count of queue entries
    uint32_t *head;          // persistent-wave work head (zeroed before launch)
    const float *ox, *oy, *oz, *dx, *dy, *dz;
    const float *tmax;       // per-ray t_max or NULL => scalar_tmax
    float scalar_tmax;
    // outputs (indexed by path id)
    uint32_t *out_prim; float *out_t, *out_b0, *out_b1, *out_b2;
    uint32_t *out_inst;      // may be NULL
    uint8_t *out_occluded;
    // shade-queue routing (closest-hit of continuation rays only)
    uint32_t *class_count;   // [kNumClasses] or NULL
    uint32_t *class_buf[kNumClasses];
    uint32_t *spill;         // [waves_in_grid][64 lanes][2 * (kMaxStack - kLdsStack)]
    uint32_t *error;
    DevCounters *counters;
    uint32_t kind;           // 0 extend, 1 extend_mis, 2 shadow, 3 extend_camera (per-kind work counters)
    uint32_t refill_min;     // refill idle lanes from the queue once this many are idle (64 => only when the wave is empty)
    uint32_t leaf_quorum;    // lanes waiting at a leaf join the record fetch once this many wait (or no lane is at a node)
    // PROBE launches only: the chains' per-path inputs (start, target, material, u1) / output (cnt = nfound) and the per-lane ring
    BssSoA bs;
    uint4 *ring;             // [waves_in_grid][kProbeRing][3][64 lanes]
};

struct ShadeJob {
    const uint32_t *queue; const uint32_t *count;
    uint32_t *ext_next, *ext_next_count;
    uint32_t *shade_next0, *shade_next0_count;   // resolve-only paths go to the miss class (kMissClass) of the next iteration
    uint32_t *shadow, *shadow_count, *mis, *mis_count;
    uint32_t *error;
    DevCounters *counters;
    uint32_t cls;            // material class of this launch (statistics)
    uint32_t *probe_next, *probe_next_count;     // paths starting a BSSRDF probe chain (class 3 only)
    BssSoA bs;
};

struct BssrdfJob {           // k_bssrdf: the vertex at the exit point of every finished probe chain
    const uint32_t *queue; const uint32_t *count;
    uint32_t *ext_next, *ext_next_count;
    uint32_t *shade_next0, *shade_next0_count;
    uint32_t *shadow, *shadow_count, *mis, *mis_count;
    uint32_t *error;
    DevCounters *counters;
    BssSoA bs;
};

}  // namespace ptd
