// kernels.h -- shared declarations between the HIP kernels (kernels.hip) and the host driver (host_common.h: scene_create.hip, render_loop.hip, ...).
#pragma once
#include <cstddef>
#include "dev_scene.h"
#include "dev_sampler.h"
#include "dev_light.h"
#include "dev_bssrdf.h"

namespace ptd {

constexpr int kNumClasses = 11;     // material-sorted shade queues, one per shade KERNEL: by lobe count 0 matte, 1 one-lobe, 2 two-lobe, 3 many-lobe (general kernels),
//       // 4 = rays that escaped + resolve-only (dead) paths: a light kernel of their own
//       // 5 = medium vertices of the volumetric integrator (k_shade_medium)
//       // 6 = perfectly specular materials (mirror, smooth glass): no next-event estimation at their vertices (path.rs:131-146);
//       //     the volumetric integrator estimates direct light at every vertex, so its router folds this class into class 1
//       // and by lobe SET, the kernels specialised for one material's BxDFs (round 5: a class each, so that one substrate, rough glass or translucent
//       // material in a scene no longer sends every metal / plastic / uber vertex back to the general kernel of its lobe count):
//       // 7 metal (k_shade<1, ., 3>), 8 plastic-like: plastic and the opaque uber without specular terms (k_shade<2, ., 4>), 9 uber (k_shade<5, ., 5>),
//       // 10 smooth subsurface (k_shade<1, ., 6>). Classes 7-9 are handed out in textured scenes too (round 6: a texture changes a material's parameters, not its lobe set),
//       // class 10 in untextured scenes only (material_class(), scene_create.hip); the volumetric router folds all four back (class_general).
constexpr int kMissClass = 4, kMediumClass = 5, kSpecClass = 6, kMetalClass = 7, kPlasticClass = 8, kUberClass = 9, kSssClass = 10;
PT_HD uint32_t class_general(uint32_t c) { return c == (uint32_t)kMetalClass ? 1u : c == (uint32_t)kPlasticClass ? 2u : (c == (uint32_t)kUberClass || c == (uint32_t)kSssClass) ? 3u : c; }
constexpr int kRouteSlots = 12;        // most staging queues a k_route block holds
// k_route's staging queues: one per shade class the scene uses ("slot"), so that a scene pays LDS only for the classes it has.
// slot_map: the slot of class c in nibble c (15 = the class does not occur in this scene).
struct RouteJob { uint32_t n_slots; unsigned long long slot_map; uint32_t cls_of_slot[kRouteSlots]; uint32_t *buf[kRouteSlots]; uint32_t *error; uint32_t drop_cls; };   // slot_map: 4 bits per class, 15 = no queue; drop_cls: the one class whose entries are dropped on purpose (escaped rays when the film kernel ends the paths), any other class without a queue raises *error
constexpr int kRouteQueueCap = 2048;   // k_route's LDS staging queues (entries)
#ifndef PT_LDS_STACK
#define PT_LDS_STACK 10
#endif
constexpr int kLdsStack = PT_LDS_STACK;       // traversal stack entries (2 words each) kept in LDS per lane; deeper entries spill to HBM. Triangle-only scenes: 10, so
                                              // that seven workgroups fit a CU's LDS; scenes with instances push a marker entry per instance entered and run five
                                              // waves per SIMD: 12 (C4 with 10: trace +2.3 %)
#ifndef PT_LDS_STACK_GENERAL
#define PT_LDS_STACK_GENERAL 12
#endif
constexpr int kLdsStackGeneral = PT_LDS_STACK_GENERAL;
constexpr int kMaxStack = 64;       // the reference's stack size (accelerators/bvh.rs:722)
#ifndef PT_LDS_STACK_QUAD
#define PT_LDS_STACK_QUAD 15
#endif
constexpr int kLdsStackQuad = PT_LDS_STACK_QUAD;   // the four-wide walk (kern_trace.h, QUAD): up to three pushes per record; five waves per SIMD x 7 KB per wave of LDS
#ifndef PT_LDS_STACK_QUAD_INST
#define PT_LDS_STACK_QUAD_INST 19
#endif
constexpr int kLdsStackQuadInst = PT_LDS_STACK_QUAD_INST;   // the four-wide walk of scenes with instances (outer tree + marker + object tree on one stack): four waves per SIMD share a CU's LDS
constexpr int kMaxStackQuad = 96;   // a reference tree of depth 64 collapses to 32 four-wide levels x 3 pushes
constexpr int kLdsStackMinQuad = kLdsStackQuad < kLdsStackQuadInst ? kLdsStackQuad : kLdsStackQuadInst, kLdsStackMinTwo = kLdsStack < kLdsStackGeneral ? kLdsStack : kLdsStackGeneral;
constexpr int kSpillEntries = (kMaxStackQuad - kLdsStackMinQuad) > (kMaxStack - kLdsStackMinTwo) ? (kMaxStackQuad - kLdsStackMinQuad) : (kMaxStack - kLdsStackMinTwo);   // per-lane HBM stack entries behind the LDS ones
constexpr int kSpillWords = 2 * kSpillEntries + 8;   // per lane in the HBM slab of a wave: the stack entries behind the LDS ones, then the world-space ray of a lane inside an instance (six words; the four-wide walk of instanced scenes keeps it here, kern_trace.h)
constexpr int kTraceBlock = 256;
#ifndef PT_FILM_LANES
#define PT_FILM_LANES 1
#endif
constexpr uint32_t kFilmLanes = PT_FILM_LANES;   // threads per pixel slot of the film kernels (kern_film.h: film_slot); a power of two <= 64. 1: measured best (profiles/r6/NOTES.md section 4)
constexpr int kProbeRing = 8;       // k_trace<.., PROBE>: matching intersections of a BSSRDF probe chain kept per lane (3 x uint4 each)

// path flags (meta >> 24)
enum : uint32_t { PF_SPECULAR = 1u, PF_PEND_SHADOW = 2u, PF_PEND_MIS = 4u, PF_DEAD = 8u, PF_NEE_UNCOUNTED = 16u,
                  PF_CAMERA_RAY = 32u,     // the ray still carries the camera's differentials (cleared at the first shaded vertex)
                  PF_STAGE_B = 64u,        // volpath with grid media or material-less shells: the vertex has done its NEE set-up and waits, in its own shade class, for the traced rays before it samples on
                  PF_FINISHED = 128u };    // the path ended at a vertex with nothing pending: its shade kernel did its last step (k_film_final leaves it alone)

// Path state in HBM: five arrays of 16-byte-aligned RECORDS indexed by path id (pid). A record holds what one kernel reads or
// writes together, so a lane moves whole 16-byte quads of one 32- or 64-byte line (dwordx4 accesses, every fetched sector fully
// used) however sparse the path ids of a queue have become -- round 1 kept one 4-byte array per field, which cost one memory
// instruction per field and fetched a 64-byte sector for every 4 bytes once the surviving paths were scattered.
//   core  64 B  {L.rgb, etascale} {beta.rgb, meta} {sobol_index, pfilm.xy} {medium, mis_medium, -, -}      generate / shade / film
//   ray   32 B  {o.xyz, d.x} {d.yz, (t_max), -}                                                            continuation ray: shade -> trace
//   hit   32 B  {prim, b0, b1, b2} {inst, t, packet, packet flags}                                         trace -> route / shade
//   nee   64 B  {sh_o.xyz, sh_d.x} {sh_d.yz, occluded | sh_prim, nee_light} {A.rgb, choice_pdf} {nb.rgb, grid medium of the shadow ray}  pending shadow ray + its terms
//   mis   64 B  {mis_o.xyz, mis_d.x} {mis_d.yz, w, spdf} {mis_prim, b0, b1, b2} {f.rgb, mis_t}              pending MIS ray + its hit
// The accessors below name single words of those records; adjacent words accessed together merge into dwordx2/x4 instructions.
//   ext  128 B  {p1.p, -} {p1.p_error, -} {p1.n, -} {-} {shadow hit} {shadow hit 2} {MIS hit} {MIS hit 2}                  volpath in scenes with material-less
//               shells only (NULL otherwise): the far end of the shadow ray (VisibilityTester::p1) and the full hit records of the current
//               shadow / MIS SEGMENT, from which the next segment of VisibilityTester::tr / Scene::intersect_tr is spawned (vol_chain_step)
struct PathSoA {
    float *core, *ray, *hit, *nee, *mis, *ext;
    static constexpr int kCoreWords = 16, kRayWords = 8, kHitWords = 8, kNeeWords = 16, kMisWords = 16, kExtWords = 32;
#define PT_REC_F(name, arr, words, off) PT_HD float &name(size_t p) const { return arr[p * words + off]; }
#define PT_REC_U(name, arr, words, off) PT_HD uint32_t &name(size_t p) const { return reinterpret_cast<uint32_t *>(arr)[p * words + off]; }
    PT_REC_F(L_r, core, 16, 0) PT_REC_F(L_g, core, 16, 1) PT_REC_F(L_b, core, 16, 2) PT_REC_F(etascale, core, 16, 3)
    PT_REC_F(beta_r, core, 16, 4) PT_REC_F(beta_g, core, 16, 5) PT_REC_F(beta_b, core, 16, 6)
    PT_REC_U(meta, core, 16, 7)                          // dim (bits 0-15) | bounces (16-23) | flags (24-31)
    PT_HD uint64_t &sobol_index(size_t p) const { return reinterpret_cast<uint64_t *>(core)[p * 8 + 4]; }   // words 8-9
    PT_REC_F(pfilm_x, core, 16, 10) PT_REC_F(pfilm_y, core, 16, 11)
    PT_REC_U(medium, core, 16, 12)                       // volpath: Ray::medium of the continuation ray (PT_NONE = vacuum)
    PT_REC_U(mis_medium, core, 16, 13)                   // volpath: the MIS ray's medium
    PT_REC_F(ox, ray, 8, 0) PT_REC_F(oy, ray, 8, 1) PT_REC_F(oz, ray, 8, 2) PT_REC_F(dx, ray, 8, 3) PT_REC_F(dy, ray, 8, 4) PT_REC_F(dz, ray, 8, 5)   // continuation ray (t_max = inf)
    PT_REC_U(hit_prim, hit, 8, 0) PT_REC_F(hit_b0, hit, 8, 1) PT_REC_F(hit_b1, hit, 8, 2) PT_REC_F(hit_b2, hit, 8, 3)   // closest hit of the continuation ray
    PT_REC_U(hit_inst, hit, 8, 4)                        // instance the hit went through (PT_NONE: top level)
    PT_REC_F(hit_t, hit, 8, 5)                           // ray.t_max after Scene::intersect (volpath: later the medium vertex's parameter)
    PT_REC_U(hit_pkt, hit, 8, 6)                         // index of the hit's TriPacket in DeviceScene::leaf (vertices, ids and flags in one 48-byte line)
    PT_REC_U(hit_pflags, hit, 8, 7)                      // that packet's flag word (class bits: kMissClass for a miss) -- all k_route reads
    // pending next-event estimation of the previous vertex
    PT_REC_F(sh_ox, nee, 16, 0) PT_REC_F(sh_oy, nee, 16, 1) PT_REC_F(sh_oz, nee, 16, 2) PT_REC_F(sh_dx, nee, 16, 3) PT_REC_F(sh_dy, nee, 16, 4) PT_REC_F(sh_dz, nee, 16, 5)   // shadow ray (t_max = 1 - eps)
    PT_REC_U(occluded, nee, 16, 6)                       // any-hit result of the shadow ray ...
    PT_REC_U(sh_prim, nee, 16, 6)                        // ... or (volpath) its closest hit: VisibilityTester::tr intersects, it does not intersect_p
    PT_REC_U(nee_light, nee, 16, 7)
    PT_REC_F(A_r, nee, 16, 8) PT_REC_F(A_g, nee, 16, 9) PT_REC_F(A_b, nee, 16, 10)   // f*Li*w/lightpdf if unoccluded
    PT_REC_F(nee_choice_pdf, nee, 16, 11)
    PT_REC_F(nb_r, nee, 16, 12) PT_REC_F(nb_g, nee, 16, 13) PT_REC_F(nb_b, nee, 16, 14)   // beta at NEE time
    PT_REC_F(mis_ox, mis, 16, 0) PT_REC_F(mis_oy, mis, 16, 1) PT_REC_F(mis_oz, mis, 16, 2) PT_REC_F(mis_dx, mis, 16, 3) PT_REC_F(mis_dy, mis, 16, 4) PT_REC_F(mis_dz, mis, 16, 5)
    PT_REC_F(mis_w, mis, 16, 6) PT_REC_F(mis_spdf, mis, 16, 7)
    PT_REC_U(mis_prim, mis, 16, 8) PT_REC_F(mis_b0, mis, 16, 9) PT_REC_F(mis_b1, mis, 16, 10) PT_REC_F(mis_b2, mis, 16, 11)   // closest hit of the MIS ray
    PT_REC_F(mis_f_r, mis, 16, 12) PT_REC_F(mis_f_g, mis, 16, 13) PT_REC_F(mis_f_b, mis, 16, 14)
    PT_REC_F(mis_t, mis, 16, 15)                         // volpath: the MIS ray's hit parameter (Scene::intersect_tr)
#undef PT_REC_F
#undef PT_REC_U
};
constexpr int kPathBytes = 4 * (PathSoA::kCoreWords + PathSoA::kRayWords + PathSoA::kHitWords + PathSoA::kNeeWords + PathSoA::kMisWords);   // 256 B per path
// Per-path subsurface probe state (allocated only for scenes with a subsurface material), as 16-byte-quad RECORDS indexed by path id like PathSoA's (round 6; rounds 1-5
// kept 25 arrays of 4-byte fields here: the probe kernel gathered five of them dword by dword at every retired segment, k_bssrdf seventeen, each a 64-byte sector
// fetched for 4 bytes once the path ids of a queue were scattered -- k_bssrdf moved 5.5x its algorithmic bytes):
//   probe  32 B  {start.xyz, u1n} {target.xyz, material id}           the sampled probe segment of TabulatedBSSRDF::sample_sp (bssrdf.rs:357-365): shade -> probe kernel,
//                                                                      which reads it ONCE per chain (at the refill) and keeps target / material / u1n in registers
//   frame  48 B  {po.xyz, nfound} {ns.xyz, iface} {ss.xyz, material}  the outgoing point and its frame (shade ->), the chain's count and the selected intersection's
//                                                                      MediumInterface, inside | outside << 16, 0xffff = none (probe kernel ->): all k_bssrdf reads
//   coef   32 B  {sigma_a.rgb, -} {sigma_s.rgb, -}                    sigma_a / sigma_s as evaluated at the entry point (subsurface.rs:100-101) -- written and read
//                                                                      only where they are not the material's constants (bss_coef_stored: textured scenes, kdsubsurface)
struct BssSoA {
    float4 *probe, *frame, *coef;
    static constexpr int kProbeQuads = 2, kFrameQuads = 3, kCoefQuads = 2;
    PT_HD uint32_t &nfound(size_t p) const { return reinterpret_cast<uint32_t *>(frame)[p * 12 + 3]; }
    PT_HD uint32_t &iface(size_t p) const { return reinterpret_cast<uint32_t *>(frame)[p * 12 + 7]; }
};
constexpr int kBssBytes = 16 * (BssSoA::kProbeQuads + BssSoA::kFrameQuads + BssSoA::kCoefQuads);   // 112 B per path
// sigma_a / sigma_s travel with the path when they are not a function of the material alone: any textured scene (the textures are evaluated at the entry point), and
// kdsubsurface materials (their conversion, subsurface_from_diffuse, is a Catmull-Rom inversion nobody wants to run twice)
PT_HD bool bss_coef_stored(uint32_t n_textures, const PtMaterial &m) { return n_textures > 0u || m.kd_subsurface != 0u; }

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

// pt_multi_render's film merge: at most one film per replica (host_common.h: kMaxDevices)
constexpr int kMaxReplicas = 64;
struct FilmSumArgs { const float4 *src[kMaxReplicas]; uint32_t n; };

struct DevCounters {  // PtCounters mirror, atomically updated once per wave
    unsigned long long camera_rays, intersect_tests, shadow_tests, nodes, tri_tests, sphere_tests;
    unsigned long long zero_num, zero_den, path_len[16], san_nan, san_neg, san_inf, splats, stages;
    unsigned long long ref_asserts;            // assert!()s of the reference's li that would have fired (PtCounters::reference_asserts)
    unsigned long long shade_items[kNumClasses], shade_bytes[kNumClasses];  // path vertices shaded / path-state + queue bytes moved
    unsigned long long regions[16];            // PT_REGION_PROFILE builds: wave cycles per k_shade region
    unsigned long long tail[16];               // PT_TRACE_UTIL builds: [0] first wave start, [1] last wave exit of the launch in flight; per kind k: [4+2k] sum of wave busy time, [5+2k] sum of launch span x waves (wall_clock64 ticks)
    unsigned long long util2[8];               // PT_TRACE_UTIL builds: wave cycles of the record step's parts: [0] loads issued + waited for, [1] node branch, [2] leaf branch, [3] pops
    unsigned long long dbg[4];                 // PT_TRACE_UTIL builds: scheduling knobs as the kernel saw them
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

// One kind of ray of a traversal launch: where its rays come from and where its results go.
struct TraceSub {
    const uint32_t *queue;   // path ids (NULL => identity)
    const uint32_t *count;   // device count of queue entries
    // rays: 32-byte records {o.xyz, d.x} {d.y, d.z, t_max, -} at ray[pid * ray_stride] (stride in 16-byte quads: 2 = ray records, 4 = the
    // leading quads of nee / mis records); t_max is read from the record only if per_ray_tmax, else scalar_tmax
    const float4 *ray; uint32_t ray_stride; uint32_t per_ray_tmax;
    float scalar_tmax;
    // outputs (indexed by path id; strides in their own units)
    float4 *out_hit; uint32_t out_hit_stride;       // closest hit {prim, b0, b1, b2} as one quad (NULL: not wanted)
    uint32_t *out_word; uint32_t out_word_stride;   // any-hit: occluded flag; closest hit without out_hit: the primitive (volpath shadow rays)
    float4 *out_hit2;                               // second quad of a hit record {inst, t, packet index, packet flags | miss class}, stride out_hit_stride (NULL: not wanted)
    float *out_t; uint32_t out_t_stride;            // may be NULL (volpath MIS rays: mis_t)
    uint32_t kind;           // 0 extend, 1 extend_mis, 2 shadow, 3 extend_camera, 4 probe chains (per-kind work counters)
    uint32_t any;            // k_trace<2, ..> (mixed launch): this kind's rays are any-hit queries (Scene::intersect_p)
};

// A traversal launch. k_trace<0 | 1, ..> walks the rays of sub[0] (closest hit | any hit); k_trace<2, ..> walks the queues of
// sub[0..2] back to back -- the continuation, MIS and shadow rays of one wavefront iteration in ONE launch, so that the iteration
// has one tail of straggling rays instead of three (run_pass in render_loop.hip).
struct TraceJob {
    TraceSub sub[3];
    uint32_t *head;          // persistent-wave work head (zeroed before launch)
    uint32_t *spill;         // [waves_in_grid][64 lanes][2 * (kMaxStack - the kernel's LDS stack entries)], allocated for the smaller LDS stack
    uint32_t *error;
    DevCounters *counters;
    uint32_t refill_min;     // refill idle lanes from the queue once this many are idle (64 => only when the wave is empty)
    uint32_t leaf_quorum;    // lanes waiting at a leaf join the record fetch once this many wait (or no lane is at a node)
    uint32_t inst_quorum;    // lanes waiting to enter / leave an object instance run the transform step once this many wait
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
    uint32_t *self_next, *self_next_count;       // the SAME class's queue of the next iteration (volpath with grid media: stage B of a vertex)
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
    // volpath in scenes with grid media or material-less shells: the exit-point vertex waits for its traced shadow / MIS rays (stage B, as the
    // surface vertices of kern_shade.h do) in a queue of its own and comes back to k_bssrdf with stage_b set
    uint32_t *self_next, *self_next_count;
    uint32_t stage_b;
};

}  // namespace ptd
