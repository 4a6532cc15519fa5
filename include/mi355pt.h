/*
 * mi355pt.h -- C ABI of the MI355X wavefront path-tracing back end.
 *
 * This is the drop-in boundary for ONE path of alexmeli100/pbrt-rust: the call
 *     integ.render(&scene)                                   src/core/api.rs:1746
 * i.e. `trait Integrator { fn render(&mut self, scene: &Scene); }`
 *                                                            src/core/integrator.rs:249-252
 * as implemented by SamplerIntegrator::render                src/core/integrator.rs:263-403
 * and PathIntegrator::li                                     src/integrators/path.rs:79-222.
 *
 * A Rust `Integrators::GpuPath` variant flattens the
 * already-built `Scene` (lights, BVHAccel nodes + ordered primitives, triangle
 * meshes, materials), the `PerspectiveCamera`, `SobolSampler` and `Film`
 * parameters into the POD structs below and calls pt_scene_create / pt_render.
 * Nothing here is a torch type; all pointers are plain host pointers unless a
 * comment says "device".  No function unwinds or aborts across the boundary:
 * every entry point returns a PtStatus.
 *
 * Ownership: the caller owns every input array for the duration of the call
 * (the library copies to HBM).  Output buffers are caller allocated.  Opaque
 * handles are released with pt_scene_destroy.
 * Threading: pt_render is blocking and not re-entrant for one pt_scene.
 */
#ifndef MI355PT_H
#define MI355PT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum PtStatus {
    PT_OK = 0,
    PT_ERR_INVALID_ARG = 1,       /* null pointer / inconsistent sizes                      */
    PT_ERR_NO_DEVICE = 2,         /* HIP runtime reports no usable gfx950 device            */
    PT_ERR_HIP = 3,               /* a HIP call failed; see pt_last_error()                  */
    PT_ERR_UNSUPPORTED = 4,       /* scene uses a feature outside the implemented rows      */
    PT_ERR_SOBOL_DIMENSIONS = 5,  /* a path consumed >= 1024 Sobol' dimensions
                                     (the reference panics: samplers/sobol.rs:69-73)         */
    PT_ERR_STACK_OVERFLOW = 6,    /* a ray's BVH traversal stack grew beyond what the walk holds: 96 pending entries in the
                                     four-wide production walk, 64 in the two-wide exact walk (= the reference's stack size;
                                     the reference indexes its 64-entry Vec without a check of its own: accelerators/bvh.rs:722) */
    PT_ERR_OUT_OF_MEMORY = 7,     /* a device allocation failed (e.g. spp_per_pass asks for more path state than the device has free);
                                     the scene stays usable: the next pt_render allocates its workspace afresh */
    PT_ERR_PROBE_CHAIN = 8        /* a BSSRDF probe chain (core/bssrdf.rs:376-394) found more than 32767 intersections or a
                                     pass needed more than 65536 wavefront iterations: the reference would still be walking its
                                     linked list; the render is abandoned instead of returning a truncated chain          */
} PtStatus;

/* ---- scene description -------------------------------------------------------------- */

/* Triangle flag bits (per triangle; replaces fields of shapes/triangle.rs:76-84 and the
 * per-mesh `n/s/uv.is_empty()` tests of shapes/triangle.rs:303-392). */
enum {
    PT_TRI_REVERSE_ORIENTATION = 1u << 0,
    PT_TRI_SWAPS_HANDEDNESS    = 1u << 1,
    PT_TRI_HAS_N               = 1u << 2,
    PT_TRI_HAS_S               = 1u << 3,
    PT_TRI_HAS_UV              = 1u << 4
};

/* Primitive shape reference: kind in the top 2 bits, index below. */
#define PT_SHAPE_TRIANGLE 0u
#define PT_SHAPE_SPHERE   1u
#define PT_SHAPE_REF(kind, index) (((uint32_t)(kind) << 30) | (uint32_t)(index))
#define PT_NONE 0xffffffffu

/* shapes/sphere.rs:27-57 (fields of Sphere after Sphere::new). Matrices are row major
 * `m.m[row][col]` exactly as core/transform.rs stores them. */
typedef struct PtSphere {
    float object_to_world[16];
    float world_to_object[16];
    float radius, z_min, z_max, theta_min, theta_max, phi_max;
    uint32_t reverse_orientation;
    uint32_t transform_swaps_handedness;
    /* kind = PT_QUADRIC_DISK: shapes/disk.rs:17-43 -- `z_min` = `z_max` = height, `radius`, `inner_radius`, `phi_max` (theta_* unused) */
    uint32_t kind;
    float inner_radius;
} PtSphere;
typedef enum PtQuadricKind { PT_QUADRIC_SPHERE = 0, PT_QUADRIC_DISK = 1 } PtQuadricKind;

typedef enum PtMaterialType {
    PT_MAT_MATTE = 0,      /* materials/matte.rs      */
    PT_MAT_MIRROR = 1,     /* materials/mirror.rs     */
    PT_MAT_GLASS = 2,      /* materials/glass.rs      */
    PT_MAT_PLASTIC = 3,    /* materials/plastic.rs    */
    PT_MAT_METAL = 4,      /* materials/metal.rs      */
    PT_MAT_UBER = 5,       /* materials/uber.rs       */
    PT_MAT_SUBSTRATE = 6,  /* materials/substrate.rs  */
    PT_MAT_SUBSURFACE = 7, /* materials/subsurface.rs (kdsubsurface: the host converts Kd/mfp with
                              subsurface_from_diffuse, bssrdf.rs:190-202, and passes sigma_a/sigma_s) */
    PT_MAT_TRANSLUCENT = 8,/* materials/translucent.rs: Kd, Ks, roughness; kr = "reflect", kt = "transmit" */
    PT_MAT_MIX = 9,        /* materials/mix.rs: kd = "amount", mix[0] / mix[1] = the two named materials */
    PT_MAT_DISNEY = 10     /* materials/disney.rs: kd = "color", eta, roughness + disney[] / disney_thin / disney_scatter */
} PtMaterialType;

/* ---- textures (core/texture.rs, textures/, core/mipmap.rs) --------------------------------------
 * A texture is a node of a tree, exactly as the reference's Arc<Textures<..>> values: children are texture indices
 * (constants are ConstantTexture nodes, as TextureParams::get_*texture creates them, paramset.rs:500-600). Float-valued
 * textures use component 0. Not covered: ImageWrap::Clamp (the reference's texel() clamps to `u` instead of `u - 1`,
 * mipmap.rs:305). */
typedef enum PtTextureType {
    PT_TEX_CONSTANT = 0,        /* textures/constant.rs                                  value[]                 */
    PT_TEX_SCALE = 1,           /* textures/scaled.rs:30-33    tex1 * tex2               child[0], child[1]      */
    PT_TEX_MIX = 2,             /* textures/mix.rs:29-35       t1*(1-amt) + t2*amt       child[0..2] (amount = 2) */
    PT_TEX_CHECKERBOARD2D = 3,  /* textures/checkerboard.rs:27-70   mapping, aamode      child[0], child[1]      */
    PT_TEX_CHECKERBOARD3D = 4,  /* textures/checkerboard.rs:87-100  world_to_texture     child[0], child[1]      */
    PT_TEX_IMAGEMAP = 5,        /* textures/imagemap.rs:167-176 + MIPMap::lookup2        mapping, image          */
    PT_TEX_UV = 6,              /* textures/uv.rs:21-33                                  mapping                 */
    PT_TEX_BILERP = 7,          /* textures/biler.rs:27-36                               mapping, v00..v11       */
    PT_TEX_FBM = 8,             /* textures/fbm.rs:22-29        fbm(p, dpdx, dpdy, omega, octaves)   world_to_texture   */
    PT_TEX_WRINKLED = 9,        /* textures/wrinkled.rs:22-29   turbulence(..)                       world_to_texture   */
    PT_TEX_WINDY = 10,          /* textures/windy.rs:20-30      |fbm(.1p, 3 oct)| * fbm(p, 6 oct)    world_to_texture   */
    PT_TEX_MARBLE = 11,         /* textures/marble.rs:36-65     spline(sin(p.y*scale + variation*fbm)) world_to_texture */
    PT_TEX_DOTS = 12            /* textures/dots.rs:28-55       mapping; child[0] = outside, child[1] = inside           */
} PtTextureType;
typedef enum PtMappingType { PT_MAP_UV = 0, PT_MAP_PLANAR = 1, PT_MAP_SPHERICAL = 2, PT_MAP_CYLINDRICAL = 3 } PtMappingType;  /* core/texture.rs:112-270 */
typedef enum PtImageWrap { PT_WRAP_REPEAT = 0, PT_WRAP_BLACK = 1 } PtImageWrap;                                             /* core/mipmap.rs:52 */

typedef struct PtTexture {
    uint32_t type;              /* PtTextureType */
    int32_t child[3];           /* texture indices, -1 = unused */
    float value[3];             /* CONSTANT */
    float v00[3], v01[3], v10[3], v11[3];   /* BILERP */
    /* TextureMapping2D (get_mapping2d, texture.rs:439-466) / IdentityMapping3D for CHECKERBOARD3D */
    uint32_t mapping;           /* PtMappingType */
    float su, sv, du, dv;       /* uv: uscale vscale udelta vdelta ; planar: du = udelta (ds), dv = vdelta (dt) */
    float vs[3], vt[3];         /* planar v1, v2 */
    float world_to_texture[16]; /* spherical / cylindrical / 3-D checkerboard: inverse(texture-to-world CTM), row major */
    uint32_t aa_closedform;     /* CHECKERBOARD2D: 0 = "none" (the reference's default), 1 = "closedform" */
    /* IMAGEMAP */
    uint32_t image;             /* index into PtSceneDesc.images */
    uint32_t trilinear;         /* "trilinear" (false => EWA) */
    float max_anisotropy;       /* "maxanisotropy" (8) */
    uint32_t wrap;              /* PtImageWrap */
    /* noise textures: "octaves" (8), "roughness" = omega (0.5); marble "scale" (1), "variation" (0.2) */
    uint32_t octaves; float omega, marble_scale, variation;
} PtTexture;

/* MIPMap pyramid (core/mipmap.rs:75-198), built by the host exactly as MIPMap::new does (power-of-two resampling with the
 * Lanczos weights of :264-291, then 2x2 box filtering per level) after the y flip, scale and inverse gamma of
 * imagemap.rs:141-157. Level l has max(1, width >> l) x max(1, height >> l) texels, rows = t, stored consecutively. */
typedef struct PtImage {
    uint32_t width, height;     /* level 0, powers of two */
    uint32_t n_levels;          /* 1 + log2(max(width, height)) */
    uint32_t channels;          /* 3 = RGBSpectrum memory, 1 = Float memory */
    const float *texels;        /* all levels, level 0 first */
} PtImage;

/* Material parameter slots that may be textured (PtMaterial.tex[slot] = texture index or -1 => the constant field). */
typedef enum PtMatParam {
    PT_MP_KD = 0, PT_MP_KS = 1, PT_MP_KR = 2, PT_MP_KT = 3, PT_MP_OPACITY = 4, PT_MP_ETA_RGB = 5, PT_MP_K_RGB = 6,
    PT_MP_SIGMA_A = 7, PT_MP_SIGMA_S = 8,
    PT_MP_SIGMA = 9, PT_MP_ROUGHNESS = 10, PT_MP_U_ROUGHNESS = 11, PT_MP_V_ROUGHNESS = 12, PT_MP_ETA = 13,
    PT_MP_BUMP = 14,            /* "bumpmap" float texture (core/material.rs:46-87); -1 = none */
    PT_MP_MFP = 15,             /* kdsubsurface "mfp" (see PtMaterial.kd_subsurface) */
    PT_MP_COUNT = 16
} PtMatParam;

typedef enum PtDisneyParam {
    PT_DS_METALLIC = 0, PT_DS_SPECULARTINT = 1, PT_DS_ANISOTROPIC = 2, PT_DS_SHEEN = 3, PT_DS_SHEENTINT = 4, PT_DS_CLEARCOAT = 5,
    PT_DS_CLEARCOATGLOSS = 6, PT_DS_SPECTRANS = 7, PT_DS_FLATNESS = 8, PT_DS_DIFFTRANS = 9
} PtDisneyParam;

/* Material parameters. Field use per type follows the reference's create_*_material parameter names; every field is the
 * value of a ConstantTexture unless tex[slot] names a texture. */
typedef struct PtMaterial {
    uint32_t type;
    float kd[3];
    float ks[3];
    float kr[3];
    float kt[3];
    float opacity[3];
    float eta_rgb[3];       /* metal: eta        */
    float k_rgb[3];         /* metal: k          */
    float sigma;            /* matte             */
    float eta;              /* glass/uber index  */
    float roughness;        /* plastic/metal/uber "roughness" */
    float u_roughness;      /* glass/substrate: uroughness; metal/uber: < 0 => use `roughness` */
    float v_roughness;
    uint32_t remap_roughness;
    /* subsurface (materials/subsurface.rs:47-106): sigma_a/sigma_s (before `scale`; tex[PT_MP_SIGMA_A / _S] may name spectrum textures), scale, and the index of the
     * material's BSSRDFTable in PtSceneDesc.bssrdf_tables (built by the host: compute_beam_diffusion_bssrdf). */
    float sigma_a[3];
    float sigma_s[3];
    float scale;
    uint32_t bssrdf_table;
    int32_t tex[16];        /* PtMatParam slot -> texture index, -1 = constant (a zeroed struct must set these to -1) */
    /* mix (materials/mix.rs:25-50): indices of "namedmaterial1" / "namedmaterial2" in the materials array (not mix or
     * subsurface materials; at most 5 lobes together); "amount" is kd / tex[PT_MP_KD]; tex[PT_MP_BUMP] = material 1's bump map
     * (material 2's bump map only perturbs a copy of the interaction that the reference then discards). */
    uint32_t mix[2];
    /* disney (materials/disney.rs:842-887): the float parameters that are constants here, indexed by PtDisneyParam;
     * "color" = kd / tex[PT_MP_KD], "eta" = eta / tex[PT_MP_ETA], "roughness" = roughness / tex[PT_MP_ROUGHNESS] may be textured. */
    float disney[10];
    uint32_t disney_thin;
    float disney_scatter[3];   /* "scatterdistance": non-black (and not thin) => DisneyBSSRDF (disney.rs:442-704) with a constant color */
    /* kdsubsurface with a textured "Kd" or "mfp" (materials/kdsubsurface.rs:96-99): kd_subsurface = 1 makes a PT_MAT_SUBSURFACE material take
     * its coefficients from subsurface_from_diffuse(table, Kd, mfp * scale) at every hit -- Kd = kd / tex[PT_MP_KD], mfp = mfp / tex[PT_MP_MFP];
     * sigma_a / sigma_s are ignored then. With constant Kd and mfp the host does that conversion once and leaves kd_subsurface = 0. */
    float mfp[3];
    uint32_t kd_subsurface;
} PtMaterial;

typedef enum PtLightType {
    PT_LIGHT_DIFFUSE_AREA = 0, /* lights/diffuse.rs  */
    PT_LIGHT_DISTANT = 1,      /* lights/distant.rs  */
    PT_LIGHT_POINT = 2,        /* lights/point.rs    */
    PT_LIGHT_INFINITE = 3,     /* lights/infinite.rs */
    PT_LIGHT_SPOT = 4          /* lights/spot.rs     */
} PtLightType;

typedef struct PtLight {
    uint32_t type;
    float L[3];              /* Lemit (area) | L (distant, world-space radiance) | I (point/spot) | unused (infinite: texels) */
    uint32_t two_sided;      /* area                                                            */
    uint32_t prim;           /* area: index of the primitive carrying this light (api.rs:1535-1546) */
    float pos[3];            /* point/spot: plight (world)                                      */
    float dir[3];            /* distant: w_light (world, normalised by the host, distant.rs:29)  */
    float cos_total_width;   /* spot                                                            */
    float cos_falloff_start; /* spot                                                            */
    float light_to_world[16];
    float world_to_light[16];
} PtLight;

/* accelerators/bvh.rs:89-95 LinearBVHNode: left child at index+1, right child at `offset`
 * for interior nodes (n_prims == 0); leaves index `ordered_prims[offset .. offset+n_prims]`. */
/* Participating media (api.rs make_medium :680-762). PT_MEDIUM_HOMOGENEOUS: HomogeneousMedium (media/homogeneous.rs:13-29): sigma_a,
 * sigma_s after "scale", Henyey-Greenstein g. PT_MEDIUM_GRID: GridDensityMedium (media/grid.rs): the same three plus a density grid
 * `density[(z * ny + y) * nx + x]` over the unit cube of medium space and `world_to_medium` = inverse(medium_to_world * translate(p0) *
 * scale(p1 - p0)); sigma_t = (sigma_a + sigma_s)[0] and 1 / max density are derived by the library as grid.rs:46-60 does. */
typedef enum PtMediumType { PT_MEDIUM_HOMOGENEOUS = 0, PT_MEDIUM_GRID = 1 } PtMediumType;
typedef struct PtMedium {
    float sigma_a[3]; float sigma_s[3]; float g;
    uint32_t type;                 /* PtMediumType */
    uint32_t nx, ny, nz;           /* grid media */
    float world_to_medium[16];
    const float *density;          /* nx * ny * nz values (host memory in PtSceneDesc) */
} PtMedium;

typedef enum PtSplitMethod { PT_SPLIT_SAH = 0, PT_SPLIT_HLBVH = 1 } PtSplitMethod;
typedef struct PtBVHNode {
    float bmin[3];
    float bmax[3];
    uint32_t offset;
    uint16_t n_prims;
    uint8_t axis;
    uint8_t pad;
} PtBVHNode;

/* Object instancing (core/api.rs:1630-1713, core/primitive.rs:40-88; static transforms only).
 * An object is a contiguous range of the prim arrays that is NOT part of the top level; the library builds one BVH per
 * object with more than one primitive (api.rs:1692-1700). An instance is a TransformedPrimitive of that object. */
typedef struct PtObject { uint32_t first_prim, n_prims; } PtObject;
typedef struct PtInstance {
    uint32_t object;
    float instance_to_world[16];   /* prim_to_world (start transform), row major */
    float world_to_instance[16];
} PtInstance;
#define PT_TOP_INSTANCE 0x80000000u  /* top_refs entry: instance index | PT_TOP_INSTANCE, else a primitive index */

/* core/bssrdf.rs:241-268 BSSRDFTable (100 albedo x 64 radius samples in the reference). */
typedef struct PtBSSRDFTable {
    uint32_t n_rho, n_radius;
    const float *rho_samples;     /* n_rho              */
    const float *radius_samples;  /* n_radius           */
    const float *profile;         /* n_rho * n_radius   */
    const float *rhoeff;          /* n_rho              */
    const float *profile_cdf;     /* n_rho * n_radius   */
} PtBSSRDFTable;

typedef struct PtSceneDesc {
    /* All triangle meshes concatenated; P is world space (shapes/triangle.rs:39-41). */
    uint32_t n_vertices;
    const float *P;          /* 3*n_vertices                      */
    const float *N;          /* 3*n_vertices or NULL              */
    const float *S;          /* 3*n_vertices or NULL              */
    const float *UV;         /* 2*n_vertices or NULL              */
    uint32_t n_triangles;
    const uint32_t *indices; /* 3*n_triangles                     */
    const uint8_t *tri_flags;/* n_triangles or NULL (=0)          */

    uint32_t n_spheres;
    const PtSphere *spheres;

    /* Scene primitives in api.rs creation order (one GeometricPrimitive per shape,
     * core/api.rs:1544-1545). */
    uint32_t n_prims;
    const uint32_t *prim_shape;    /* PT_SHAPE_REF(kind, index)                   */
    const uint32_t *prim_material; /* index into materials or PT_NONE             */
    const uint32_t *prim_light;    /* index into lights (area light) or PT_NONE   */

    uint32_t n_materials;
    const PtMaterial *materials;

    uint32_t n_lights;             /* scene.lights order (core/scene.rs:24-27)    */
    const PtLight *lights;

    /* InfiniteAreaLight data (at most one infinite light carries a map):
     * level-0 texels of the MIPMap after the host's resampling (RGB, row major,
     * power-of-two resolution; core/mipmap.rs:93-150) and the 2w x 2h scalar importance
     * image (lights/infinite.rs:62-81). NULL texels => no infinite light data. */
    uint32_t env_width, env_height;
    const float *env_texels;       /* 3*env_width*env_height       */
    const float *env_importance;   /* (2*env_width)*(2*env_height) */
    /* InfiniteAreaLight::power (infinite.rs:103-109) reads `map.lookup((.5,.5), .5)`: MIPMap level `levels - 2`, a host-side
     * pyramid value (the texel itself for a 1x1 map). Used by the "power" light sample strategy only. */
    float env_power_lookup[3];

    /* Accelerator: either prebuilt by the caller (reference order) or NULL => the
     * library builds the same SAH tree (accelerators/bvh.rs:200-375,662-693). */
    uint32_t max_node_prims;       /* bvh "maxnodeprims", default 4 */
    uint32_t n_nodes;
    const PtBVHNode *nodes;
    const uint32_t *ordered_prims; /* n_top entries (n_prims without instancing): positions in the top-level list */

    /* Instancing (all zero/NULL when unused). top_refs lists the scene's top-level primitives in creation order
     * (RenderOptions.primitives, api.rs:1585-1617,1703-1712); NULL => every primitive, in order. */
    uint32_t n_objects; const PtObject *objects;
    uint32_t n_instances; const PtInstance *instances;
    uint32_t n_top; const uint32_t *top_refs;

    uint32_t n_bssrdf_tables; const PtBSSRDFTable *bssrdf_tables;

    uint32_t n_textures; const PtTexture *textures;
    /* alpha masks (TriangleMesh.alpha_mask / shadow_alpha_mask, triangle.rs:29-30,275-285,497-545): per triangle, the
     * index of a float texture or -1; both arrays may be NULL. */
    const int32_t *tri_alpha; const int32_t *tri_shadow_alpha;
    uint32_t n_images; const PtImage *images;
    const float *ewa_weight_lut;   /* [128] = exp(-2 r2) - exp(-2), r2 = i/127 (mipmap.rs:40-50); required with EWA image maps */
    /* Participating media for the volumetric path integrator (media/homogeneous.rs, core/medium.rs). Ignored by
     * PT_INTEGRATOR_PATH exactly as the reference's PathIntegrator ignores Ray::medium. prim_medium_inside / _outside are the
     * GeometricPrimitive's MediumInterface per primitive (PT_NONE = no medium; api.rs:1540-1560); NULL = no interfaces. A
     * primitive without a material (prim_material = PT_NONE: `Material "none"`, api.rs:597) is a medium-interface shell: shadow and
     * MIS rays go on behind it, segment by segment (VisibilityTester::tr light.rs:125-150, Scene::intersect_tr scene.rs:68-87), and
     * the path itself steps over it with the reference's `bounces -= 1; continue` (volpath.rs:152-156) -- which skips the loop's
     * increment: the count DROPS at every shell and wraps below zero at a camera ray's first one (the release build has no overflow
     * checks), so such a path ends at its next vertex. Reproduced as it is; maxdepth <= 254 keeps the 8-bit count equivalent. */
    uint32_t n_media; const PtMedium *media;
    const uint32_t *prim_medium_inside; const uint32_t *prim_medium_outside;
    /* bvh "splitmethod" (api.rs make_accelerator -> bvh.rs:918-940) for every accelerator the library builds (ignored for
     * an adopted one). PT_SPLIT_SAH: host SAH builder = the reference's tree. PT_SPLIT_HLBVH: built on the GPU
     * (Morton sort + LBVH treelets + SAH upper levels, bvh.rs:377-660; see csrc/gpu_bvh.hip for the exact tree). */
    uint32_t split_method;
} PtSceneDesc;

/* ---- render parameters ---------------------------------------------------------------- */

typedef enum PtIntegratorType { PT_INTEGRATOR_PATH = 0, PT_INTEGRATOR_VOLPATH = 1 } PtIntegratorType;
typedef enum PtSamplerType { PT_SAMPLER_SOBOL = 0, PT_SAMPLER_HALTON = 1 } PtSamplerType;
/* "lightsamplestrategy" (path.rs:236, lightdistrib.rs:14-34). PT_LS_SPATIAL is SpatialLightDistribution (lightdistrib.rs:105-340): one
 * Distribution1D over ALL lights per voxel of a <= 64^3 grid. The reference fills a voxel on first touch (a lock-free hash,
 * lightdistrib.rs:249-337); a voxel's content is a pure function of the voxel, so WHEN it is filled cannot be observed. The library
 * fills every voxel up front when voxels x lights <= 2^25 (one launch, nothing in the render loop), and otherwise on first touch,
 * once per wavefront iteration: the vertices about to be shaded name their voxels (k_light_touch), the new ones are computed by
 * one k_light_grid_contrib launch (128 Halton points x every light, as compute_distribution does) and live until the scene is
 * destroyed. There is no limit on the number of lights other than memory: a touched voxel costs 8 (n_lights + 3) bytes -- every
 * emissive triangle is a light (api.rs:1531-1546). PT_LS_SPATIAL_EAGER / _LAZY force one of the two forms (tests; same result). */
typedef enum PtLightStrategy { PT_LS_UNIFORM = 0, PT_LS_POWER = 1, PT_LS_SPATIAL = 2, PT_LS_SPATIAL_EAGER = 3, PT_LS_SPATIAL_LAZY = 4 } PtLightStrategy;

typedef struct PtRenderParams {
    /* Film (core/film.rs:55-100). */
    int32_t full_resolution[2];
    int32_t cropped_pixel_bounds[4];  /* xmin, ymin, xmax, ymax */
    float filter_radius[2];
    float filter_table[256];          /* 16x16, film.rs:76-89 */
    float max_sample_luminance;
    float scale;
    /* Sampler (samplers/sobol.rs:34-57). sample_bounds = Film::get_sample_bounds(). */
    uint32_t spp;
    int32_t sample_bounds[4];         /* xmin, ymin, xmax, ymax */
    /* Camera (cameras/perspective.rs:40-86). */
    float raster_to_camera[16];
    float camera_to_world[16];
    float lens_radius;
    float focal_distance;
    float shutter_open, shutter_close;
    /* Integrator (integrators/path.rs:225-253). */
    uint32_t max_depth;
    float rr_threshold;
    int32_t pixel_bounds[4];          /* xmin, ymin, xmax, ymax */
    uint32_t light_strategy;          /* PtLightStrategy */
    /* Sharding of the 16x16 sample tiles of integrator.rs:276-283: this call renders the
     * tiles with (tile_index % tile_world) == tile_rank. (1 GPU: rank 0 of 1.) */
    uint32_t tile_rank, tile_world;
    /* Samples per pixel traced per wavefront pass. 0 => the library chooses: as many as fit 60 % of the device's free
     * memory, at most 2^28 paths in flight (~90 GB at 1920x1080 x 128 samples), in passes of equal size. */
    uint32_t spp_per_pass;
    /* 1 => record per-kernel HIP-event timings (see pt_get_kernel_stats); 2 => also exact per-class launch sizes
     * (one extra host sync per iteration). */
    uint32_t profile;
    /* Sampler (api.rs make_sampler): PT_SAMPLER_SOBOL (samplers/sobol.rs) or PT_SAMPLER_HALTON (samplers/halton.rs, the
     * reference's default when a scene names no sampler, api.rs:215-241); "samplepixelcenter" (halton.rs:226). */
    uint32_t sampler_type;
    uint32_t sample_at_pixel_center;
    /* Integrator "path" (integrators/path.rs) or "volpath" (integrators/volpath.rs: medium sampling per segment, transmittance
     * on shadow / MIS rays, Henyey-Greenstein scattering). camera_medium: index into PtSceneDesc.media of the medium the camera
     * sits in (CameraSample rays start in it, perspective.rs:114), PT_NONE for none; read only by PT_INTEGRATOR_VOLPATH. */
    uint32_t integrator;
    uint32_t camera_medium;
} PtRenderParams;

/* Device-side work counters: mirrors of the reference's stat counters
 * (integrator.rs:36, scene.rs:14-15, path.rs:24-25) plus the roofline denominators. */
typedef struct PtCounters {
    uint64_t camera_rays;
    uint64_t intersect_tests;        /* Scene::intersect calls            */
    uint64_t shadow_tests;           /* Scene::intersect_p calls          */
    uint64_t bvh_nodes_visited;      /* pt_set_trace_exact(1): Bounds3f::intersect_p2 executed, the reference's count (bvh.rs:705-814);
                                        default (production traversal): four-wide traversal records fetched (128 B each)   */
    uint64_t triangle_tests;         /* Triangle::intersect(_p) entered via the BVH */
    uint64_t sphere_tests;
    uint64_t zero_radiance_paths_num, zero_radiance_paths_den;
    uint64_t path_length_hist[16];   /* bounces at termination, clamped to 15 */
    uint64_t sanitized_nan, sanitized_negative, sanitized_infinite; /* integrator.rs:350-368 */
    uint64_t film_splats;            /* film pixels touched by add_sample */
    uint64_t wavefront_stages;       /* sum over stages of lanes processed */
    uint64_t reference_asserts;      /* assert!()s of the reference's li that would have fired (path.rs:143,162-163,184,201,213;
                                        volpath.rs:176,194,210,223): the reference panics at the first one, a C ABI cannot -- the
                                        sample is carried on (and sanitised by integrator.rs:350-368's rules); non-zero tells the
                                        host "pbrt-rust would have aborted this render"                                          */
} PtCounters;

typedef struct PtKernelStat {
    char name[32];
    uint64_t launches;
    double total_ms;                 /* HIP-event time on the render stream */
    uint64_t items;                  /* rays / path vertices / pixels processed */
    uint64_t bvh_nodes;              /* trace kernels: Bounds3f::intersect_p2 executed (32 B each);
                                        shade kernels: path-state + queue bytes moved                 */
    uint64_t triangle_tests;         /* trace kernels: triangle packets tested (48 B each)          */
    char kernel[48];                 /* the kernel symbol behind this launch kind, as `rocprofv3 --stats` prints it */
} PtKernelStat;

typedef struct pt_scene pt_scene;
typedef struct pt_multi_scene pt_multi_scene;

/* ---- entry points ------------------------------------------------------------------- */

/* Select the default HIP device of this process and bind the calling thread to it. Every host thread that calls into the
 * library is bound to one device at a time; a pt_scene lives on the device that was bound when it was created and re-binds
 * the calling thread on use, so one process may hold scenes on several devices (or use pt_multi_* below). */
int pt_init(int device_ordinal);
int pt_device_count(int *n_devices);
const char *pt_last_error(void);
/* Which BVH walk the traversal kernels of this process run (all scenes, from the next launch on; also env PT_TRACE_EXACT=1 at pt_init).
 * 0 (default, production): four-wide records built from the same tree -- children visited in the order of BVHAccel::intersect / intersect_p
 * (accelerators/bvh.rs:705-814), so every hit (primitive, t, barycentrics), every film and every other counter is what the two-wide walk gives;
 * PtCounters.bvh_nodes_visited then counts the records fetched. 1: the two-wide walk that tests the reference's nodes one for one, for comparing
 * bvh_nodes_visited with the reference's counter. Returns the previous setting. */
int pt_set_trace_exact(int exact);

/* Limits: at most 2^25 - 1 interior BVH nodes and 2^25 - 1 leaf packets (primitives + instance references) per scene, all
 * accelerators of the scene together (traversal stack entries keep 25-bit references), and the four-wide traversal records (128 B per two
 * interior levels) + the packets (48 B each) below 3.5 GB together (one allocation addressed with 32-bit offsets): 2^25 packets with their
 * records fit; more returns PT_ERR_UNSUPPORTED. */
int pt_scene_create(const PtSceneDesc *desc, pt_scene **out_scene);
void pt_scene_destroy(pt_scene *scene);

/* Size and contents of the accelerator actually used (built or adopted). */
int pt_scene_bvh_info(const pt_scene *scene, uint32_t *n_nodes, uint32_t *n_prims);
int pt_scene_bvh_read(const pt_scene *scene, PtBVHNode *nodes, uint32_t *ordered_prims);

/* Replaces `integ.render(&scene)` (core/api.rs:1746). Accumulates the un-normalised film:
 * film_xyzw[4*(y*W+x)+{0,1,2}] = sum of XYZ contributions (film.rs:153-157) and [+3] = filter
 * weight sum, over the cropped pixel bounds, W = xmax-xmin. `film_xyzw` is ADDED to (the
 * caller zeroes it), so shards from several ranks can be summed. If film_is_device != 0 the
 * pointer is a HIP device pointer on the selected device. */
int pt_render(pt_scene *scene, const PtRenderParams *params, float *film_xyzw, int film_is_device);
/* The samples per pixel per wavefront pass pt_render would use for these parameters on the device as it is now (params->spp_per_pass, or the library's choice from
 * the free memory when that is 0; passes are of equal size). No reference counterpart: the reference renders a tile at a time. */
int pt_pass_size(pt_scene *scene, const PtRenderParams *params, uint32_t *spp_per_pass);

/* Film::write_image normalisation (film.rs:217-258): rgb = max(0, xyz_to_rgb(xyz)/w) * scale.
 * Pure host arithmetic on a host buffer. */
int pt_film_resolve(const float *film_xyzw, uint32_t n_pixels, float scale, float *rgb_out);

/* One process, several GPUs -- the shape of the reference itself: ONE process fans the 16x16 tiles out over its workers
 * (core/integrator.rs:294-296, rayon) and merges them into one Film (integrator.rs:392-396). pt_multi_scene_create replicates the
 * scene on every listed device (an ordinal may repeat: the replicas then share that device); pt_multi_render renders, on one
 * host thread + stream per replica, the tiles with tile_index % (tile_world * n) == tile_rank + i * tile_world on replica i
 * (pt_multi_tile_shard: the caller's own shard split n ways, so it nests inside a multi-process launch), sums the replicas'
 * films onto the first device and ADDS the result to film_xyzw, which is a pointer on device_ordinals[0] (film_is_device != 0) or
 * a host buffer. The merge (Film::merge_film_tile across devices, integrator.rs:392-396): every replica on another device pushes
 * its film into a landing buffer of its own on the first device as soon as it has finished (hipMemcpyPeerAsync from its own host
 * thread: the copies of different replicas run concurrently, one xGMI link each), then ONE kernel sums all films in replica order
 * and adds the sum to film_xyzw. Counters are the sums over the replicas. One pt_multi_scene may be rendered at any film size:
 * the per-replica films and landing buffers grow on demand. */
int pt_multi_scene_create(const PtSceneDesc *desc, const int *device_ordinals, uint32_t n_devices, pt_multi_scene **out_scene);
void pt_multi_scene_destroy(pt_multi_scene *scene);
int pt_multi_render(pt_multi_scene *scene, const PtRenderParams *params, float *film_xyzw, int film_is_device);
int pt_multi_get_counters(const pt_multi_scene *scene, PtCounters *out);
int pt_multi_get_kernel_stats(const pt_multi_scene *scene, uint32_t replica, PtKernelStat *out, uint32_t max_entries, uint32_t *n_out);
/* Wall-clock times of the last pt_multi_render: per replica the duration of its pt_render (its device's busy time) and of its peer
 * copy, and `merge_ms` = from the moment the LAST replica finished rendering to the summed film (copy tail + the sum kernel).
 * render_ms / copy_ms hold max_replicas entries (either may be NULL). */
int pt_multi_get_timing(const pt_multi_scene *scene, double *merge_ms, double *render_ms, double *copy_ms, uint32_t max_replicas);
/* Wall-clock time of pt_multi_scene_create: the whole call (`wall_ms`) and per replica its own scene creation (`replica_ms`, max_replicas entries; either may be
 * NULL). Replica 0 is created first (it builds the top-level BVH), replicas 1.. adopt its tree and are created concurrently, one host thread per replica. */
int pt_multi_get_create_timing(const pt_multi_scene *scene, double *wall_ms, double *replica_ms, uint32_t max_replicas);
/* How replica i's film reaches the first device (decided once, at pt_multi_scene_create): PT_PEER_SAME_DEVICE (it lives there: no copy),
 * PT_PEER_ENABLED (hipDeviceEnablePeerAccess succeeded in both directions: hipMemcpyPeerAsync goes device to device over xGMI) or
 * PT_PEER_STAGED (no peer access: the runtime stages the copy through host memory -- correct, slower). `peer` holds max_replicas entries. */
enum { PT_PEER_SAME_DEVICE = 0, PT_PEER_ENABLED = 1, PT_PEER_STAGED = 2 };
int pt_multi_get_peer_access(const pt_multi_scene *scene, int *peer, uint32_t max_replicas);
/* The tile shard of replica `replica` of `n_replicas` inside the caller's shard (tile_rank of tile_world). Pure host arithmetic. */
void pt_multi_tile_shard(uint32_t tile_rank, uint32_t tile_world, uint32_t replica, uint32_t n_replicas, uint32_t *rank_out, uint32_t *world_out);

int pt_get_counters(const pt_scene *scene, PtCounters *out);
/* Per-kernel statistics of the last pt_render with params->profile != 0. Returns the number
 * of entries written (<= max_entries) through *n_out. */
int pt_get_kernel_stats(const pt_scene *scene, PtKernelStat *out, uint32_t max_entries, uint32_t *n_out);

/* Parity / unit entry points (the kernels behind them are the ones pt_render launches). */
/* BVHAccel::intersect for n rays: out_prim = primitive index (into prims) or PT_NONE,
 * out_t, out_b = hit t and barycentrics (b0,b1,b2) exactly as shapes/triangle.rs:209-213. */
int pt_trace_closest(pt_scene *scene, uint32_t n, const float *origins, const float *dirs,
                     const float *tmax, uint32_t *out_prim, float *out_t, float *out_b);
/* BVHAccel::intersect_p: out_hit[i] = 1 if occluded. */
int pt_trace_any(pt_scene *scene, uint32_t n, const float *origins, const float *dirs,
                 const float *tmax, uint8_t *out_hit);
/* SobolSampler: for each of n (pixel x, pixel y, sample number) triples produce `n_dims`
 * consecutive sample_dimension() values starting at dimension 0 (samplers/sobol.rs:68-86). */
int pt_sobol_samples(const int32_t sample_bounds[4], uint32_t n, const int32_t *pixel_xy,
                     const uint32_t *sample_num, uint32_t n_dims, float *out, uint64_t *out_index);
/* HaltonSampler (samplers/halton.rs:122-165): same contract, dimensions 0/1 are the offsets inside the pixel. */
int pt_halton_samples(const int32_t sample_bounds[4], uint32_t sample_at_pixel_center, uint32_t n, const int32_t *pixel_xy,
                      const uint32_t *sample_num, uint32_t n_dims, float *out, uint64_t *out_index);
/* PerspectiveCamera::generate_ray_differential main ray for n camera samples
 * (pfilm.xy, time, plens.xy) -> origin, direction (cameras/perspective.rs:120-179). */
int pt_camera_rays(const PtRenderParams *params, uint32_t n, const float *camera_samples,
                   float *out_origins, float *out_dirs);
/* Distribution1D (core/sampling.rs:6-85) over func[n], built as the library builds the environment map's rows and the light
 * distributions; for each of n_u numbers u: discrete == 0: sample_continous (sampling.rs:38-64, the routine behind
 * InfiniteAreaLight::sample_li) -> out_x = the sampled value in [0,1), out_pdf, out_offset; discrete != 0: sample_discrete
 * (sampling.rs:66-85, the light choice) -> out_offset, out_pdf (out_x = 0). Twin of the reference's tests/sampling.rs:202-283. */
int pt_dist1d_sample(const float *func, uint32_t n, int discrete, uint32_t n_u, const float *u,
                     float *out_x, float *out_pdf, int32_t *out_offset);

#ifdef __cplusplus
}
#endif
#endif /* MI355PT_H */
