// dev_scene.h -- device-resident scene layout + triangle / bounds tests + BVH traversal.
//   accelerators/bvh.rs:705-814 (traversal order), core/geometry/bounds.rs:559-580 (slab test),
//   shapes/triangle.rs:136-398,400-548 (watertight test, partials, shading geometry).
//
// HBM layout (all read-only during render):
//   nodes      : PtBVHNode[n_nodes]     32 B, reference order (left child = i+1, right = offset)
//   leaf_tris  : TriPacket[n_prims]     48 B, in ordered_prims order -> a leaf is a contiguous run
//                one quad per axis: p0 p1 p2 along x | prim id, along y | shape ref, along z | flags
//   P/N/S/UV, indices, tri_flags, prim_* tables: shading-time data only (never touched by traversal)
#pragma once
#include "dev_math.h"
#include "../../include/mi355pt.h"

namespace ptd {

struct TriPacket {           // 48 bytes, 16-byte aligned, one quad per AXIS: {p0.x p1.x p2.x | prim} {p0.y p1.y p2.y | shape} {p0.z p1.z p2.z | flags}.
    float x[3]; uint32_t prim;   // A traversal lane loads the three quads in the order (kx, ky, kz) of its ray's watertight permutation (triangle.rs:147-160), so the
    float y[3]; uint32_t shape;  // vertices arrive permuted and the test's 18 selects per packet are gone; the three id words arrive permuted with them (tp_aux).
    float z[3]; uint32_t flags;
};
enum { TP_ALPHA = 1u << 7, TP_BOGUS = 1u << 8, TP_SPHERE = 1u << 9, TP_LAST = 1u << 10, TP_INSTANCE = 1u << 11 };  // flags: bits 0-4 = PT_TRI_* bits; LAST = last packet of its leaf
// flags bits 12-15: shade-queue class of the primitive's material (what k_route needs; kernels.h: kNumClasses), bits 16-31: its material index,
// 0xffff = "none or too large: read prim_material" -- so that neither routing nor shading has to chase prim -> material -> class through HBM
constexpr uint32_t kTpClassShift = 12, kTpClassMask = 15u, kTpMatShift = 16, kTpMatNone = 0xffffu;

// Two-wide traversal record (64 B, four dwordx4 loads): one per INTERIOR node of the reference tree, holding the
// bounds of both children, so that a ray fetches once per interior node it enters instead of once per node it tests.
// Child reference: bit 31 set => leaf, low bits = first packet of the leaf in `leaf`; else index of the child's record.
struct WideNode {
    float lmin[3]; float lmax0;
    float lmax12[2]; float rmin01[2];
    float rmin2; float rmax[3];
    uint32_t left_ref, right_ref;
    uint32_t meta;   // bits 0-7: split axis (bvh.rs:671,686)
    uint32_t pad;
};
// Four-wide traversal record of the PRODUCTION traversal (128 B = one L2 line, eight dwordx4 loads): one per interior node N of the reference tree at
// an even interior level -- the boxes of N's grandchildren, structure of arrays, in tree order: slots 0, 1 = the children of N's left child L,
// slots 2, 3 = the children of its right child R; a child of N that is a leaf fills the first slot of its pair with its own box and leaves the second
// empty (bounds +inf / -inf, reference PT_NONE). The ray visits the slots in the order the reference's binary walk reaches them -- near side of N
// first, inside a pair the near side of L (R) first -- so leaves are met in the reference's order and every hit is the reference's hit; only the
// number of boxes tested differs (boxes of L and R themselves are never tested: a child that passes its own test lies inside its parent's box and
// the parent would have passed too). meta: the slot order for each of the eight sign octants of a ray, three bits per octant (scene_create.hip), from the split
// axes of N, L and R (bvh.rs:671,686).
// A lane reads the bound planes of an axis through two dwordx4 loads whose offsets depend on the ray's sign along that axis, so "near" and "far"
// planes arrive sorted and the selects of Bounds3f::intersect_p2's `bounds[dir_is_neg]` indexing cost nothing.
struct QuadNode {
    float lo[3][4];          // [axis][slot]
    float hi[3][4];
    uint32_t ref[4];         // bit 31 set => leaf, low bits = first packet; else index of the child's QuadNode; PT_NONE = empty slot
    uint32_t meta; uint32_t pad[3];
};
static_assert(sizeof(QuadNode) == 128, "QuadNode is one 128-byte line");
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kRefMask = 0x01ffffffu;   // the two-wide (exact) walk: 25 bits, 33 M records / packets -- its stack entries pack the 6 bits of a skipped-entry count above them
constexpr uint32_t kRefMaskQuad = 0x7fffffffu;   // the production walk: 31 bits; what bounds a scene there is the pool of records + packets, addressed in 16-byte quads (64 GB)

// TransformedPrimitive (primitive.rs:40-88) on device: transforms + entry into the object's BVH
struct DevInstance {
    float world_to_instance[16], instance_to_world[16];
    float root_min[3], root_max[3];   // bounds of the object's BVH root (unused when `single`)
    uint32_t root_ref;                // wide-record index or kLeafBit | first packet
    uint32_t single;                  // object holds one primitive: no BVH, no root test (api.rs:1692)
    uint32_t identity;                // Transform::is_identity(instance_to_world) (primitive.rs:73)
    uint32_t root_ref4;               // the same root in the four-wide records (QuadNode index or kLeafBit | first packet)
};

// media/grid.rs GridDensityMedium: what the device needs besides the PtMedium record
struct DevGridAux { const float *density; float sigma_t, inv_max_density; };

// core/bssrdf.rs:241-268 BSSRDFTable (device copy of PtBSSRDFTable)
struct DevBssTable { int n_rho, n_radius; const float *rho_samples, *radius_samples, *profile, *rhoeff, *profile_cdf; };

// MIPMap pyramid on the device (PtImage + per-level offsets in floats)
struct DevImage { uint32_t width, height, n_levels, channels; const float *texels; uint32_t level_offset[16]; };

struct DeviceScene {
    const WideNode *wide; uint32_t n_nodes;   // n_nodes = nodes of the reference tree (0 => empty scene)
    float root_min[3], root_max[3]; uint32_t root_ref;  // the root's own bounds and reference
    const QuadNode *quad; uint32_t root_ref4; uint32_t leaf_off, pool_quads;   // (leaf_off: offset of `leaf` from `quad` in 16-byte quads -- one allocation of pool_quads quads, < 64 GB)
    //         // four-wide records of the same trees (production traversal) and the root's reference among them
    const TriPacket *leaf; uint32_t n_prims;
    const float *P; const float *N; const float *S; const float *UV;
    const uint32_t *indices; const uint8_t *tri_flags; uint32_t n_triangles;
    const PtSphere *spheres; uint32_t n_spheres;
    const DevInstance *instances; uint32_t n_instances;
    const uint32_t *prim_shape; const uint32_t *prim_material; const uint32_t *prim_light;
    const PtMaterial *materials; uint32_t n_materials;
    const PtLight *lights; uint32_t n_lights;
    const float *light_area;            // per light: Shape::area() of its primitive
    // per light, six quads {p0, flags} {p1, triangle} {p2, area} {Lemit, two_sided} {n_sample, 1 / area} {n_pdf, -}: what sampling a TRIANGLE
    // area light and evaluating its pdf read, in one place instead of behind the chain lights -> prim_shape -> indices -> P, with the
    // per-light constants of both computed once by k_light_area (the same instruction sequences the shade kernels ran per vertex).
    // flags: the triangle's PT_TRI_* byte | 0x100 = the record is valid | 0x200 = Triangle::intersect rejects every hit (degenerate
    // partials, triangle.rs:236-264); n_sample = Triangle::sample's normal before the face-forward to interpolated normals
    // (triangle.rs:566-577), n_pdf = the interaction normal Triangle::intersect leaves without a shape (triangle.rs:283-300)
    const float4 *light_rec;
    const uint32_t *infinite_lights; uint32_t n_infinite;
    const uint8_t *mat_class;           // per material: shade-queue class
    const DevBssTable *bss_tables; uint32_t n_bss_tables;   // subsurface materials (row a23)
    const PtMedium *media; uint32_t n_media;                 // media + per-primitive MediumInterface (volpath, dev_medium.h)
    uint32_t has_shells;                                     // a primitive without a material: a medium-interface shell (api.rs:597; volpath's transmittance loops walk through it)
    const DevGridAux *grid_aux; uint32_t has_grid;           // per medium: GridDensityMedium's device density array, sigma_t, 1 / max density (grid.rs:46-60); any grid medium in the scene
    const uint32_t *prim_med_in; const uint32_t *prim_med_out;
    // textures (8f-1): nodes, one postfix program per node (tex_prog[tex_prog_offset[i] .. tex_prog_offset[i+1])), images
    const PtTexture *textures; uint32_t n_textures; const uint32_t *tex_prog_offset; const uint32_t *tex_prog;
    const DevImage *images; const float *ewa_lut;
    const int32_t *tri_alpha; const int32_t *tri_shadow_alpha;   // per triangle float-texture index or -1 (NULL: no masks)
    // env map
    uint32_t env_w, env_h; const float *env_texels;
    const float *env_func; const float *env_cdf; const float *env_func_int;  // conditional rows (2h x 2w [+1]), marginal appended
    const float *env_marg_func; const float *env_marg_cdf; float env_marg_int;
    float wb_min[3], wb_max[3];
    float world_radius; float world_center[3];
};

struct Ray { V3 o, d; float t_max; };

struct Hit { uint32_t prim; float t, b0, b1, b2; };

PT_DEV V3 ld3(const float *p, uint32_t i) { return V3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }

// Watertight ray-triangle test shared by intersect / intersect_p (triangle.rs:136-233 == :400-495).
// The part of the test that depends on the ray only (triangle.rs:147-160): permutation axis and shear constants. The traversal
// kernel evaluates it once per ray instead of once per triangle.
struct TriRay { int kz; float Sx, Sy, Sz; };
PT_DEV TriRay tri_ray_setup(V3 rd) {
    TriRay r;
    r.kz = max_dimension(vabs(rd));
    // permute(kx, ky, kz) with kx = kz+1 mod 3, ky = kx+1 mod 3
    float dx, dy, dz;
    if (r.kz == 0) { dx = rd.y; dy = rd.z; dz = rd.x; }
    else if (r.kz == 1) { dx = rd.z; dy = rd.x; dz = rd.y; }
    else { dx = rd.x; dy = rd.y; dz = rd.z; }
    r.Sx = -dx / dz; r.Sy = -dy / dz; r.Sz = 1.0f / dz;
    return r;
}
// permute(v, kx, ky, kz) of triangle.rs:152-160 for the ray's kz
PT_DEV V3 tri_permute(V3 v, int kz) { const bool k0 = kz == 0, k1 = kz == 1; return V3(k0 ? v.y : (k1 ? v.z : v.x), k0 ? v.z : (k1 ? v.x : v.y), k0 ? v.x : (k1 ? v.y : v.z)); }
// word `axis` (0 prim, 1 shape, 2 flags) of a packet whose quads were loaded in the order (kx, ky, kz)
PT_DEV uint32_t tp_aux(uint32_t w0, uint32_t w1, uint32_t w2, int kz, int axis) {   // quad 0 holds axis kx = kz + 1, quad 1 axis ky = kz + 2 (mod 3), quad 2 axis kz
    const bool k0 = kz == 0, k1 = kz == 1;
    if (axis == 2) return k0 ? w1 : (k1 ? w0 : w2);
    if (axis == 1) return k0 ? w0 : (k1 ? w2 : w1);
    return k0 ? w2 : (k1 ? w1 : w0);
}
// The test proper on vertices already translated to the ray origin and permuted (triangle.rs:161-233 == :424-495).
PT_DEV bool tri_hit_core(V3 p0t, V3 p1t, V3 p2t, const TriRay &tr, float t_max, float &t, float &b0, float &b1, float &b2) {
    const float Sx = tr.Sx, Sy = tr.Sy, Sz = tr.Sz;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {  // f64 fallback, triangle.rs:178-189
        double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    // the three rejections of triangle.rs:191-208 folded into one predicate and one exit (same comparisons): with tens of lanes
    // testing different triangles an early return rarely skips anything for the wave
    const bool mixed = ((e0 < 0.0f) | (e1 < 0.0f) | (e2 < 0.0f)) & ((e0 > 0.0f) | (e1 > 0.0f) | (e2 > 0.0f));
    float det = e0 + e1 + e2;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    float tscaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    const float tmd = t_max * det;
    const bool out_neg = (det < 0.0f) & ((tscaled >= 0.0f) | (tscaled < tmd));
    const bool out_pos = (det > 0.0f) & ((tscaled <= 0.0f) | (tscaled >= tmd));
    if (mixed | (det == 0.0f) | out_neg | out_pos) return false;
    float invdet = 1.0f / det;
    b0 = e0 * invdet; b1 = e1 * invdet; b2 = e2 * invdet;
    t = tscaled * invdet;
    float maxzt = max_component(vabs(V3(p0t.z, p1t.z, p2t.z)));
    float deltaz = gammaf(3) * maxzt;
    float maxxt = max_component(vabs(V3(p0t.x, p1t.x, p2t.x)));
    float maxyt = max_component(vabs(V3(p0t.y, p1t.y, p2t.y)));
    float deltax = gammaf(5) * (maxxt + maxzt);
    float deltay = gammaf(5) * (maxyt + maxzt);
    float deltae = 2.0f * (gammaf(2) * maxxt * maxyt + deltay * maxxt + deltax * maxyt);
    float maxe = max_component(vabs(V3(e0, e1, e2)));
    float deltat = 3.0f * (gammaf(3) * maxe * maxzt + deltae * maxzt + deltaz * maxe) * fabsf(invdet);
    if (t <= deltat) return false;
    return true;
}
// Watertight ray-triangle test shared by intersect / intersect_p (triangle.rs:136-233 == :400-495): translate, permute, tri_hit_core.
PT_DEV bool tri_hit_params(V3 p0, V3 p1, V3 p2, V3 ro, const TriRay &tr, float t_max, float &t, float &b0, float &b1, float &b2) {
    return tri_hit_core(tri_permute(p0 - ro, tr.kz), tri_permute(p1 - ro, tr.kz), tri_permute(p2 - ro, tr.kz), tr, t_max, t, b0, b1, b2);
}

PT_DEV bool tri_hit_params(V3 p0, V3 p1, V3 p2, V3 ro, V3 rd, float t_max, float &t, float &b0, float &b1, float &b2) {
    return tri_hit_params(p0, p1, p2, ro, tri_ray_setup(rd), t_max, t, b0, b1, b2);
}

PT_DEV void tri_uvs(const DeviceScene &s, uint32_t tri, uint32_t i0, uint32_t i1, uint32_t i2, P2 uv[3]) {  // triangle.rs:109-115
    if (s.tri_flags[tri] & PT_TRI_HAS_UV) {
        uv[0] = P2(s.UV[2 * i0], s.UV[2 * i0 + 1]); uv[1] = P2(s.UV[2 * i1], s.UV[2 * i1 + 1]); uv[2] = P2(s.UV[2 * i2], s.UV[2 * i2 + 1]);
    } else { uv[0] = P2(0.0f, 0.0f); uv[1] = P2(1.0f, 0.0f); uv[2] = P2(1.0f, 1.0f); }
}
// dpdu/dpdv + the "intersection is bogus" rejection (triangle.rs:236-264). false => degenerate triangle.
PT_DEV bool tri_partials(V3 p0, V3 p1, V3 p2, const P2 uv[3], V3 &dpdu, V3 &dpdv) {
    float duv02x = uv[0].x - uv[2].x, duv02y = uv[0].y - uv[2].y;
    float duv12x = uv[1].x - uv[2].x, duv12y = uv[1].y - uv[2].y;
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = duv02x * duv12y - duv02y * duv12x;
    bool degenerateuv = fabsf(determinant) < 1.0e-8f;
    dpdu = V3(); dpdv = V3();
    if (!degenerateuv) {
        float invdet = 1.0f / determinant;
        dpdu = (dp02 * duv12y - dp12 * duv02y) * invdet;
        dpdv = (dp02 * -duv12x + dp12 * duv02x) * invdet;
    }
    if (degenerateuv || length_squared(cross(dpdu, dpdv)) == 0.0f) {
        V3 ng = cross(p2 - p0, p1 - p0);
        if (length_squared(ng) == 0.0f) return false;
        coordinate_system(normalize(ng), dpdu, dpdv);
    }
    return true;
}

// What Triangle::intersect leaves in `isect` (triangle.rs:266-392 + interaction.rs:186-249).
struct SurfaceInteraction {
    V3 p, p_error, n, wo;
    V3 dpdu;             // == shading.dpdu unless the mesh has N/S
    V3 sh_n, sh_dpdu;
    V3 dpdv; P2 uv;      // read by texture evaluation only (dead code elsewhere)
    V3 sh_dpdv, sh_dndu, sh_dndv;   // shading.dpdv / dndu / dndv: bump mapping only
    bool has_shape, shape_flip;     // SurfaceInteraction.shape is Some (triangles) / its reverse_orientation ^ swaps_handedness
    uint32_t prim;
};
// `with_shape` = the `s: Option<Arc<Shapes>>` argument (None inside Shape::pdf_wi, shape.rs:72).
// tri_fill_from: the triangle's vertices and flag byte are already at hand (the 48-byte TriPacket the traversal hit); the index
// triple is fetched only for meshes with per-vertex N / S / UV.
PT_DEVX void tri_fill_from(const DeviceScene &s, uint32_t tri, uint32_t fl, V3 p0, V3 p1, V3 p2, V3 ray_d, float b0, float b1, float b2, bool with_shape, SurfaceInteraction &si) {
    uint32_t i0 = 0, i1 = 0, i2 = 0;
    if (fl & (PT_TRI_HAS_N | PT_TRI_HAS_S | PT_TRI_HAS_UV)) { i0 = s.indices[3 * tri]; i1 = s.indices[3 * tri + 1]; i2 = s.indices[3 * tri + 2]; }
    P2 uv[3];   // triangle.rs:109-115
    if (fl & PT_TRI_HAS_UV) { uv[0] = P2(s.UV[2 * i0], s.UV[2 * i0 + 1]); uv[1] = P2(s.UV[2 * i1], s.UV[2 * i1 + 1]); uv[2] = P2(s.UV[2 * i2], s.UV[2 * i2 + 1]); }
    else { uv[0] = P2(0.0f, 0.0f); uv[1] = P2(1.0f, 0.0f); uv[2] = P2(1.0f, 1.0f); }
    V3 dpdu, dpdv; tri_partials(p0, p1, p2, uv, dpdu, dpdv);
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    float xabs = fabsf(b0 * p0.x) + fabsf(b1 * p1.x) + fabsf(b2 * p2.x);
    float yabs = fabsf(b0 * p0.y) + fabsf(b1 * p1.y) + fabsf(b2 * p2.y);
    float zabs = fabsf(b0 * p0.z) + fabsf(b1 * p1.z) + fabsf(b2 * p2.z);
    si.p_error = V3(xabs, yabs, zabs) * gammaf(7);
    si.p = p0 * b0 + p1 * b1 + p2 * b2;
    si.dpdu = dpdu; si.sh_dpdu = dpdu; si.dpdv = dpdv; si.sh_dpdv = dpdv;
    si.sh_dndu = V3(0.0f, 0.0f, 0.0f); si.sh_dndv = V3(0.0f, 0.0f, 0.0f);
    si.uv = P2(uv[0].x * b0 + uv[1].x * b1 + uv[2].x * b2, uv[0].y * b0 + uv[1].y * b1 + uv[2].y * b2);   // triangle.rs:266-268
    bool flip = ((fl & PT_TRI_REVERSE_ORIENTATION) != 0) != ((fl & PT_TRI_SWAPS_HANDEDNESS) != 0);
    si.has_shape = with_shape; si.shape_flip = flip;
    V3 nn = normalize(cross(dp02, dp12));
    si.n = nn; si.sh_n = nn;
    si.wo = -ray_d;  // triangle.rs:296
    if (flip) { si.n = -nn; si.sh_n = -nn; }
    if (fl & (PT_TRI_HAS_N | PT_TRI_HAS_S)) {
        V3 ns;
        if (fl & PT_TRI_HAS_N) {
            ns = ld3(s.N, i0) * b0 + ld3(s.N, i1) * b1 + ld3(s.N, i2) * b2;
            if (length_squared(ns) > 0.0f) ns = normalize(ns); else ns = si.n;
        } else ns = si.n;
        V3 ss;
        if (fl & PT_TRI_HAS_S) {
            ss = ld3(s.S, i0) * b0 + ld3(s.S, i1) * b1 + ld3(s.S, i2) * b2;
            if (length_squared(ss) > 0.0f) ss = normalize(ss); else ss = normalize(si.dpdu);
        } else ss = normalize(si.dpdu);
        V3 ts = cross(ss, ns);
        if (length_squared(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
        else coordinate_system(ns, ss, ts);
        if (fl & PT_TRI_HAS_N) {  // dndu / dndv, triangle.rs:349-386 (read by bump mapping only)
            const float duv02x = uv[0].x - uv[2].x, duv02y = uv[0].y - uv[2].y, duv12x = uv[1].x - uv[2].x, duv12y = uv[1].y - uv[2].y;
            const V3 n0 = ld3(s.N, i0), n1 = ld3(s.N, i1), n2 = ld3(s.N, i2);
            const V3 dn1 = n0 - n2, dn2 = n1 - n2;
            const float det = duv02x * duv12y - duv02y * duv12x;
            if (fabsf(det) < 1.0e-8f) {
                const V3 dn = cross(n2 - n0, n1 - n0);
                if (length_squared(dn) != 0.0f) coordinate_system(dn, si.sh_dndu, si.sh_dndv);
            } else {
                const float invdet = 1.0f / det;
                si.sh_dndu = (dn1 * duv12y - dn2 * duv02y) * invdet;
                si.sh_dndv = (dn1 * -duv12x + dn2 * duv02x) * invdet;
            }
        }
        if (fl & PT_TRI_REVERSE_ORIENTATION) ts = -ts;
        si.sh_n = normalize(cross(ss, ts));  // set_shading_geometry(.., true), interaction.rs:228-249
        if (with_shape) {
            if (flip) si.sh_n = -si.sh_n;
            si.n = face_forward(si.n, si.sh_n);
        }
        si.sh_dpdu = ss; si.sh_dpdv = ts;
    }
}
PT_DEVX void tri_fill_interaction(const DeviceScene &s, uint32_t tri, V3 ray_d, float b0, float b1, float b2, bool with_shape, SurfaceInteraction &si) {
    const uint32_t i0 = s.indices[3 * tri], i1 = s.indices[3 * tri + 1], i2 = s.indices[3 * tri + 2];
    tri_fill_from(s, tri, s.tri_flags[tri], ld3(s.P, i0), ld3(s.P, i1), ld3(s.P, i2), ray_d, b0, b1, b2, with_shape, si);
}
PT_DEV float tri_area(V3 p0, V3 p1, V3 p2) { return 0.5f * length(cross(p1 - p0, p2 - p0)); }  // triangle.rs:550-554

// Bounds3f::intersect_p2 (bounds.rs:559-580). bmin/bmax passed as two float4-ish groups.
PT_DEV bool slab_test(const float bmin[3], const float bmax[3], V3 ro, V3 inv_dir, bool nx, bool ny, bool nz, float ray_tmax) {
    float tmin = ((nx ? bmax[0] : bmin[0]) - ro.x) * inv_dir.x;
    float tmax = ((nx ? bmin[0] : bmax[0]) - ro.x) * inv_dir.x;
    float tymin = ((ny ? bmax[1] : bmin[1]) - ro.y) * inv_dir.y;
    float tymax = ((ny ? bmin[1] : bmax[1]) - ro.y) * inv_dir.y;
    const float k = 1.0f + 2.0f * gammaf(3);
    tmax *= k; tymax *= k;
    if (tmin > tymax || tymin > tmax) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = ((nz ? bmax[2] : bmin[2]) - ro.z) * inv_dir.z;
    float tzmax = ((nz ? bmin[2] : bmax[2]) - ro.z) * inv_dir.z;
    tzmax *= k;
    if (tmin > tzmax || tzmin > tmax) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    return (tmin < ray_tmax) && (tmax > 0.0f);
}
// The same arithmetic split in two: the part that does not depend on ray.t_max (returned bool: slabs overlap and
// tmax > 0) and the entry distance tmin_out; intersect_p2 == slab_geo(..) && tmin_out < ray.t_max.
PT_DEV bool slab_geo(const float bmin[3], const float bmax[3], V3 ro, V3 inv_dir, bool nx, bool ny, bool nz, float &tmin_out) {
    // Straight-line form of bounds.rs:559-580 (same operations and comparisons, the early returns folded into one predicate): two
    // boxes are tested per traversal step by 64 lanes, an early return never skips work for the wave, it only costs branches.
    // tmin_out is meaningful only when the result is true.
    float tmin = ((nx ? bmax[0] : bmin[0]) - ro.x) * inv_dir.x;
    float tmax = ((nx ? bmin[0] : bmax[0]) - ro.x) * inv_dir.x;
    const float tymin = ((ny ? bmax[1] : bmin[1]) - ro.y) * inv_dir.y;
    float tymax = ((ny ? bmin[1] : bmax[1]) - ro.y) * inv_dir.y;
    const float tzmin = ((nz ? bmax[2] : bmin[2]) - ro.z) * inv_dir.z;
    float tzmax = ((nz ? bmin[2] : bmax[2]) - ro.z) * inv_dir.z;
    const float k = 1.0f + 2.0f * gammaf(3);
    tmax *= k; tymax *= k; tzmax *= k;
    const bool miss_xy = (tmin > tymax) | (tymin > tmax);
    tmin = (tymin > tmin) ? tymin : tmin;
    tmax = (tymax < tmax) ? tymax : tmax;
    const bool miss_z = (tmin > tzmax) | (tzmin > tmax);
    tmin = (tzmin > tmin) ? tzmin : tmin;
    tmax = (tzmax < tmax) ? tzmax : tmax;
    tmin_out = tmin;
    return !(miss_xy | miss_z) & (tmax > 0.0f);
}

}  // namespace ptd
