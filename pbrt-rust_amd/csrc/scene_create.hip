// scene_create.hip -- pt_scene_create / destroy / bvh read-back: validation of the caller's arrays, accelerators, traversal records, uploads (host_common.h has the map).
#include "host_common.h"

namespace pth {
// Shade class of a material = the kernel its vertices are shaded by (kernels.h: kNumClasses): by the number of BxDFs the material
// can produce, decided from its constant parameters (a textured parameter can take any value).
uint8_t material_class(const PtMaterial &m, bool specialise, bool untextured) {
    // `specialise`: hand out the classes of the lobe-SET kernels (metal, plastic-like, uber; kernels.h; PT_SHADE_SPECIALISE != 0) -- textured scenes too since round 6 (ADVICE r5:
    // one texture anywhere used to send every metal / plastic / uber vertex of the scene to the general kernels); the smooth-subsurface class in untextured scenes only
    auto textured = [&](int slot) { return m.tex[slot] >= 0; };
    auto black = [](const float c[3]) { return !(c[0] > 0.0f) && !(c[1] > 0.0f) && !(c[2] > 0.0f); };   // .clamps(0, inf).is_black()
    switch (m.type) {
    case PT_MAT_MATTE: return 0;
    case PT_MAT_MIRROR: return (uint8_t)kSpecClass;
    case PT_MAT_METAL: return specialise ? (uint8_t)kMetalClass : 1;
    case PT_MAT_SUBSTRATE: return 1;
    case PT_MAT_GLASS:   // glass.rs:57-92: one FresnelSpecular lobe when both roughnesses are 0, else up to two microfacet lobes
        if (textured(PT_MP_U_ROUGHNESS) || textured(PT_MP_V_ROUGHNESS)) return 2;
        return (m.u_roughness == 0.0f && m.v_roughness == 0.0f) ? (uint8_t)kSpecClass : 2;
    case PT_MAT_PLASTIC: return specialise ? (uint8_t)kPlasticClass : 2;
    case PT_MAT_UBER: {   // uber.rs:40-106: without specular reflection / transmission and fully opaque it is Lambertian + microfacet
        const bool opaque = !textured(PT_MP_OPACITY) && m.opacity[0] >= 1.0f && m.opacity[1] >= 1.0f && m.opacity[2] >= 1.0f;
        const bool no_spec = !textured(PT_MP_KR) && !textured(PT_MP_KT) && black(m.kr) && black(m.kt);
        if (opaque && no_spec) return specialise ? (uint8_t)kPlasticClass : 2;
        return specialise ? (uint8_t)kUberClass : 3;
    }
    case PT_MAT_SUBSURFACE:   // constant zero roughness: one FresnelSpecular lobe + the BSSRDF (subsurface.rs:84-109)
        if (specialise && untextured && m.u_roughness == 0.0f && m.v_roughness == 0.0f && !textured(PT_MP_U_ROUGHNESS) && !textured(PT_MP_V_ROUGHNESS)) return (uint8_t)kSssClass;
        return 3;
    default: return 3;
    }
}

// Distribution1D::new on the host (sampling.rs:12-34) for the uniform / power strategies and the env map.
void dist1d(const std::vector<float> &func, std::vector<float> &cdf, float &func_int) {
    size_t n = func.size();
    cdf.assign(n + 1, 0.0f);
    for (size_t i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / (float)n;
    func_int = cdf[n];
    if (func_int == 0.0f) { for (size_t i = 1; i < n + 1; ++i) cdf[i] = (float)i / (float)n; }
    else { for (size_t i = 1; i < n + 1; ++i) cdf[i] /= func_int; }
}

}  // namespace pth

extern "C" {

int pt_scene_create(const PtSceneDesc *d, pt_scene **out) {
    if (!d || !out) return fail(PT_ERR_INVALID_ARG, "null argument");
    if (d->n_prims == 0 || !d->prim_shape || !d->prim_material || !d->prim_light) return fail(PT_ERR_INVALID_ARG, "scene has no primitives");
    if (d->n_triangles && (!d->P || !d->indices)) return fail(PT_ERR_INVALID_ARG, "triangle arrays missing");
    if (d->n_spheres && !d->spheres) return fail(PT_ERR_INVALID_ARG, "sphere array missing");
    for (uint32_t i = 0; i < 3 * d->n_triangles; ++i) if (d->indices[i] >= d->n_vertices) return fail(PT_ERR_INVALID_ARG, "vertex index out of range");
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        uint32_t s = d->prim_shape[i];
        const uint32_t kind = s >> 30, idx = s & 0x3fffffffu;
        if (!((kind == PT_SHAPE_TRIANGLE && idx < d->n_triangles) || (kind == PT_SHAPE_SPHERE && idx < d->n_spheres))) return fail(PT_ERR_INVALID_ARG, "primitive shape reference out of range");
        if (d->prim_material[i] != PT_NONE && d->prim_material[i] >= d->n_materials) return fail(PT_ERR_INVALID_ARG, "material index out of range");
        if (d->prim_light[i] != PT_NONE && d->prim_light[i] >= d->n_lights) return fail(PT_ERR_INVALID_ARG, "light index out of range");
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const PtMaterial &m = d->materials[i];
        if (m.type != PT_MAT_SUBSURFACE) continue;
        if (!d->bssrdf_tables || m.bssrdf_table >= d->n_bssrdf_tables) return fail(PT_ERR_INVALID_ARG, "subsurface material without a BSSRDF table");
        const PtBSSRDFTable &t = d->bssrdf_tables[m.bssrdf_table];
        if (t.n_rho < 2 || t.n_radius < 2 || !t.rho_samples || !t.radius_samples || !t.profile || !t.rhoeff || !t.profile_cdf) return fail(PT_ERR_INVALID_ARG, "incomplete BSSRDF table");
    }
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const PtTexture &t = d->textures[i];
        if (t.type > PT_TEX_DOTS) return fail(PT_ERR_UNSUPPORTED, "texture type not implemented");
        for (int k = 0; k < 3; ++k) if (t.child[k] >= (int32_t)d->n_textures) return fail(PT_ERR_INVALID_ARG, "texture child index out of range");
        if (t.type == PT_TEX_IMAGEMAP) {
            if (!d->images || t.image >= d->n_images) return fail(PT_ERR_INVALID_ARG, "image texture without an image");
            const PtImage &im = d->images[t.image];
            auto pow2 = [](uint32_t v) { return v && !(v & (v - 1)); };
            if (!pow2(im.width) || !pow2(im.height) || !im.texels || (im.channels != 1 && im.channels != 3) || im.n_levels == 0 || im.n_levels > 16)
                return fail(PT_ERR_INVALID_ARG, "PtImage must be a power-of-two MIPMap pyramid with 1 or 3 channels");
            if (t.wrap > PT_WRAP_BLACK) return fail(PT_ERR_UNSUPPORTED, "ImageWrap::Clamp is not implemented");
            if (!t.trilinear && !d->ewa_weight_lut) return fail(PT_ERR_INVALID_ARG, "EWA image texture without ewa_weight_lut");
        }
        if ((t.type == PT_TEX_SCALE || t.type == PT_TEX_CHECKERBOARD2D || t.type == PT_TEX_CHECKERBOARD3D || t.type == PT_TEX_DOTS) && (t.child[0] < 0 || t.child[1] < 0)) return fail(PT_ERR_INVALID_ARG, "texture node needs two children");
        if (t.type == PT_TEX_MIX && (t.child[0] < 0 || t.child[1] < 0 || t.child[2] < 0)) return fail(PT_ERR_INVALID_ARG, "mix texture needs three children");
    }
    for (const int32_t *arr : {d->tri_alpha, d->tri_shadow_alpha})
        if (arr) for (uint32_t i = 0; i < d->n_triangles; ++i) if (arr[i] >= (int32_t)d->n_textures) return fail(PT_ERR_INVALID_ARG, "alpha-mask texture index out of range");
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const PtMaterial &m = d->materials[i];
        if (m.type > PT_MAT_DISNEY) return fail(PT_ERR_INVALID_ARG, "unknown material type");
        if (m.type == PT_MAT_DISNEY) {   // disney.rs:741-836: BxDFs the parameters can produce; the shade class holds five
            const bool thin = m.disney_thin != 0;
            const float dw = (1.0f - m.disney[PT_DS_METALLIC]) * (1.0f - m.disney[PT_DS_SPECTRANS]);
            int n = 1 + (m.disney[PT_DS_CLEARCOAT] > 0.0f) + (m.disney[PT_DS_SPECTRANS] > 0.0f) + (thin ? 1 : 0);
            if (dw > 0.0f) n += (thin ? 2 : 1) + 1 + (m.disney[PT_DS_SHEEN] > 0.0f);
            if (n > 5) return fail(PT_ERR_UNSUPPORTED, "disney material with more than 5 BxDFs");
            if (disney_has_bssrdf(m) && d->n_textures && m.tex[PT_MP_KD] >= 0) return fail(PT_ERR_UNSUPPORTED, "disney: a textured color together with scatterdistance");
            if (disney_has_bssrdf(m) && (m.disney_scatter[0] <= 0.0f || m.disney_scatter[1] <= 0.0f || m.disney_scatter[2] <= 0.0f)) return fail(PT_ERR_INVALID_ARG, "disney: scatterdistance must be positive in every channel");
        }
        if (m.type == PT_MAT_MIX) {   // mix.rs:25-50: two plain materials whose lobes fit the five-lobe shade class together
            auto lobes = [](const PtMaterial &q) { switch (q.type) { case PT_MAT_GLASS: return 2; case PT_MAT_PLASTIC: return 2; case PT_MAT_UBER: return 5; case PT_MAT_TRANSLUCENT: return 4; case PT_MAT_DISNEY: return 5; default: return 1; } };
            int total = 0;
            for (int k = 0; k < 2; ++k) {
                if (m.mix[k] >= d->n_materials) return fail(PT_ERR_INVALID_ARG, "mix material index out of range");
                const PtMaterial &q = d->materials[m.mix[k]];
                if (q.type == PT_MAT_MIX || q.type == PT_MAT_SUBSURFACE || disney_has_bssrdf(q)) return fail(PT_ERR_UNSUPPORTED, "mix of mix / subsurface materials");
                total += lobes(q);
            }
            if (total > 5) return fail(PT_ERR_UNSUPPORTED, "mix material with more than 5 BxDFs");
        }
        for (int k = 0; k < 16; ++k) {
            if (d->n_textures == 0 && m.tex[k] > 0) return fail(PT_ERR_INVALID_ARG, "material references a texture but the scene has none");
            if (d->n_textures && m.tex[k] >= (int32_t)d->n_textures) return fail(PT_ERR_INVALID_ARG, "material texture index out of range");
        }
    }
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        const PtLight &L = d->lights[i];
        if (L.type == PT_LIGHT_DIFFUSE_AREA && L.prim >= d->n_prims) return fail(PT_ERR_INVALID_ARG, "area light primitive out of range");
        if (L.type == PT_LIGHT_INFINITE && !d->env_texels) return fail(PT_ERR_INVALID_ARG, "infinite light without env_texels");
    }
    int st = ensure_device();
    if (st) return st;
    pt_scene *sc = new pt_scene();
    sc->device = g_device;
    auto bail = [&](int code) { pt_scene_destroy(sc); return code; };
    DeviceScene &ds = sc->ds;
    // ---- accelerators: one BVH per multi-primitive object (api.rs:1692-1700) + the top-level BVH (adopted or built)
    const bool instanced = d->n_instances > 0 && d->top_refs && d->n_top > 0;
    if (d->n_instances && !instanced) return bail(fail(PT_ERR_INVALID_ARG, "instances given without top_refs"));
    const uint32_t n_top = instanced ? d->n_top : d->n_prims;
    auto prim_bound = [&](uint32_t i, pth::PrimBound &out) {
        if ((d->prim_shape[i] >> 30) == PT_SHAPE_SPHERE) {  // Shape::world_bound = transform_bounds(object_bound) (shape.rs:23-25, sphere.rs:53-57, transform.rs:592-605)
            const PtSphere &S = d->spheres[d->prim_shape[i] & 0x3fffffffu];
            M4 o2w; std::memcpy(o2w.m, S.object_to_world, 64);
            const float lo[3] = {-S.radius, -S.radius, S.z_min}, hi[3] = {S.radius, S.radius, S.z_max};
            const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 1, 1}, {1, 1, 0}, {1, 0, 1}, {1, 1, 1}};
            for (int c = 0; c < 8; ++c) {
                V3 p = xf_point(o2w, V3(corner[c][0] ? hi[0] : lo[0], corner[c][1] ? hi[1] : lo[1], corner[c][2] ? hi[2] : lo[2]));
                const float pc[3] = {p.x, p.y, p.z};
                for (int k = 0; k < 3; ++k) { out.lo[k] = c ? std::fmin(out.lo[k], pc[k]) : pc[k]; out.hi[k] = c ? std::fmax(out.hi[k], pc[k]) : pc[k]; }
            }
            return;
        }
        uint32_t tri = d->prim_shape[i] & 0x3fffffffu;  // Triangle::world_bound (triangle.rs:130-134)
        const float *a = d->P + 3 * (size_t)d->indices[3 * tri], *b = d->P + 3 * (size_t)d->indices[3 * tri + 1], *c = d->P + 3 * (size_t)d->indices[3 * tri + 2];
        for (int k = 0; k < 3; ++k) { out.lo[k] = std::fmin(std::fmin(a[k], b[k]), c[k]); out.hi[k] = std::fmax(std::fmax(a[k], b[k]), c[k]); }
    };
    const uint32_t maxp = d->max_node_prims ? d->max_node_prims : 4;
    if (d->split_method > PT_SPLIT_HLBVH) return bail(fail(PT_ERR_INVALID_ARG, "unknown split_method"));
    // BVHAccel::new (bvh.rs:145-198): SAH on the host (the reference's tree) or HLBVH on the device (gpu_bvh.hip)
    auto build_accel = [&](const std::vector<pth::PrimBound> &pb, std::vector<PtBVHNode> &nodes, std::vector<uint32_t> &ordered) -> int {
        if (d->split_method != PT_SPLIT_HLBVH) { pth::build_sah_bvh(pb, maxp, nodes, ordered); return PT_OK; }
        const char *msg = "HLBVH build failed";
        if (pth::build_hlbvh_gpu(pb, maxp, nodes, ordered, &msg)) return fail(PT_ERR_HIP, msg);
        return PT_OK;
    };
    struct ObjAccel { std::vector<PtBVHNode> nodes; std::vector<uint32_t> ordered; };
    std::vector<ObjAccel> obj(instanced ? d->n_objects : 0);
    if (instanced) {
        for (uint32_t o = 0; o < d->n_objects; ++o) {
            const PtObject &O = d->objects[o];
            if (O.n_prims == 0 || (uint64_t)O.first_prim + O.n_prims > d->n_prims) return bail(fail(PT_ERR_INVALID_ARG, "object primitive range out of bounds"));
            if (O.n_prims == 1) continue;
            std::vector<pth::PrimBound> pb(O.n_prims);
            for (uint32_t i = 0; i < O.n_prims; ++i) prim_bound(O.first_prim + i, pb[i]);
            if ((st = build_accel(pb, obj[o].nodes, obj[o].ordered))) return bail(st);
            for (auto &e : obj[o].ordered) e += O.first_prim;
        }
        for (uint32_t i = 0; i < d->n_instances; ++i) if (d->instances[i].object >= d->n_objects) return bail(fail(PT_ERR_INVALID_ARG, "instance object index out of range"));
        for (uint32_t i = 0; i < n_top; ++i) {
            uint32_t r = d->top_refs[i];
            if ((r & PT_TOP_INSTANCE) ? ((r & ~PT_TOP_INSTANCE) >= d->n_instances) : (r >= d->n_prims)) return bail(fail(PT_ERR_INVALID_ARG, "top_refs entry out of range"));
        }
    }
    auto top_ref = [&](uint32_t pos) { return instanced ? d->top_refs[pos] : pos; };
    if (d->nodes && d->n_nodes && d->ordered_prims) {
        sc->nodes.assign(d->nodes, d->nodes + d->n_nodes);
        sc->ordered.assign(d->ordered_prims, d->ordered_prims + n_top);
        for (uint32_t i = 0; i < n_top; ++i) if (sc->ordered[i] >= n_top) return bail(fail(PT_ERR_INVALID_ARG, "ordered_prims entry out of range"));
        for (uint32_t i = 0; i < d->n_nodes; ++i) {
            const PtBVHNode &n = sc->nodes[i];
            // (an interior node's second child lies behind its first child's whole subtree -- the flattened tree is in pre-order, bvh.rs:662-703: a back edge would
            //  send the depth-first walks below, and the device's, round in circles)
            bool ok = n.n_prims ? ((uint64_t)n.offset + n.n_prims <= n_top) : (n.offset < d->n_nodes && n.offset > i + 1 && n.axis < 3);
            if (!ok) return bail(fail(PT_ERR_INVALID_ARG, "malformed BVH node"));
        }
        // The four-wide walk never tests the boxes of a record's collapsed children nor the root's: sound when every child's box lies inside its parent's, as in
        // every tree a union of primitive bounds builds. An adopted tree that is not nested is walked two-wide, box by box, like the reference does.
        for (uint32_t i = 0; i < d->n_nodes && !sc->exact_walk_only; ++i) {
            const PtBVHNode &n = sc->nodes[i];
            if (n.n_prims) continue;
            for (uint32_t c : {i + 1u, (uint32_t)n.offset}) for (int k = 0; k < 3; ++k)
                if (!(sc->nodes[c].bmin[k] >= n.bmin[k] && sc->nodes[c].bmax[k] <= n.bmax[k])) sc->exact_walk_only = true;
        }
    } else {
        std::vector<pth::PrimBound> pb(n_top);
        for (uint32_t i = 0; i < n_top; ++i) {
            const uint32_t r = top_ref(i);
            if (!(r & PT_TOP_INSTANCE)) { prim_bound(r, pb[i]); continue; }
            // TransformedPrimitive::world_bound = prim_to_world.motion_bounds(inner bound) (primitive.rs:53-55, transform.rs:1564-1567,592-605)
            const PtInstance &I = d->instances[r & ~PT_TOP_INSTANCE];
            const PtObject &O = d->objects[I.object];
            pth::PrimBound inner;
            if (O.n_prims == 1) prim_bound(O.first_prim, inner);
            else for (int k = 0; k < 3; ++k) { inner.lo[k] = obj[I.object].nodes[0].bmin[k]; inner.hi[k] = obj[I.object].nodes[0].bmax[k]; }
            M4 i2w; std::memcpy(i2w.m, I.instance_to_world, 64);
            const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 1, 1}, {1, 1, 0}, {1, 0, 1}, {1, 1, 1}};
            for (int c = 0; c < 8; ++c) {
                V3 p = xf_point(i2w, V3(corner[c][0] ? inner.hi[0] : inner.lo[0], corner[c][1] ? inner.hi[1] : inner.lo[1], corner[c][2] ? inner.hi[2] : inner.lo[2]));
                const float pc[3] = {p.x, p.y, p.z};
                for (int k = 0; k < 3; ++k) { pb[i].lo[k] = c ? std::fmin(pb[i].lo[k], pc[k]) : pc[k]; pb[i].hi[k] = c ? std::fmax(pb[i].hi[k], pc[k]) : pc[k]; }
            }
        }
        if ((st = build_accel(pb, sc->nodes, sc->ordered))) return bail(st);
    }
    // uploads
#define UP(field, src, count) if ((st = sc->upload(&ds.field, src, (size_t)(count)))) return bail(st)
    // Two-wide traversal records (dev_scene.h: WideNode) and the packet order of every accelerator, concatenated:
    // [top level][object 0][object 1]... ; references inside an accelerator are offset by its bases.
    std::vector<uint32_t> leaf_last, packet_refs;
    std::vector<WideNode> wide;
    std::vector<DevInstance> dinst(instanced ? d->n_instances : 0);
    std::vector<QuadNode> quad;
    auto append_accel = [&](const std::vector<PtBVHNode> &nn, const std::vector<uint32_t> &refs, uint32_t &root_ref, uint32_t &root_ref4) {
        const uint32_t wbase = (uint32_t)wide.size(), pbase = (uint32_t)packet_refs.size();
        std::vector<uint32_t> wide_id(nn.size(), 0);
        uint32_t n_int = 0;
        for (size_t i = 0; i < nn.size(); ++i) { if (nn[i].n_prims == 0) wide_id[i] = wbase + n_int++; else leaf_last.push_back(pbase + nn[i].offset + nn[i].n_prims - 1); }
        auto ref_of = [&](uint32_t i) { return nn[i].n_prims ? (kLeafBit | (pbase + nn[i].offset)) : wide_id[i]; };
        wide.resize(wbase + n_int);
        for (size_t i = 0; i < nn.size(); ++i) {
            if (nn[i].n_prims) continue;
            WideNode &w = wide[wide_id[i]];
            const PtBVHNode &l = nn[i + 1], &r = nn[nn[i].offset];
            w.lmin[0] = l.bmin[0]; w.lmin[1] = l.bmin[1]; w.lmin[2] = l.bmin[2]; w.lmax0 = l.bmax[0];
            w.lmax12[0] = l.bmax[1]; w.lmax12[1] = l.bmax[2]; w.rmin01[0] = r.bmin[0]; w.rmin01[1] = r.bmin[1];
            w.rmin2 = r.bmin[2]; w.rmax[0] = r.bmax[0]; w.rmax[1] = r.bmax[1]; w.rmax[2] = r.bmax[2];
            w.left_ref = ref_of((uint32_t)i + 1); w.right_ref = ref_of(nn[i].offset);
            w.meta = nn[i].axis; w.pad = 0;
        }
        // four-wide records of the same tree (dev_scene.h: QuadNode): two binary levels per record, emitted depth first
        {
            const float inf = std::numeric_limits<float>::infinity();
            std::vector<uint32_t> todo;   // binary interior nodes that root a record, in emission order (their record = quad[qbase + position])
            std::vector<uint32_t> quad_id(nn.size(), PT_NONE);
            auto qref_of = [&](uint32_t i) { return nn[i].n_prims ? (kLeafBit | (pbase + nn[i].offset)) : quad_id[i]; };
            const uint32_t qbase = (uint32_t)quad.size();
            if (!nn.empty() && nn[0].n_prims == 0) {
                // pre-order numbering: a record's interior grandchildren root the next records, left to right
                std::vector<uint32_t> stack{0};
                while (!stack.empty()) {
                    const uint32_t i = stack.back(); stack.pop_back();
                    quad_id[i] = g_test_pool_pad_records + qbase + (uint32_t)todo.size(); todo.push_back(i);   // (test hook: unused records in front of the pool, host_device.hip)
                    uint32_t kids[4]; int nk = 0;
                    for (uint32_t c : {(uint32_t)i + 1u, (uint32_t)nn[i].offset}) {
                        if (nn[c].n_prims) continue;
                        kids[nk++] = c + 1u; kids[nk++] = nn[c].offset;
                    }
                    for (int k = nk - 1; k >= 0; --k) if (nn[kids[k]].n_prims == 0) stack.push_back(kids[k]);
                }
            }
            quad.resize(qbase + todo.size());
            for (size_t t = 0; t < todo.size(); ++t) {
                const uint32_t i = todo[t];
                QuadNode &q = quad[qbase + t];
                for (int a = 0; a < 3; ++a) for (int k = 0; k < 4; ++k) { q.lo[a][k] = inf; q.hi[a][k] = -inf; }
                for (int k = 0; k < 4; ++k) q.ref[k] = PT_NONE;
                q.pad[0] = q.pad[1] = q.pad[2] = 0;
                const uint32_t c2[2] = {i + 1u, (uint32_t)nn[i].offset};
                uint32_t axes[3] = {nn[i].axis, 0u, 0u};
                auto put = [&](int slot, uint32_t n) {
                    for (int a = 0; a < 3; ++a) { q.lo[a][slot] = nn[n].bmin[a]; q.hi[a][slot] = nn[n].bmax[a]; }
                    q.ref[slot] = qref_of(n);
                };
                for (int side = 0; side < 2; ++side) {
                    const uint32_t c = c2[side];
                    if (nn[c].n_prims) put(2 * side, c);
                    else { axes[1 + side] = nn[c].axis; put(2 * side, c + 1u); put(2 * side + 1, nn[c].offset); }
                }
                // order word: for each of the eight sign octants o = nx | ny << 1 | nz << 2, three bits at 3 o: bit 0 = the ray is negative along N's
                // axis (the right pair comes first), bit 1 = along L's (slot 1 before slot 0), bit 2 = along R's (slot 3 before slot 2)
                q.meta = 0;
                for (uint32_t o = 0; o < 8; ++o) q.meta |= (((o >> axes[0]) & 1u) | (((o >> axes[1]) & 1u) << 1) | (((o >> axes[2]) & 1u) << 2)) << (3u * o);
            }
            root_ref4 = nn.empty() ? 0u : qref_of(0);
        }
        packet_refs.insert(packet_refs.end(), refs.begin(), refs.end());
        root_ref = ref_of(0);
    };
    {
        std::vector<uint32_t> top_order(n_top);
        for (uint32_t i = 0; i < n_top; ++i) top_order[i] = top_ref(sc->ordered[i]);
        append_accel(sc->nodes, top_order, ds.root_ref, ds.root_ref4);
        ds.n_nodes = (uint32_t)sc->nodes.size();
        for (int k = 0; k < 3; ++k) { ds.root_min[k] = sc->nodes[0].bmin[k]; ds.root_max[k] = sc->nodes[0].bmax[k]; }
        std::vector<uint32_t> obj_root(obj.size(), 0), obj_root4(obj.size(), 0);
        for (size_t o = 0; o < obj.size(); ++o) {
            if (d->objects[o].n_prims == 1) {  // single primitive: a one-packet "leaf" without a BVH
                obj_root[o] = obj_root4[o] = kLeafBit | (uint32_t)packet_refs.size();
                leaf_last.push_back((uint32_t)packet_refs.size());
                packet_refs.push_back(d->objects[o].first_prim);
            } else append_accel(obj[o].nodes, obj[o].ordered, obj_root[o], obj_root4[o]);
        }
        for (size_t i = 0; i < dinst.size(); ++i) {
            const PtInstance &I = d->instances[i]; DevInstance &D = dinst[i];
            std::memcpy(D.world_to_instance, I.world_to_instance, 64); std::memcpy(D.instance_to_world, I.instance_to_world, 64);
            D.single = d->objects[I.object].n_prims == 1; D.root_ref = obj_root[I.object]; D.root_ref4 = obj_root4[I.object];
            for (int k = 0; k < 3; ++k) { D.root_min[k] = D.single ? 0.0f : obj[I.object].nodes[0].bmin[k]; D.root_max[k] = D.single ? 0.0f : obj[I.object].nodes[0].bmax[k]; }
            bool ident = true;
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) ident = ident && I.instance_to_world[4 * r + c] == ((r == c) ? 1.0f : 0.0f);
            D.identity = ident;
        }
        // The production walk addresses records and packets as 16-byte quads of one pool (checked where the pool is allocated: < 2^31 of each, < 64 GB together). The
        // two-wide walk of pt_set_trace_exact packs a skipped-entry count above 25-bit references: a larger scene has no exact walk (launch_trace refuses it).
        if (quad.size() + g_test_pool_pad_records >= ((size_t)1 << 28) || packet_refs.size() >= ((size_t)1 << 31) - 2) return bail(fail(PT_ERR_UNSUPPORTED, "scene exceeds 2^28 four-wide BVH records / 2^31 packets"));
        if (wide.size() > (size_t)kRefMask || packet_refs.size() > (size_t)kRefMask) { sc->quad_walk_only = true; wide.clear(); wide.shrink_to_fit(); }
        // an adopted tree that is not nested can only be walked two-wide, a scene beyond 2^25 records only four-wide: together no walk is left (such a scene
        // used to be accepted here and then failed at every traversal launch)
        if (sc->exact_walk_only && sc->quad_walk_only) return bail(fail(PT_ERR_UNSUPPORTED, "adopted BVH whose child boxes do not nest inside their parents' (needs the two-wide walk) in a scene beyond 2^25 records / packets (has the four-wide walk only)"));
        if (sc->exact_walk_only) fprintf(stderr, "mi355pt: warning: the adopted BVH's child boxes do not nest inside their parents'; this scene is walked two-wide, box by box (slower than the four-wide production walk)\n");
        if (wide.empty()) wide.resize(1);
        UP(wide, wide.data(), wide.size());
        UP(instances, dinst.data(), dinst.size()); ds.n_instances = (uint32_t)dinst.size();
    }
    UP(P, d->P, 3 * (size_t)d->n_vertices);
    if (d->N) UP(N, d->N, 3 * (size_t)d->n_vertices);
    if (d->S) UP(S, d->S, 3 * (size_t)d->n_vertices);
    if (d->UV) UP(UV, d->UV, 2 * (size_t)d->n_vertices);
    UP(indices, d->indices, 3 * (size_t)d->n_triangles); ds.n_triangles = d->n_triangles;
    {
        std::vector<uint8_t> fl(d->n_triangles, 0);
        if (d->tri_flags) fl.assign(d->tri_flags, d->tri_flags + d->n_triangles);
        for (auto &f : fl) { if (!d->N) f &= ~PT_TRI_HAS_N; if (!d->S) f &= ~PT_TRI_HAS_S; if (!d->UV) f &= ~PT_TRI_HAS_UV; }
        UP(tri_flags, fl.data(), fl.size());
    }
    UP(prim_shape, d->prim_shape, d->n_prims); UP(prim_material, d->prim_material, d->n_prims); UP(prim_light, d->prim_light, d->n_prims);
    ds.n_prims = d->n_prims;
    UP(materials, d->materials, d->n_materials); ds.n_materials = d->n_materials;
    if (d->n_media && d->media) {   // participating media (volpath only)
        UP(media, d->media, d->n_media); ds.n_media = d->n_media;
        std::vector<DevGridAux> aux(d->n_media, DevGridAux{nullptr, 0.0f, 0.0f});
        for (uint32_t i = 0; i < d->n_media; ++i) {   // GridDensityMedium::new (grid.rs:40-72)
            const PtMedium &m = d->media[i];
            if (m.type > PT_MEDIUM_GRID) return bail(fail(PT_ERR_INVALID_ARG, "unknown medium type"));
            if (m.type != PT_MEDIUM_GRID) continue;
            const size_t nvox = (size_t)m.nx * m.ny * m.nz;
            if (!m.density || nvox == 0 || nvox > ((size_t)1 << 31)) return bail(fail(PT_ERR_INVALID_ARG, "grid medium without a density grid"));
            // grid.rs:46-52: `sigma_t = (sigma_a + sigma_s)[0]`; a spectrally varying coefficient is reported with error!() and rendering goes on with the first channel
            for (int k = 1; k < 3; ++k) if (m.sigma_a[k] + m.sigma_s[k] != m.sigma_a[0] + m.sigma_s[0]) { fprintf(stderr, "mi355pt: GridDensityMedium requires spectrally uniform attenuation coefficient (medium %u: using channel 0, as grid.rs:46-52 does)\n", i); break; }
            float maxd = 0.0f;
            for (size_t k = 0; k < nvox; ++k) maxd = std::fmax(maxd, m.density[k]);
            if (!(maxd > 0.0f)) return bail(fail(PT_ERR_INVALID_ARG, "grid medium with no positive density"));
            if ((st = sc->upload(&aux[i].density, m.density, nvox))) return bail(st);
            aux[i].sigma_t = m.sigma_a[0] + m.sigma_s[0]; aux[i].inv_max_density = 1.0f / maxd;
            ds.has_grid = 1u;
        }
        UP(grid_aux, aux.data(), aux.size());
        if (d->prim_medium_inside && d->prim_medium_outside) {
            for (uint32_t i = 0; i < d->n_prims; ++i)
                if ((d->prim_medium_inside[i] != PT_NONE && d->prim_medium_inside[i] >= d->n_media) || (d->prim_medium_outside[i] != PT_NONE && d->prim_medium_outside[i] >= d->n_media))
                    return bail(fail(PT_ERR_INVALID_ARG, "primitive medium index out of range"));
            UP(prim_med_in, d->prim_medium_inside, d->n_prims); UP(prim_med_out, d->prim_medium_outside, d->n_prims);
        }
        sc->class_used[kMediumClass] = true;
    }
    for (uint32_t i = 0; i < d->n_prims; ++i) if (d->prim_material[i] == PT_NONE) sc->has_null_material = true;
    ds.has_shells = sc->has_null_material ? 1u : 0u;
    UP(spheres, d->spheres, d->n_spheres); ds.n_spheres = d->n_spheres;
    UP(lights, d->lights, d->n_lights); ds.n_lights = d->n_lights; sc->n_lights = d->n_lights;
    if (d->n_lights) sc->host_lights.assign(d->lights, d->lights + d->n_lights);
    if (d->env_texels) { sc->env_w = d->env_width; sc->env_h = d->env_height; for (int k = 0; k < 3; ++k) sc->env_texel0[k] = d->env_power_lookup[k]; }
    {
        std::vector<uint8_t> mc(std::max<uint32_t>(1, d->n_materials), 0);
        // (the volumetric router folds the lobe-set classes back into the lobe-count ones -- kern_aux.h: k_medium_route)
        for (uint32_t i = 0; i < d->n_materials; ++i) { mc[i] = material_class(d->materials[i], g_shade_specialise, d->n_textures == 0); sc->class_used[mc[i]] = true; }
        UP(mat_class, mc.data(), mc.size());
        std::vector<DevBssTable> bt(d->n_bssrdf_tables);
        for (uint32_t i = 0; i < d->n_bssrdf_tables; ++i) {
            const PtBSSRDFTable &t = d->bssrdf_tables[i];
            bt[i].n_rho = (int)t.n_rho; bt[i].n_radius = (int)t.n_radius;
            if (!t.rho_samples || !t.radius_samples || !t.profile || !t.rhoeff || !t.profile_cdf) continue;   // unreferenced slot
            if ((st = sc->upload(&bt[i].rho_samples, t.rho_samples, t.n_rho))) return bail(st);
            if ((st = sc->upload(&bt[i].radius_samples, t.radius_samples, t.n_radius))) return bail(st);
            if ((st = sc->upload(&bt[i].profile, t.profile, (size_t)t.n_rho * t.n_radius))) return bail(st);
            if ((st = sc->upload(&bt[i].rhoeff, t.rhoeff, t.n_rho))) return bail(st);
            if ((st = sc->upload(&bt[i].profile_cdf, t.profile_cdf, (size_t)t.n_rho * t.n_radius))) return bail(st);
        }
        UP(bss_tables, bt.data(), bt.size()); ds.n_bss_tables = d->n_bssrdf_tables;
        if (d->n_textures) {   // textures: nodes as given + one postfix program per node (children before parent)
            std::vector<PtMaterial> mats(d->materials, d->materials + d->n_materials);
            UP(textures, d->textures, d->n_textures); ds.n_textures = d->n_textures;
            std::vector<uint32_t> off(d->n_textures + 1, 0), prog;
            for (uint32_t r = 0; r < d->n_textures; ++r) {
                off[r] = (uint32_t)prog.size();
                // iterative post-order; the value-stack depth is tracked to validate kTexStack
                struct Fr { int node; int next; };
                std::vector<Fr> st{{(int)r, 0}};
                int depth = 0, max_depth = 0; size_t guard = 0;
                while (!st.empty()) {
                    Fr &f = st.back();
                    const PtTexture &t = d->textures[f.node];
                    const int nchild = (t.type == PT_TEX_MIX) ? 3 : (t.type == PT_TEX_SCALE || t.type == PT_TEX_CHECKERBOARD2D || t.type == PT_TEX_CHECKERBOARD3D || t.type == PT_TEX_DOTS) ? 2 : 0;
                    if (f.next < nchild) { const int c = t.child[f.next++]; st.push_back({c, 0}); if (++guard > 4096 || st.size() > 64) return bail(fail(PT_ERR_INVALID_ARG, "texture graph too deep or cyclic")); continue; }
                    prog.push_back((uint32_t)f.node);
                    depth += 1 - nchild; max_depth = std::max(max_depth, depth + nchild);
                    st.pop_back();
                }
                if (max_depth > kTexStack) return bail(fail(PT_ERR_UNSUPPORTED, "texture expression needs a deeper value stack than kTexStack"));
            }
            off[d->n_textures] = (uint32_t)prog.size();
            UP(tex_prog_offset, off.data(), off.size()); UP(tex_prog, prog.data(), prog.size());
            std::vector<DevImage> imgs(d->n_images);
            for (uint32_t i = 0; i < d->n_images; ++i) {
                const PtImage &im = d->images[i];
                imgs[i].width = im.width; imgs[i].height = im.height; imgs[i].n_levels = im.n_levels; imgs[i].channels = im.channels;
                if (!im.texels || im.n_levels == 0 || im.n_levels > 16) continue;   // unreferenced slot
                size_t o = 0;
                for (uint32_t l = 0; l < im.n_levels; ++l) { imgs[i].level_offset[l] = (uint32_t)o; o += (size_t)std::max(1u, im.width >> l) * std::max(1u, im.height >> l) * im.channels; }
                if ((st = sc->upload(&imgs[i].texels, im.texels, o))) return bail(st);
            }
            UP(images, imgs.data(), imgs.size());
            if (d->ewa_weight_lut) UP(ewa_lut, d->ewa_weight_lut, 128);
            auto any_mask = [&](const int32_t *a) { if (!a) return false; for (uint32_t i = 0; i < d->n_triangles; ++i) if (a[i] >= 0) return true; return false; };
            if (any_mask(d->tri_alpha)) UP(tri_alpha, d->tri_alpha, d->n_triangles);
            if (any_mask(d->tri_shadow_alpha)) UP(tri_shadow_alpha, d->tri_shadow_alpha, d->n_triangles);
        }
        for (uint32_t i = 0; i < d->n_materials; ++i) if (d->materials[i].type == PT_MAT_SUBSURFACE || disney_has_bssrdf(d->materials[i])) sc->has_bssrdf = true;
        std::vector<uint32_t> inf;
        for (uint32_t i = 0; i < d->n_lights; ++i) if (d->lights[i].type == PT_LIGHT_INFINITE) inf.push_back(i);
        UP(infinite_lights, inf.data(), inf.size()); ds.n_infinite = (uint32_t)inf.size();
    }
    if (d->env_texels) {  // Distribution2D::new (sampling.rs:100-117) over the importance image
        ds.env_w = d->env_width; ds.env_h = d->env_height;
        UP(env_texels, d->env_texels, 3 * (size_t)ds.env_w * ds.env_h);
        size_t nu = 2 * (size_t)ds.env_w, nv = 2 * (size_t)ds.env_h;
        std::vector<float> func(d->env_importance, d->env_importance + nu * nv), cdf(nv * (nu + 1)), fint(nv), mcdf; float mint;
        for (size_t v = 0; v < nv; ++v) {
            std::vector<float> row(func.begin() + v * nu, func.begin() + (v + 1) * nu), c; float fi;
            dist1d(row, c, fi);
            std::copy(c.begin(), c.end(), cdf.begin() + v * (nu + 1)); fint[v] = fi;
        }
        dist1d(fint, mcdf, mint);
        UP(env_func, func.data(), func.size()); UP(env_cdf, cdf.data(), cdf.size()); UP(env_func_int, fint.data(), fint.size());
        UP(env_marg_func, fint.data(), fint.size()); UP(env_marg_cdf, mcdf.data(), mcdf.size()); ds.env_marg_int = mint;
    }
#undef UP
    // world bound = root node bounds (bvh.rs:697-703); Light::preprocess -> bounding sphere (bounds.rs:516-524)
    for (int k = 0; k < 3; ++k) { ds.wb_min[k] = sc->nodes[0].bmin[k]; ds.wb_max[k] = sc->nodes[0].bmax[k]; }
    {
        float c[3]; bool inside = true;
        for (int k = 0; k < 3; ++k) { c[k] = (ds.wb_min[k] + ds.wb_max[k]) * (1.0f / 2.0f); inside = inside && c[k] >= ds.wb_min[k] && c[k] <= ds.wb_max[k]; }
        float dx = ds.wb_max[0] - c[0], dy = ds.wb_max[1] - c[1], dz = ds.wb_max[2] - c[2];
        ds.world_radius = inside ? std::sqrt(dx * dx + dy * dy + dz * dz) : 0.0f;
        for (int k = 0; k < 3; ++k) ds.world_center[k] = c[k];
    }
    // leaf triangle packets + light areas (device)
    {
        const uint32_t *d_ordered = nullptr;
        const uint32_t n_packets = (uint32_t)packet_refs.size();
        if ((st = sc->upload(&d_ordered, packet_refs.data(), packet_refs.size()))) return bail(st);
        TriPacket *leaf = nullptr; float *area = nullptr; float4 *lrec = nullptr;
        // the four-wide records and the packets share ONE allocation, so that the production traversal addresses both with 32-bit quad indices from
        // one base (records of 8 quads, packets of 3; < 64 GB): [records][packets + 2] (+2: a packet's fourth quad is loaded with it)
        if (quad.empty()) quad.resize(1);
        const size_t pad_bytes = (size_t)g_test_pool_pad_records * sizeof(QuadNode);
        const size_t quad_bytes = pad_bytes + quad.size() * sizeof(QuadNode), pool_bytes = quad_bytes + ((size_t)n_packets + 2) * sizeof(TriPacket);
        if (pool_bytes / 16 >= (size_t)0xfffffff0u) return bail(fail(PT_ERR_UNSUPPORTED, "scene exceeds 64 GB of traversal records + packets"));
        uint8_t *pool = nullptr;
        if ((st = sc->dalloc(&pool, pool_bytes))) return bail(st);
        // (record 0 is read as a dummy by the leaf lanes of k_trace<.., 2>, which aim their five node-only loads at it: with the test hook's pad in front of the pool that is pad memory)
        if (pad_bytes && hipMemset(pool, 0, pad_bytes) != hipSuccess) return bail(fail(PT_ERR_HIP, "memset of the pool pad"));
        if (hipMemcpy(pool + pad_bytes, quad.data(), quad_bytes - pad_bytes, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(PT_ERR_HIP, "upload of the four-wide records"));
        leaf = reinterpret_cast<TriPacket *>(pool + quad_bytes);
        ds.quad = reinterpret_cast<const QuadNode *>(pool); ds.leaf_off = (uint32_t)(quad_bytes / 16); ds.pool_quads = (uint32_t)(pool_bytes / 16);
        sc->pool_big = pool_bytes >= (size_t)0xE0000000u;   // (beyond what the buffer loads' 32-bit byte offsets reach: k_trace<.., 2>)
        if (hipMemset(leaf, 0, ((size_t)n_packets + 2) * sizeof(TriPacket)) != hipSuccess) return bail(fail(PT_ERR_HIP, "memset"));
        if ((st = sc->dalloc(&area, std::max<uint32_t>(1, d->n_lights)))) return bail(st);
        if ((st = sc->dalloc(&lrec, 6 * (size_t)std::max<uint32_t>(1, d->n_lights)))) return bail(st);
        hipLaunchKernelGGL(k_build_packets, dim3((n_packets + 255) / 256), dim3(256), 0, 0, ds, d_ordered, n_packets, leaf);
        ds.leaf = leaf;
        {
            const uint32_t *d_last = nullptr;
            if ((st = sc->upload(&d_last, leaf_last.data(), leaf_last.size()))) return bail(st);
            hipLaunchKernelGGL(k_mark_leaf_ends, dim3(((uint32_t)leaf_last.size() + 255) / 256), dim3(256), 0, 0, leaf, d_last, (uint32_t)leaf_last.size());
        }
        if (d->n_lights) hipLaunchKernelGGL(k_light_area, dim3((d->n_lights + 255) / 256), dim3(256), 0, 0, ds, area, lrec);
        ds.light_area = area; ds.light_rec = lrec;
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return bail(fail(PT_ERR_HIP, "scene preparation kernels failed"));
    }
    *out = sc;
    return PT_OK;
}

void pt_scene_destroy(pt_scene *sc) {
    if (!sc) return;
    if (sc->device != g_device) bind_device(sc->device);
    for (void *p : sc->allocs) hipFree(p);
    if (sc->slab) hipFree(sc->slab);
    if (sc->qbuf) hipFree(sc->qbuf);
    if (sc->bss_slab) hipFree(sc->bss_slab);
    if (sc->ext_slab) hipFree(sc->ext_slab);
    if (sc->film_rgbw) hipFree(sc->film_rgbw);
    sc->drop_timings();
    for (auto e : sc->event_pool) hipEventDestroy(e);
    if (sc->stream) hipStreamDestroy(sc->stream);
    delete sc;
}

int pt_scene_bvh_info(const pt_scene *sc, uint32_t *n_nodes, uint32_t *n_prims) {
    if (!sc || !n_nodes || !n_prims) return fail(PT_ERR_INVALID_ARG, "null argument");
    *n_nodes = (uint32_t)sc->nodes.size(); *n_prims = (uint32_t)sc->ordered.size();
    return PT_OK;
}
int pt_scene_bvh_read(const pt_scene *sc, PtBVHNode *nodes, uint32_t *ordered) {
    if (!sc || !nodes || !ordered) return fail(PT_ERR_INVALID_ARG, "null argument");
    std::memcpy(nodes, sc->nodes.data(), sc->nodes.size() * sizeof(PtBVHNode));
    std::memcpy(ordered, sc->ordered.data(), sc->ordered.size() * 4);
    return PT_OK;
}

}  // extern "C"
