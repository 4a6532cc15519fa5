// frontend.cpp -- .pbrt scene-file front end in front of the C ABI: the compiled-host counterpart of
// pbrt-rust_amd/host.py. It restates the directive semantics of the reference's host side and flattens the result into
// the POD structs of include/mi355pt.h; nothing here runs per sample.
//   pbrtparser/pbrtparser.rs:34-87 (directive dispatch), core/api.rs:941-1327 (transforms, attribute stack, named
//   coordinate systems), :1329-1460 (Camera/Film/Sampler/PixelFilter/Integrator options), :1462-1619 (LightSource,
//   AreaLightSource, Shape: one GeometricPrimitive per triangle / sphere, one DiffuseAreaLight per emissive shape),
//   :1630-1713 (ObjectBegin/End/Instance), :1715-1748 (WorldEnd), core/paramset.rs:500-600 (texture-or-constant lookups),
//   cameras/perspective.rs:40-86,298-356, core/film.rs:55-112,364-398, filters/*.rs, lights/*.rs create_* functions,
//   materials/*.rs create_* functions, shapes/{triangle.rs:700-760, sphere.rs:424-431, plymesh.rs}.
// Out of scope here (an error names the directive): other cameras / samplers / integrators than perspective / sobol, halton / path,
// participating media, spectral (non-RGB) parameters, image formats other than PFM, per-shape material parameter
// overrides, animated transforms (ActiveTransform / TransformTimes are accepted and ignored for static scenes).
#include "../../include/mi355pt.h"
#include "fe_bssrdf.h"
#include "fe_image.h"
#include "fe_imageio.h"
#include "fe_math.h"
#include "fe_params.h"
#include "fe_ply.h"

#include <cstdio>
#include <functional>
#include <fstream>
#include <memory>
#include <set>
#include <sstream>

namespace fe {

static thread_local std::string g_error;

struct GState {
    Transform ctm; bool reverse = false;
    int material = -1;
    bool has_area = false; float area_L[3] = {1, 1, 1}; bool area_two_sided = false;
    std::map<std::string, int> float_tex, spec_tex, named_materials;
    std::string medium_inside, medium_outside;     // MediumInterface (api.rs:1243-1253) keeps the NAMES; they are looked up when a shape / the camera is made (api.rs:382-403)
};

struct Scene {
    // flattened arrays (the same layout pbrt-rust_amd/host.py produces)
    std::vector<float> P, N, S, UV; std::vector<uint8_t> vert_has_n, vert_has_s, vert_has_uv;
    std::vector<uint32_t> indices; std::vector<uint8_t> tri_flags; std::vector<int32_t> tri_alpha, tri_shadow_alpha;
    std::vector<PtSphere> spheres;
    std::vector<uint32_t> prim_shape, prim_material, prim_light, prim_med_in, prim_med_out;
    std::vector<PtMedium> media; std::map<std::string, int> named_media; int camera_medium = -1; bool volpath = false;
    std::vector<std::vector<float>> media_density; std::vector<int> media_density_of;   // grid media: density arrays and the medium each belongs to
    std::vector<PtMaterial> materials; std::vector<PtLight> lights;
    std::vector<PtTexture> textures; std::vector<Pyramid> pyramids; std::vector<PtImage> images; std::vector<float> ewa_lut;
    std::vector<std::unique_ptr<BssTable>> bss_tables; std::vector<std::pair<float, float>> bss_keys; std::vector<PtBSSRDFTable> bss_desc;
    std::vector<float> env_texels, env_importance; uint32_t env_w = 0, env_h = 0; float env_power_lookup[3] = {0, 0, 0};
    std::vector<PtObject> objects; std::vector<std::string> object_names; std::vector<PtInstance> instances; std::vector<uint32_t> top_refs;
    std::map<std::string, std::pair<uint32_t, uint32_t>> object_ranges; std::string current_object; bool in_object = false;
    // options
    int xres = 1280, yres = 720; float crop[4] = {0, 1, 0, 1}; float film_scale = 1.0f, max_lum = INFINITY; std::string filename = "pbrt.exr";   // film.rs:371
    std::string filter = "box"; ParamSet filter_params;
    ParamSet camera_params; Transform camera_to_world; std::string camera_name = "perspective";
    int spp = 16; std::string sampler = "halton";   // RenderOptions::default (api.rs:215-241)
    bool sample_at_center = false;
    int maxdepth = 5; float rr_threshold = 1.0f; std::string strategy = "spatial"; bool has_pixel_bounds = false; int pixel_bounds[4] = {0, 0, 0, 0};
    uint32_t max_node_prims = 4, split_method = PT_SPLIT_SAH;
    PtSceneDesc desc{}; PtRenderParams rp{};
    std::string base_dir;
};

static void copy3(float dst[3], const float src[3]) { dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; }

class Api {
public:
    explicit Api(Scene &s) : sc(s) { new_material("matte", ParamSet()); gs.material = 0; }   // api.rs:345-361: default matte Kd .5

    void run(Lexer &lx) {
        for (;;) {
            Token t = lx.next();
            if (t.kind == Token::End) break;
            if (t.kind != Token::Word) fail(t, "expected a directive");
            directive(t, lx);
        }
    }

private:
    Scene &sc; GState gs; std::vector<GState> stack; std::vector<Transform> tstack;
    std::map<std::string, Transform> named_cs;
    bool in_world = false;

    [[noreturn]] void fail(const Token &t, const std::string &msg) { throw std::runtime_error("line " + std::to_string(t.line) + ": " + msg); }
    // what the reference reports with warn!() / error!() and then carries on with (api.rs:388-399,753-756)
    static void warn(const Token &t, const std::string &msg) { std::fprintf(stderr, "mi355front: warning: line %d: %s\n", t.line, msg.c_str()); }
    std::set<std::string> undefined_media_reported;
    int medium_index(const std::string &nm) {   // GraphicsState::create_medium_interface (api.rs:382-403): an undefined name is an error message and no medium
        if (nm.empty()) return -1;
        auto it = sc.named_media.find(nm);
        if (it != sc.named_media.end()) return it->second;
        if (undefined_media_reported.insert(nm).second) std::fprintf(stderr, "mi355front: error: Named medium \"%s\" undefined\n", nm.c_str());
        return -1;
    }
    static std::string str_arg(Lexer &lx, const Token &d) { Token t = lx.next(); if (t.kind != Token::Str) throw std::runtime_error("line " + std::to_string(d.line) + ": " + d.text + " needs a quoted name"); return t.text; }
    static void nums(Lexer &lx, const Token &d, float *out, int n) {
        Token p = lx.peek(); bool br = p.kind == Token::LBracket; if (br) lx.next();
        for (int i = 0; i < n; ++i) { Token t = lx.next(); if (t.kind != Token::Num) throw std::runtime_error("line " + std::to_string(d.line) + ": " + d.text + " needs " + std::to_string(n) + " numbers"); out[i] = t.num; }
        if (br) { Token t = lx.next(); if (t.kind != Token::RBracket) throw std::runtime_error("line " + std::to_string(d.line) + ": missing ]"); }
    }

    void directive(const Token &d, Lexer &lx) {
        const std::string &w = d.text;
        float v[16];
        if (w == "Identity") gs.ctm = Transform();
        else if (w == "Translate") { nums(lx, d, v, 3); gs.ctm = gs.ctm * Transform::translate(v3(v[0], v[1], v[2])); }
        else if (w == "Scale") { nums(lx, d, v, 3); gs.ctm = gs.ctm * Transform::scale(v[0], v[1], v[2]); }
        else if (w == "Rotate") { nums(lx, d, v, 4); gs.ctm = gs.ctm * Transform::rotate(v[0], v3(v[1], v[2], v[3])); }
        else if (w == "LookAt") { nums(lx, d, v, 9); gs.ctm = gs.ctm * Transform::look_at(v3(v[0], v[1], v[2]), v3(v[3], v[4], v[5]), v3(v[6], v[7], v[8])); }
        else if (w == "Transform" || w == "ConcatTransform") {   // api.rs:981-1010: the 16 values are column major
            nums(lx, d, v, 16);
            Mat4 m; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.m[i][j] = v[4 * j + i];
            Transform t(m);
            gs.ctm = (w == "Transform") ? t : gs.ctm * t;
        }
        else if (w == "CoordinateSystem") named_cs[str_arg(lx, d)] = gs.ctm;
        else if (w == "CoordSysTransform") { auto it = named_cs.find(str_arg(lx, d)); if (it != named_cs.end()) gs.ctm = it->second; }
        else if (w == "ActiveTransform") { Token t = lx.next(); if (t.kind != Token::Word) fail(d, "ActiveTransform needs All | StartTime | EndTime"); }
        else if (w == "TransformTimes") nums(lx, d, v, 2);
        else if (w == "TransformBegin") tstack.push_back(gs.ctm);
        else if (w == "TransformEnd") { if (tstack.empty()) fail(d, "unmatched TransformEnd"); gs.ctm = tstack.back(); tstack.pop_back(); }
        else if (w == "AttributeBegin") { stack.push_back(gs); tstack.push_back(gs.ctm); }
        else if (w == "AttributeEnd") { if (stack.empty()) fail(d, "unmatched AttributeEnd"); gs = stack.back(); stack.pop_back(); if (!tstack.empty()) tstack.pop_back(); }
        else if (w == "ReverseOrientation") gs.reverse = !gs.reverse;
        else if (w == "Camera") { sc.camera_name = str_arg(lx, d); sc.camera_params = read_params(lx); sc.camera_to_world = gs.ctm.inverse(); named_cs["camera"] = sc.camera_to_world; }
        else if (w == "Film") { std::string n = str_arg(lx, d); ParamSet p = read_params(lx); film(d, n, p); }
        else if (w == "Sampler") { sc.sampler = str_arg(lx, d); ParamSet p = read_params(lx); sc.spp = p.one_int("pixelsamples", 16); sc.sample_at_center = p.one_bool("samplepixelcenter", false); }
        else if (w == "PixelFilter") { sc.filter = str_arg(lx, d); sc.filter_params = read_params(lx); }
        else if (w == "Integrator") { std::string n = str_arg(lx, d); ParamSet p = read_params(lx); integrator(d, n, p); }
        else if (w == "Accelerator") { std::string n = str_arg(lx, d); ParamSet p = read_params(lx); if (n != "bvh") fail(d, "only the bvh accelerator is supported");
                                       { const std::string sm = p.one_string("splitmethod", "sah"); if (sm == "hlbvh") sc.split_method = PT_SPLIT_HLBVH; else if (sm != "sah") fail(d, "splitmethod \"" + sm + "\" is not supported (sah, hlbvh)"); } sc.max_node_prims = (uint32_t)p.one_int("maxnodeprims", 4); }
        else if (w == "WorldBegin") { in_world = true; gs.ctm = Transform(); named_cs["world"] = gs.ctm; }
        else if (w == "WorldEnd") { in_world = false; sc.camera_medium = medium_index(gs.medium_outside); }   // api.rs:1738-1741: the camera's medium is the OUTSIDE medium of the graphics state at WorldEnd
        else if (w == "MakeNamedMedium") {   // api.rs:706-722,1219-1241
            std::string n = str_arg(lx, d); ParamSet p = read_params(lx);
            const std::string ty = p.one_string("type", "");
            if (ty.empty()) { warn(d, "No parameter string \"type\" found in MakeNamedMedium"); return; }                 // api.rs:1225-1226
            if (ty != "homogeneous" && ty != "heterogeneous") { warn(d, "Medium \"" + ty + "\" unknown."); return; }       // api.rs:753-756: warns, the name stays undefined
            float siga[3] = {0.0011f, 0.0024f, 0.014f}, sigs[3] = {2.55f, 3.21f, 3.77f};
            const std::string preset = p.one_string("preset", "");
            if (!preset.empty()) { auto it = named_media().find(preset); if (it != named_media().end()) { copy3(siga, it->second.sigma_a); copy3(sigs, it->second.sigma_prime_s); } }
            const float scale = p.one_float("scale", 1.0f);
            p.rgb("sigma_a", siga); p.rgb("sigma_s", sigs);
            PtMedium m{}; for (int k = 0; k < 3; ++k) { m.sigma_a[k] = siga[k] * scale; m.sigma_s[k] = sigs[k] * scale; } m.g = p.one_float("g", 0.0f);
            m.type = PT_MEDIUM_HOMOGENEOUS;
            if (ty == "heterogeneous") {   // api.rs:723-752: GridDensityMedium over [p0, p1] of the current transform's space
                const std::vector<float> *dens = p.floats("float", "density");
                if (!dens || dens->empty()) fail(d, "No \"density\" values provided for heterogeneous medium?");
                const int nx = p.one_int("nx", 1), ny = p.one_int("ny", 1), nz = p.one_int("nz", 1);
                if (nx <= 0 || ny <= 0 || nz <= 0 || dens->size() != (size_t)nx * ny * nz) fail(d, "GridDensityMedium has " + std::to_string(dens->size()) + " density values; expected nx*ny*nz = " + std::to_string((long long)nx * ny * nz));
                float p0[3] = {0.0f, 0.0f, 0.0f}, p1[3] = {1.0f, 1.0f, 1.0f};
                p.vec3("point3", "p0", p0); p.vec3("point3", "p1", p1);
                const Transform med2w = gs.ctm * Transform::translate(Vec3{p0[0], p0[1], p0[2]}) * Transform::scale(p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]);
                m.type = PT_MEDIUM_GRID; m.nx = (uint32_t)nx; m.ny = (uint32_t)ny; m.nz = (uint32_t)nz;
                med2w.flat(m.world_to_medium, true);
                sc.media_density.emplace_back(*dens);
                m.density = nullptr;   // patched when the scene description is assembled (the vectors may still move)
                sc.media_density_of.push_back((int)sc.media.size());
            }
            sc.media.push_back(m); sc.named_media[n] = (int)sc.media.size() - 1;
        }
        else if (w == "MediumInterface") {
            const std::string in = str_arg(lx, d); std::string out = in;
            { Token t = lx.peek(); if (t.kind == Token::Str) out = str_arg(lx, d); }   // one name = both sides (pbrtparser)
            gs.medium_inside = in; gs.medium_outside = out;
        }
        else if (w == "Material") { std::string n = str_arg(lx, d); ParamSet p = read_params(lx); gs.material = (n.empty() || n == "none") ? -1 : new_material(n, p); }
        else if (w == "MakeNamedMaterial") { std::string n = str_arg(lx, d); ParamSet p = read_params(lx); gs.named_materials[n] = new_material(p.one_string("type", "matte"), p); }
        else if (w == "NamedMaterial") { std::string n = str_arg(lx, d); auto it = gs.named_materials.find(n); if (it == gs.named_materials.end()) fail(d, "named material \"" + n + "\" not defined"); gs.material = it->second; }
        else if (w == "Texture") { std::string name = str_arg(lx, d), ty = str_arg(lx, d), cls = str_arg(lx, d); ParamSet p = read_params(lx); texture(d, name, ty, cls, p); }
        else if (w == "LightSource") { std::string n = str_arg(lx, d); ParamSet p = read_params(lx); light(d, n, p); }
        else if (w == "AreaLightSource") {
            std::string n = str_arg(lx, d); ParamSet p = read_params(lx);
            if (n != "diffuse" && n != "area") fail(d, "area light \"" + n + "\" unknown");
            float L[3] = {1, 1, 1}; p.rgb("L", L); float s = p.one_float("scale", 1.0f);   // diffuse.rs create_diffuse_arealight
            for (int i = 0; i < 3; ++i) gs.area_L[i] = L[i] * s;
            gs.area_two_sided = p.one_bool("twosided", false); gs.has_area = true;
        }
        else if (w == "Shape") { std::string n = str_arg(lx, d); ParamSet p = read_params(lx); shape(d, n, p); }
        else if (w == "ObjectBegin") { stack.push_back(gs); tstack.push_back(gs.ctm); sc.current_object = str_arg(lx, d); sc.in_object = true; sc.object_ranges[sc.current_object] = {(uint32_t)sc.prim_shape.size(), 0u}; }
        else if (w == "ObjectEnd") { sc.in_object = false; if (stack.empty()) fail(d, "unmatched ObjectEnd"); gs = stack.back(); stack.pop_back(); if (!tstack.empty()) tstack.pop_back(); }
        else if (w == "ObjectInstance") object_instance(d, str_arg(lx, d));
        else if (w == "Include") {
            std::string fn = str_arg(lx, d);
            std::ifstream f(sc.base_dir + fn); if (!f) fail(d, "Include: cannot open \"" + fn + "\"");
            std::stringstream ss; ss << f.rdbuf(); std::string text = ss.str();
            Lexer sub(text); run(sub);
        }
        else if (w == "MakeNamedMedium" || w == "MediumInterface") fail(d, w + ": participating media are out of scope (SURVEY 8f-4)");
        else fail(d, "unknown directive " + w);
        (void)in_world;
    }

    // ---- options ----------------------------------------------------------------------------------------------
    void film(const Token &d, const std::string &name, const ParamSet &p) {   // film.rs:364-398
        if (name != "image") fail(d, "film \"" + name + "\" unknown");
        sc.xres = p.one_int("xresolution", 1280); sc.yres = p.one_int("yresolution", 720);
        if (const std::vector<float> *cw = p.floats("float", "cropwindow")) {
            if (cw->size() == 4) {
                auto cl = [](float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); };
                sc.crop[0] = cl(std::fmin((*cw)[0], (*cw)[1])); sc.crop[1] = cl(std::fmax((*cw)[0], (*cw)[1]));
                sc.crop[2] = cl(std::fmin((*cw)[2], (*cw)[3])); sc.crop[3] = cl(std::fmax((*cw)[2], (*cw)[3]));
            }
        }
        sc.film_scale = p.one_float("scale", 1.0f); sc.max_lum = p.one_float("maxsampleluminance", INFINITY);
        sc.filename = p.one_string("filename", "pbrt.exr");
    }
    void integrator(const Token &d, const std::string &name, const ParamSet &p) {   // path.rs:225-253, volpath.rs:188-227 (same parameters)
        if (name != "path" && name != "volpath") fail(d, "integrator \"" + name + "\": only \"path\" and \"volpath\" run on this back end");
        sc.volpath = name == "volpath";
        sc.maxdepth = p.one_int("maxdepth", 5); sc.rr_threshold = p.one_float("rrthreshold", 1.0f);
        sc.strategy = p.one_string("lightsamplestrategy", "spatial");
        if (const std::vector<float> *pb = p.floats("int", "pixelbounds")) if (pb->size() == 4) { sc.has_pixel_bounds = true; for (int i = 0; i < 4; ++i) sc.pixel_bounds[i] = (int)(*pb)[i]; }
    }

    // ---- textures (api.rs pbrt_texture; textures/*.rs create_*) -----------------------------------------------
    int const_tex(const float v[3]) { PtTexture t{}; t.type = PT_TEX_CONSTANT; t.child[0] = t.child[1] = t.child[2] = -1; copy3(t.value, v); sc.textures.push_back(t); return (int)sc.textures.size() - 1; }
    // TextureParams::get_{spectrum,float}texture: a named texture, else a constant (paramset.rs:500-600)
    int child_tex(const Token &d, const ParamSet &p, const std::string &name, bool is_float, float dflt) {
        std::string tn = p.texture(name);
        if (!tn.empty()) { auto &tab = is_float ? gs.float_tex : gs.spec_tex; auto it = tab.find(tn); if (it == tab.end()) fail(d, "texture \"" + tn + "\" not declared"); return it->second; }
        float v[3] = {dflt, dflt, dflt};
        if (is_float) { float f = p.one_float(name, dflt); v[0] = v[1] = v[2] = f; } else p.rgb(name, v);
        return const_tex(v);
    }
    void mapping(PtTexture &t, const ParamSet &p) {   // get_mapping2d, texture.rs:439-466
        std::string ty = p.one_string("mapping", "uv");
        t.mapping = ty == "planar" ? PT_MAP_PLANAR : ty == "spherical" ? PT_MAP_SPHERICAL : ty == "cylindrical" ? PT_MAP_CYLINDRICAL : PT_MAP_UV;
        t.su = p.one_float("uscale", 1.0f); t.sv = p.one_float("vscale", 1.0f); t.du = p.one_float("udelta", 0.0f); t.dv = p.one_float("vdelta", 0.0f);
        float v1[3] = {1, 0, 0}, v2[3] = {0, 1, 0}; p.vec3("vector3", "v1", v1); p.vec3("vector3", "v2", v2); copy3(t.vs, v1); copy3(t.vt, v2);
        gs.ctm.flat(t.world_to_texture, true);
    }
    void texture(const Token &d, const std::string &name, const std::string &ty, const std::string &cls, const ParamSet &p) {
        const bool is_float = ty == "float";
        if (!is_float && ty != "color" && ty != "spectrum") fail(d, "texture type \"" + ty + "\" unknown");
        PtTexture t{}; t.child[0] = t.child[1] = t.child[2] = -1;
        if (cls == "constant") { t.type = PT_TEX_CONSTANT; float v[3] = {1, 1, 1}; if (is_float) { float f = p.one_float("value", 1.0f); v[0] = v[1] = v[2] = f; } else p.rgb("value", v); copy3(t.value, v); }
        else if (cls == "scale") { t.type = PT_TEX_SCALE; t.child[0] = child_tex(d, p, "tex1", is_float, 1.0f); t.child[1] = child_tex(d, p, "tex2", is_float, 1.0f); }
        else if (cls == "mix") { t.type = PT_TEX_MIX; t.child[0] = child_tex(d, p, "tex1", is_float, 0.0f); t.child[1] = child_tex(d, p, "tex2", is_float, 1.0f); t.child[2] = child_tex(d, p, "amount", true, 0.5f); }
        else if (cls == "checkerboard") {
            const int dim = p.one_int("dimension", 2);
            if (dim != 2 && dim != 3) fail(d, std::to_string(dim) + " dimensional checkerboard texture not supported");
            t.child[0] = child_tex(d, p, "tex1", is_float, 1.0f); t.child[1] = child_tex(d, p, "tex2", is_float, 0.0f);
            if (dim == 2) { t.type = PT_TEX_CHECKERBOARD2D; mapping(t, p); t.aa_closedform = p.one_string("aamode", "none") == "none" ? 0u : 1u; }
            else { t.type = PT_TEX_CHECKERBOARD3D; gs.ctm.flat(t.world_to_texture, true); }
        }
        else if (cls == "imagemap") {
            t.type = PT_TEX_IMAGEMAP; mapping(t, p);
            std::string fn = p.one_string("filename", ""), wrap = p.one_string("wrap", "repeat");
            if (wrap == "clamp") fail(d, "imagemap wrap \"clamp\" is not supported");
            const bool is_png = fn.size() > 4 && (fn.substr(fn.size() - 4) == ".png" || fn.substr(fn.size() - 4) == ".tga");
            Image im = read_image(sc.base_dir + fn);
            sc.pyramids.push_back(prepare_image(im, p.one_float("scale", 1.0f), p.one_bool("gamma", is_png), is_float ? 1 : 3, wrap == "black" ? 1 : 0));
            t.image = (uint32_t)sc.pyramids.size() - 1; t.trilinear = p.one_bool("trilinear", false) ? 1u : 0u;
            t.max_anisotropy = p.one_float("maxanisotropy", 8.0f); t.wrap = wrap == "black" ? PT_WRAP_BLACK : PT_WRAP_REPEAT;
        }
        else if (cls == "uv") { if (is_float) fail(d, "uv textures are spectrum-only"); t.type = PT_TEX_UV; mapping(t, p); }
        else if (cls == "bilerp") {
            t.type = PT_TEX_BILERP; mapping(t, p);
            struct { const char *n; float d; float *dst; } vs[4] = {{"v00", 0, t.v00}, {"v01", 1, t.v01}, {"v10", 0, t.v10}, {"v11", 1, t.v11}};
            for (auto &e : vs) { float v[3] = {e.d, e.d, e.d}; if (is_float) { float f = p.one_float(e.n, e.d); v[0] = v[1] = v[2] = f; } else p.rgb(e.n, v); copy3(e.dst, v); }
        }
        else if (cls == "fbm" || cls == "wrinkled" || cls == "windy" || cls == "marble") {
            if (cls == "marble" && is_float) fail(d, "marble textures are spectrum-only");
            t.type = cls == "fbm" ? PT_TEX_FBM : cls == "wrinkled" ? PT_TEX_WRINKLED : cls == "windy" ? PT_TEX_WINDY : PT_TEX_MARBLE;
            gs.ctm.flat(t.world_to_texture, true);
            t.octaves = (uint32_t)p.one_int("octaves", 8); t.omega = p.one_float("roughness", 0.5f);
            t.marble_scale = p.one_float("scale", 1.0f); t.variation = p.one_float("variation", 0.2f);
        }
        else if (cls == "dots") { t.type = PT_TEX_DOTS; mapping(t, p); t.child[0] = child_tex(d, p, "outside", is_float, 0.0f); t.child[1] = child_tex(d, p, "inside", is_float, 1.0f); }
        else fail(d, "texture class \"" + cls + "\" unknown");
        sc.textures.push_back(t);
        (is_float ? gs.float_tex : gs.spec_tex)[name] = (int)sc.textures.size() - 1;
    }

    // ---- materials (materials/*.rs create_*_material) -----------------------------------------------------------
    int new_material(const std::string &kind, const ParamSet &p) {
        PtMaterial m{};
        for (int i = 0; i < 16; ++i) m.tex[i] = -1;
        auto spec = [&](const char *name, int slot, float dst[3], float dflt) {
            std::string tn = p.texture(name);
            dst[0] = dst[1] = dst[2] = dflt;
            if (!tn.empty()) { auto it = gs.spec_tex.find(tn); if (it == gs.spec_tex.end()) throw std::runtime_error("spectrum texture \"" + tn + "\" not declared"); m.tex[slot] = it->second; }
            else p.rgb(name, dst);
        };
        auto flt = [&](const char *name, int slot, float dflt) -> float {
            std::string tn = p.texture(name);
            if (!tn.empty()) { auto it = gs.float_tex.find(tn); if (it == gs.float_tex.end()) throw std::runtime_error("float texture \"" + tn + "\" not declared"); m.tex[slot] = it->second; return dflt; }
            return p.one_float(name, dflt);
        };
        auto has = [&](const char *name) { return !p.texture(name).empty() || p.find("float", name) != nullptr; };
        float dummy[3];
        m.opacity[0] = m.opacity[1] = m.opacity[2] = 1.0f; m.eta = 1.5f; m.roughness = 0.1f; m.u_roughness = m.v_roughness = -1.0f;
        m.eta_rgb[0] = 0.2f; m.eta_rgb[1] = 0.92f; m.eta_rgb[2] = 1.1f; m.k_rgb[0] = 3.9f; m.k_rgb[1] = 2.45f; m.k_rgb[2] = 2.14f;   // copper (RGB of COPPER_N / COPPER_K)
        m.remap_roughness = p.one_bool("remaproughness", true) ? 1u : 0u;
        (void)flt("bumpmap", PT_MP_BUMP, 0.0f);
        if (kind == "matte") { m.type = PT_MAT_MATTE; spec("Kd", PT_MP_KD, m.kd, 0.5f); m.sigma = flt("sigma", PT_MP_SIGMA, 0.0f); }
        else if (kind == "plastic") { m.type = PT_MAT_PLASTIC; spec("Kd", PT_MP_KD, m.kd, 0.25f); spec("Ks", PT_MP_KS, m.ks, 0.25f); m.roughness = flt("roughness", PT_MP_ROUGHNESS, 0.1f); }
        else if (kind == "mirror") { m.type = PT_MAT_MIRROR; spec("Kr", PT_MP_KR, m.kr, 0.9f); }
        else if (kind == "glass") {
            m.type = PT_MAT_GLASS; spec("Kr", PT_MP_KR, m.kr, 1.0f); spec("Kt", PT_MP_KT, m.kt, 1.0f);
            m.eta = has("eta") ? flt("eta", PT_MP_ETA, 1.5f) : flt("index", PT_MP_ETA, 1.5f);
            m.u_roughness = flt("uroughness", PT_MP_U_ROUGHNESS, 0.0f); m.v_roughness = flt("vroughness", PT_MP_V_ROUGHNESS, 0.0f);
        }
        else if (kind == "metal") {
            m.type = PT_MAT_METAL; spec("eta", PT_MP_ETA_RGB, dummy, 0.0f); if (m.tex[PT_MP_ETA_RGB] < 0) p.rgb("eta", m.eta_rgb);
            spec("k", PT_MP_K_RGB, dummy, 0.0f); if (m.tex[PT_MP_K_RGB] < 0) p.rgb("k", m.k_rgb);
            m.roughness = flt("roughness", PT_MP_ROUGHNESS, 0.01f);
            m.u_roughness = has("uroughness") ? flt("uroughness", PT_MP_U_ROUGHNESS, -1.0f) : -1.0f; m.v_roughness = has("vroughness") ? flt("vroughness", PT_MP_V_ROUGHNESS, -1.0f) : -1.0f;
        }
        else if (kind == "uber") {
            m.type = PT_MAT_UBER; spec("Kd", PT_MP_KD, m.kd, 0.25f); spec("Ks", PT_MP_KS, m.ks, 0.25f); spec("Kr", PT_MP_KR, m.kr, 0.0f); spec("Kt", PT_MP_KT, m.kt, 0.0f);
            spec("opacity", PT_MP_OPACITY, m.opacity, 1.0f); m.roughness = flt("roughness", PT_MP_ROUGHNESS, 0.1f);
            m.u_roughness = has("uroughness") ? flt("uroughness", PT_MP_U_ROUGHNESS, -1.0f) : -1.0f; m.v_roughness = has("vroughness") ? flt("vroughness", PT_MP_V_ROUGHNESS, -1.0f) : -1.0f;
            m.eta = has("eta") ? flt("eta", PT_MP_ETA, 1.5f) : flt("index", PT_MP_ETA, 1.5f);
        }
        else if (kind == "translucent") {   // translucent.rs:82-92
            m.type = PT_MAT_TRANSLUCENT; spec("Kd", PT_MP_KD, m.kd, 0.25f); spec("Ks", PT_MP_KS, m.ks, 0.25f);
            spec("reflect", PT_MP_KR, m.kr, 0.5f); spec("transmit", PT_MP_KT, m.kt, 0.5f); m.roughness = flt("roughness", PT_MP_ROUGHNESS, 0.1f);
        }
        else if (kind == "substrate") { m.type = PT_MAT_SUBSTRATE; spec("Kd", PT_MP_KD, m.kd, 0.5f); spec("Ks", PT_MP_KS, m.ks, 0.5f);
                                        m.u_roughness = flt("uroughness", PT_MP_U_ROUGHNESS, 0.1f); m.v_roughness = flt("vroughness", PT_MP_V_ROUGHNESS, 0.1f); }
        else if (kind == "subsurface" || kind == "kdsubsurface") {
            m.type = PT_MAT_SUBSURFACE; spec("Kr", PT_MP_KR, m.kr, 1.0f); spec("Kt", PT_MP_KT, m.kt, 1.0f);
            m.u_roughness = flt("uroughness", PT_MP_U_ROUGHNESS, 0.0f); m.v_roughness = flt("vroughness", PT_MP_V_ROUGHNESS, 0.0f);
            m.eta = p.one_float("eta", 1.33f); float g = p.one_float("g", 0.0f); const float scale = p.one_float("scale", 1.0f);
            float siga[3] = {0.0011f, 0.0024f, 0.014f}, sigs[3] = {2.55f, 3.21f, 3.77f};
            if (kind == "subsurface") {   // subsurface.rs:108-139
                std::string nm = p.one_string("name", "");
                if (!nm.empty()) { auto it = named_media().find(nm); if (it != named_media().end()) { copy3(siga, it->second.sigma_a); copy3(sigs, it->second.sigma_prime_s); g = 0.0f; } }
                // get_spectrumtexture("sigma_a", siga) (subsurface.rs:127-128): a texture where one is named (evaluated at every hit), else the constant
                for (int w = 0; w < 2; ++w) {
                    const char *pn = w ? "sigma_s" : "sigma_a"; const int slot = w ? PT_MP_SIGMA_S : PT_MP_SIGMA_A;
                    const std::string tn = p.texture(pn);
                    if (!tn.empty()) { auto it = gs.spec_tex.find(tn); if (it == gs.spec_tex.end()) throw std::runtime_error("spectrum texture \"" + tn + "\" not declared"); m.tex[slot] = it->second; }
                    else p.rgb(pn, w ? sigs : siga);
                }
                m.scale = scale; m.bssrdf_table = bss_table(g, m.eta);
            } else if (!p.texture("Kd").empty() || !p.texture("mfp").empty()) {   // kdsubsurface.rs:96-99 with a textured Kd / mfp: the library converts at every hit
                spec("Kd", PT_MP_KD, m.kd, 0.5f); spec("mfp", PT_MP_MFP, m.mfp, 1.0f);
                m.kd_subsurface = 1; m.scale = scale; m.bssrdf_table = bss_table(g, m.eta);
                siga[0] = siga[1] = siga[2] = sigs[0] = sigs[1] = sigs[2] = 0.0f;
            } else {                      // kdsubsurface.rs:96-126: constant Kd / mfp -> subsurface_from_diffuse on the host
                float kd[3] = {0.5f, 0.5f, 0.5f}, mfp[3] = {1, 1, 1}; p.rgb("Kd", kd); p.rgb("mfp", mfp);
                m.bssrdf_table = bss_table(g, m.eta);
                for (int i = 0; i < 3; ++i) { kd[i] = std::fmax(kd[i], 0.0f); mfp[i] = std::fmax(mfp[i], 0.0f) * scale; }
                subsurface_from_diffuse(*sc.bss_tables[m.bssrdf_table], kd, mfp, siga, sigs);
                m.scale = 1.0f;
            }
            copy3(m.sigma_a, siga); copy3(m.sigma_s, sigs);
        }
        else if (kind == "disney") {   // disney.rs:842-887
            m.type = PT_MAT_DISNEY; spec("color", PT_MP_KD, m.kd, 0.5f); m.eta = flt("eta", PT_MP_ETA, 1.5f); m.roughness = flt("roughness", PT_MP_ROUGHNESS, 0.5f);
            static const struct { const char *name; float dflt; } ds[10] = {{"metallic", 0.0f}, {"speculartint", 0.0f}, {"anisotropic", 0.0f}, {"sheen", 0.0f}, {"sheentint", 0.5f},
                {"clearcoat", 0.0f}, {"clearcoatgloss", 1.0f}, {"spectrans", 0.0f}, {"flatness", 0.0f}, {"difftrans", 0.0f}};
            for (int k = 0; k < 10; ++k) {
                if (!p.texture(ds[k].name).empty()) throw std::runtime_error(std::string("disney: a textured \"") + ds[k].name + "\" is not supported (color, eta and roughness may be textured)");
                m.disney[k] = p.one_float(ds[k].name, ds[k].dflt);
            }
            float sd[3] = {0.0f, 0.0f, 0.0f}; p.rgb("scatterdistance", sd);
            if (!p.texture("scatterdistance").empty()) throw std::runtime_error("disney: a textured scatterdistance is not supported");
            if ((sd[0] != 0.0f || sd[1] != 0.0f || sd[2] != 0.0f) && m.tex[PT_MP_KD] >= 0) throw std::runtime_error("disney: a textured color together with scatterdistance is not supported");
            copy3(m.disney_scatter, sd);
            m.disney_thin = p.one_bool("thin", false) ? 1u : 0u;
        }
        else if (kind == "mix") {   // api.rs:615-640 + mix.rs:52-56: an undefined named material falls back to a matte made from these parameters
            m.type = PT_MAT_MIX; spec("amount", PT_MP_KD, m.kd, 0.5f);
            for (int k = 0; k < 2; ++k) {
                const std::string nm = p.one_string(k ? "namedmaterial2" : "namedmaterial1", "");
                auto it = gs.named_materials.find(nm);
                m.mix[k] = (uint32_t)(it != gs.named_materials.end() ? it->second : new_material("matte", p));
                const uint32_t ty = sc.materials[m.mix[k]].type;
                if (ty == PT_MAT_MIX || ty == PT_MAT_SUBSURFACE) throw std::runtime_error("mix of \"mix\" / subsurface materials is not supported");
            }
            m.tex[PT_MP_BUMP] = sc.materials[m.mix[0]].tex[PT_MP_BUMP];   // only the first material's bump map survives (mix.rs:31-45)
        }
        else throw std::runtime_error("material \"" + kind + "\" is not implemented on this back end");
        sc.materials.push_back(m);
        return (int)sc.materials.size() - 1;
    }
    uint32_t bss_table(float g, float eta) {
        for (size_t i = 0; i < sc.bss_keys.size(); ++i) if (sc.bss_keys[i].first == g && sc.bss_keys[i].second == eta) return (uint32_t)i;
        sc.bss_tables.emplace_back(new BssTable(compute_beam_diffusion_bssrdf(g, eta))); sc.bss_keys.push_back({g, eta});
        return (uint32_t)sc.bss_tables.size() - 1;
    }

    // ---- lights (lights/*.rs create_*) ---------------------------------------------------------------------------
    void light(const Token &d, const std::string &kind, const ParamSet &p) {
        PtLight l{}; l.prim = PT_NONE;
        gs.ctm.flat(l.light_to_world); gs.ctm.flat(l.world_to_light, true);
        const float sc_ = p.one_float("scale", 1.0f);
        if (kind == "distant") {   // distant.rs:124-132
            float L[3] = {1, 1, 1}, from[3] = {0, 0, 0}, to[3] = {0, 0, 1}; p.rgb("L", L); p.vec3("point3", "from", from); p.vec3("point3", "to", to);
            Vec3 wv = normalize(gs.ctm.vector(v3(from[0] - to[0], from[1] - to[1], from[2] - to[2])));
            l.type = PT_LIGHT_DISTANT; for (int i = 0; i < 3; ++i) l.L[i] = L[i] * sc_; l.dir[0] = wv.x; l.dir[1] = wv.y; l.dir[2] = wv.z;
        } else if (kind == "point") {   // point.rs:99-106 (translation by (P.x, P.y, P.x))
            float I[3] = {1, 1, 1}, from[3] = {0, 0, 0}; p.rgb("I", I); p.vec3("point3", "from", from);
            Transform t = gs.ctm * Transform::translate(v3(from[0], from[1], from[0]));
            Vec3 pos = t.point(v3(0, 0, 0));
            l.type = PT_LIGHT_POINT; for (int i = 0; i < 3; ++i) l.L[i] = I[i] * sc_; l.pos[0] = pos.x; l.pos[1] = pos.y; l.pos[2] = pos.z;
        } else if (kind == "spot") {   // spot.rs:118-147
            float I[3] = {1, 1, 1}, from[3] = {0, 0, 0}, to[3] = {0, 0, 1}; p.rgb("I", I); p.vec3("point3", "from", from); p.vec3("point3", "to", to);
            const float coneangle = p.one_float("coneangle", 30.0f), conedelta = p.one_float("conedeltaangle", 5.0f);
            Vec3 dir = normalize(v3(to[0] - from[0], to[1] - from[1], to[2] - from[2])), du, dv;
            if (std::fabs(dir.x) > std::fabs(dir.y)) du = v3(-dir.z, 0.0f, dir.x) / std::sqrt(dir.x * dir.x + dir.z * dir.z);
            else du = v3(0.0f, dir.z, -dir.y) / std::sqrt(dir.y * dir.y + dir.z * dir.z);
            dv = cross(dir, du);
            Mat4 m; m.m[0][0] = du.x; m.m[0][1] = du.y; m.m[0][2] = du.z; m.m[1][0] = dv.x; m.m[1][1] = dv.y; m.m[1][2] = dv.z; m.m[2][0] = dir.x; m.m[2][1] = dir.y; m.m[2][2] = dir.z;
            Transform t = gs.ctm * Transform::translate(v3(from[0], from[1], from[2])) * Transform(m).inverse();
            Vec3 pos = t.point(v3(0, 0, 0));
            l.type = PT_LIGHT_SPOT; for (int i = 0; i < 3; ++i) l.L[i] = I[i] * sc_; l.pos[0] = pos.x; l.pos[1] = pos.y; l.pos[2] = pos.z;
            const float rad = 3.14159265358979323846f / 180.0f;
            l.cos_total_width = std::cos(rad * coneangle); l.cos_falloff_start = std::cos(rad * (coneangle - conedelta));
            t.flat(l.light_to_world); t.flat(l.world_to_light, true);
        } else if (kind == "infinite" || kind == "exinfinite") {   // infinite.rs:243-259
            float L[3] = {1, 1, 1}; p.rgb("L", L); for (int i = 0; i < 3; ++i) L[i] *= sc_;
            std::string map = p.one_string("mapname", "");
            l.type = PT_LIGHT_INFINITE;
            if (map.empty()) { sc.env_w = sc.env_h = 1; sc.env_texels.assign(L, L + 3); }
            else {
                Image im = read_image(sc.base_dir + map);
                sc.env_w = (uint32_t)im.w; sc.env_h = (uint32_t)im.h; sc.env_texels.resize(im.rgb.size());
                for (size_t i = 0; i < im.rgb.size(); ++i) sc.env_texels[i] = im.rgb[i] * L[i % 3];   // infinite.rs:46-50
                if ((sc.env_w & (sc.env_w - 1)) || (sc.env_h & (sc.env_h - 1))) {   // MIPMap::new resamples to powers of two (mipmap.rs:81-140); le / importance / power read that pyramid
                    Pyramid py = build_mipmap(sc.env_texels, (int)sc.env_w, (int)sc.env_h, 3, 0);
                    sc.env_w = (uint32_t)py.width; sc.env_h = (uint32_t)py.height;
                    sc.env_texels.assign(py.texels.begin(), py.texels.begin() + (size_t)py.width * py.height * 3);
                }
            }
            sc.env_importance = env_importance(sc.env_texels, (int)sc.env_w, (int)sc.env_h);
            {   // InfiniteAreaLight::power reads map.lookup((.5,.5), .5) = triangle(levels - 2, st) (infinite.rs:103-109, mipmap.rs:202-223)
                Pyramid py = build_mipmap(sc.env_texels, (int)sc.env_w, (int)sc.env_h, 3, 0);
                if (py.n_levels == 1) copy3(sc.env_power_lookup, py.texels.data());
                else {
                    size_t off = 0; int lw = py.width, lh = py.height;
                    for (int l = 0; l < py.n_levels - 2; ++l) { off += (size_t)lw * lh * 3; lw = std::max(1, lw / 2); lh = std::max(1, lh / 2); }
                    const float s = 0.5f * (float)lw - 0.5f, t = 0.5f * (float)lh - 0.5f;
                    const long s0 = (long)std::floor(s), t0 = (long)std::floor(t); const float ds = s - (float)s0, dt = t - (float)t0;
                    auto tx = [&](long a, long b, int k) { a %= lw; if (a < 0) a += lw; b %= lh; if (b < 0) b += lh; return py.texels[off + ((size_t)b * lw + a) * 3 + k]; };
                    for (int k = 0; k < 3; ++k)
                        sc.env_power_lookup[k] = tx(s0, t0, k) * ((1.0f - ds) * (1.0f - dt)) + tx(s0, t0 + 1, k) * ((1.0f - ds) * dt) + tx(s0 + 1, t0, k) * (ds * (1.0f - dt)) + tx(s0 + 1, t0 + 1, k) * (ds * dt);
                }
            }
        } else fail(d, "LightSource: light type \"" + kind + "\" unknown.");
        sc.lights.push_back(l);
    }
    uint32_t new_area_light(uint32_t prim) {   // api.rs:1531-1546: one DiffuseAreaLight per shape
        PtLight l{}; l.type = PT_LIGHT_DIFFUSE_AREA; copy3(l.L, gs.area_L); l.two_sided = gs.area_two_sided ? 1u : 0u; l.prim = prim;
        gs.ctm.flat(l.light_to_world); gs.ctm.flat(l.world_to_light, true);
        sc.lights.push_back(l); return (uint32_t)sc.lights.size() - 1;
    }

    // ---- shapes ------------------------------------------------------------------------------------------------------
    int mask_tex(const Token &d, const ParamSet &p, const char *name) {   // triangle.rs:727-756
        std::string tn = p.texture(name);
        if (!tn.empty()) { auto it = gs.float_tex.find(tn); if (it == gs.float_tex.end()) fail(d, std::string("Couldn't find float texture \"") + tn + "\" for \"" + name + "\" parameter"); return it->second; }
        if (p.one_float(name, 1.0f) == 0.0f) { float z[3] = {0, 0, 0}; return const_tex(z); }
        return -1;
    }
    void add_prim(uint32_t shape_ref) {
        const uint32_t prim = (uint32_t)sc.prim_shape.size();
        sc.prim_shape.push_back(shape_ref); sc.prim_material.push_back(gs.material < 0 ? PT_NONE : (uint32_t)gs.material);
        { const int mi = medium_index(gs.medium_inside), mo = medium_index(gs.medium_outside); sc.prim_med_in.push_back(mi < 0 ? PT_NONE : (uint32_t)mi); sc.prim_med_out.push_back(mo < 0 ? PT_NONE : (uint32_t)mo); }
        sc.prim_light.push_back((gs.has_area && !sc.in_object) ? new_area_light(prim) : PT_NONE);   // api.rs:1605-1608: area lights inside instances are dropped
        if (sc.in_object) sc.object_ranges[sc.current_object].second += 1; else sc.top_refs.push_back(prim);
    }
    void trianglemesh(const Token &d, const std::vector<uint32_t> &idx, const std::vector<float> &P, const std::vector<float> *Np, const std::vector<float> *Sp, const std::vector<float> *UVp, int alpha, int shadow_alpha) {
        const size_t nv = P.size() / 3, nt = idx.size() / 3, v0 = sc.P.size() / 3;
        for (uint32_t ix : idx) if (ix >= nv) fail(d, "trianglemesh has out of-bounds vertex index");
        const bool has_n = Np && Np->size() == P.size(), has_s = Sp && Sp->size() == P.size(), has_uv = UVp && UVp->size() == 2 * nv;
        for (size_t i = 0; i < nv; ++i) {   // vertices are stored in world space (triangle.rs:60-73)
            Vec3 p = gs.ctm.point(v3(P[3 * i], P[3 * i + 1], P[3 * i + 2])); sc.P.push_back(p.x); sc.P.push_back(p.y); sc.P.push_back(p.z);
            Vec3 n = has_n ? gs.ctm.normal(v3((*Np)[3 * i], (*Np)[3 * i + 1], (*Np)[3 * i + 2])) : v3(0, 0, 0); sc.N.push_back(n.x); sc.N.push_back(n.y); sc.N.push_back(n.z);
            Vec3 s = has_s ? gs.ctm.vector(v3((*Sp)[3 * i], (*Sp)[3 * i + 1], (*Sp)[3 * i + 2])) : v3(0, 0, 0); sc.S.push_back(s.x); sc.S.push_back(s.y); sc.S.push_back(s.z);
            sc.UV.push_back(has_uv ? (*UVp)[2 * i] : 0.0f); sc.UV.push_back(has_uv ? (*UVp)[2 * i + 1] : 0.0f);
        }
        uint8_t fl = (gs.reverse ? PT_TRI_REVERSE_ORIENTATION : 0) | (gs.ctm.swaps_handedness() ? PT_TRI_SWAPS_HANDEDNESS : 0) | (has_n ? PT_TRI_HAS_N : 0) | (has_s ? PT_TRI_HAS_S : 0) | (has_uv ? PT_TRI_HAS_UV : 0);
        any_n |= has_n; any_s |= has_s; any_uv |= has_uv;
        for (size_t t = 0; t < nt; ++t) {
            const uint32_t tri = (uint32_t)sc.tri_flags.size();
            for (int k = 0; k < 3; ++k) sc.indices.push_back(idx[3 * t + k] + (uint32_t)v0);
            sc.tri_flags.push_back(fl); sc.tri_alpha.push_back(alpha); sc.tri_shadow_alpha.push_back(shadow_alpha);
            add_prim(((uint32_t)PT_SHAPE_TRIANGLE << 30) | tri);
        }
    }
    void shape(const Token &d, const std::string &kind, const ParamSet &p) {
        if (kind == "trianglemesh") {   // triangle.rs:700-760
            const std::vector<float> *vi = p.floats("int", "indices"), *P = p.floats("point3", "P");
            if (!vi || !P) fail(d, "trianglemesh needs \"integer indices\" and \"point P\"");
            std::vector<uint32_t> idx; for (float f : *vi) idx.push_back((uint32_t)f);
            const std::vector<float> *uv = p.floats("point2", "uv"); if (!uv) uv = p.floats("point2", "st"); if (!uv) uv = p.floats("float", "uv"); if (!uv) uv = p.floats("float", "st");
            trianglemesh(d, idx, *P, p.floats("normal", "N"), p.floats("vector3", "S"), uv, mask_tex(d, p, "alpha"), mask_tex(d, p, "shadowalpha"));
        } else if (kind == "plymesh") {
            PlyMesh m = read_ply(sc.base_dir + p.one_string("filename", ""));
            trianglemesh(d, m.indices, m.P, m.N.empty() ? nullptr : &m.N, nullptr, m.UV.empty() ? nullptr : &m.UV, mask_tex(d, p, "alpha"), mask_tex(d, p, "shadowalpha"));
        } else if (kind == "sphere") {   // sphere.rs:31-50,424-431
            const float r = p.one_float("radius", 1.0f); float zmin = p.one_float("zmin", -r), zmax = p.one_float("zmax", r); const float phimax = p.one_float("phimax", 360.0f);
            auto cl = [](float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); };
            PtSphere s{}; gs.ctm.flat(s.object_to_world); gs.ctm.flat(s.world_to_object, true);
            s.radius = r; s.z_min = cl(std::fmin(zmin, zmax), -r, r); s.z_max = cl(std::fmax(zmin, zmax), -r, r);
            s.theta_min = std::acos(cl(std::fmin(zmin, zmax) / r, -1.0f, 1.0f)); s.theta_max = std::acos(cl(std::fmax(zmin, zmax) / r, -1.0f, 1.0f));
            s.phi_max = (3.14159265358979323846f / 180.0f) * cl(phimax, 0.0f, 360.0f);
            s.reverse_orientation = gs.reverse ? 1u : 0u; s.transform_swaps_handedness = gs.ctm.swaps_handedness() ? 1u : 0u;
            sc.spheres.push_back(s);
            add_prim(((uint32_t)PT_SHAPE_SPHERE << 30) | ((uint32_t)sc.spheres.size() - 1));
        } else if (kind == "disk") {   // disk.rs:17-43,175-189: a PtSphere record of kind PT_QUADRIC_DISK
            auto cl = [](float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); };
            PtSphere s{}; gs.ctm.flat(s.object_to_world); gs.ctm.flat(s.world_to_object, true);
            s.kind = PT_QUADRIC_DISK; s.radius = p.one_float("radius", 1.0f); s.inner_radius = p.one_float("innerradius", 0.0f);
            s.z_min = s.z_max = p.one_float("height", 0.0f);
            s.phi_max = (3.14159265358979323846f / 180.0f) * cl(p.one_float("phimax", 360.0f), 0.0f, 360.0f);
            s.reverse_orientation = gs.reverse ? 1u : 0u; s.transform_swaps_handedness = gs.ctm.swaps_handedness() ? 1u : 0u;
            sc.spheres.push_back(s);
            add_prim(((uint32_t)PT_SHAPE_SPHERE << 30) | ((uint32_t)sc.spheres.size() - 1));
        } else fail(d, "shape \"" + kind + "\" is not implemented on this back end");
    }
    void object_instance(const Token &d, const std::string &name) {   // api.rs:1669-1713
        auto it = sc.object_ranges.find(name);
        if (it == sc.object_ranges.end()) fail(d, "Unable to find instance named \"" + name + "\"");
        if (it->second.second == 0) return;
        uint32_t oid = 0; for (; oid < sc.object_names.size(); ++oid) if (sc.object_names[oid] == name) break;
        if (oid == sc.object_names.size()) { sc.object_names.push_back(name); PtObject o; o.first_prim = it->second.first; o.n_prims = it->second.second; sc.objects.push_back(o); }
        PtInstance in{}; in.object = oid; gs.ctm.flat(in.instance_to_world); gs.ctm.flat(in.world_to_instance, true);
        sc.instances.push_back(in);
        sc.top_refs.push_back(PT_TOP_INSTANCE | ((uint32_t)sc.instances.size() - 1));
    }
public:
    bool any_n = false, any_s = false, any_uv = false;
};

// ---- WorldEnd: descriptor + render parameters (film.rs:55-112, perspective.rs:40-86,298-356, path.rs:225-253) ---------
static void filter_table(const std::string &name, const ParamSet &p, float radius[2], float table[256]) {
    float rx, ry;
    auto widths = [&](float d) { rx = p.one_float("xwidth", d); ry = p.one_float("ywidth", d); };
    std::function<float(float, float)> eval;
    if (name == "box") { widths(0.5f); eval = [](float, float) { return 1.0f; }; }
    else if (name == "gaussian") { widths(2.0f); const float a = p.one_float("alpha", 2.0f); const float ex = std::exp(-a * rx * rx), ey = std::exp(-a * ry * ry);
        eval = [=](float x, float y) { return std::fmax(0.0f, std::exp(-a * x * x) - ex) * std::fmax(0.0f, std::exp(-a * y * y) - ey); }; }
    else if (name == "triangle") { widths(2.0f); eval = [=](float x, float y) { return std::fmax(0.0f, rx - std::fabs(x)) * std::fmax(0.0f, ry - std::fabs(y)); }; }
    else if (name == "mitchell") {   // filters/mitchell.rs:25-38, coefficients as written there
        widths(2.0f); const float B = p.one_float("B", 1.0f / 3.0f), C = p.one_float("C", 1.0f / 3.0f);
        auto m1 = [=](float x) { const float a = std::fabs(2.0f * x);
            return a > 1.0f ? ((-B - 6.0f * C) * a * a * a + (6.0f * B * 30.0f * C) * a * a + (-12.0f * B - 48.0f * C) * a + (8.0f * B + 24.0f * C)) * (1.0f / 6.0f)
                            : ((12.0f - 9.0f * B - 6.0f * C) * a * a * a + (-18.0f + 12.0f * B + 6.0f * C) * x * x + (6.0f - 2.0f * B)) * (1.0f / 6.0f); };
        const float irx = 1.0f / rx, iry = 1.0f / ry;
        eval = [=](float x, float y) { return m1(x * irx) * m1(y * iry); };
    }
    else if (name == "sinc") {   // filters/sinc.rs:17-43, window test as written there
        widths(4.0f); const float tau = p.one_float("tau", 3.0f);
        auto sinc = [](float x) { const float y = std::fabs(x); return y < 1e-5f ? 1.0f : std::sin(3.14159265358979323846f * y) / (3.14159265358979323846f * y); };
        auto ws = [=](float x, float r) { const float y = std::fabs(x); if (y < r) return 0.0f; return sinc(y) * sinc(y / tau); };
        eval = [=](float x, float y) { return ws(x, rx) * ws(y, ry); };
    }
    else throw std::runtime_error("Filter \"" + name + "\" unknown.");
    radius[0] = rx; radius[1] = ry;
    for (int y = 0; y < 16; ++y)   // film.rs:76-89
        for (int x = 0; x < 16; ++x) table[y * 16 + x] = eval(((float)x + 0.5f) * rx / 16.0f, ((float)y + 0.5f) * ry / 16.0f);
}

static void finish(Scene &sc, const Api &api) {
    if (sc.camera_name != "perspective") throw std::runtime_error("camera \"" + sc.camera_name + "\": only \"perspective\" runs on this back end");
    if (sc.sampler != "sobol" && sc.sampler != "halton") throw std::runtime_error("sampler \"" + sc.sampler + "\": only \"sobol\" and \"halton\" run on this back end (the path depends on the GlobalSampler dimension bookkeeping)");
    PtSceneDesc &d = sc.desc; d = PtSceneDesc{};
    d.n_vertices = (uint32_t)(sc.P.size() / 3); d.P = sc.P.data();
    d.N = api.any_n ? sc.N.data() : nullptr; d.S = api.any_s ? sc.S.data() : nullptr; d.UV = api.any_uv ? sc.UV.data() : nullptr;
    d.n_triangles = (uint32_t)sc.tri_flags.size(); d.indices = sc.indices.data(); d.tri_flags = sc.tri_flags.data();
    d.n_spheres = (uint32_t)sc.spheres.size(); d.spheres = sc.spheres.data();
    d.n_prims = (uint32_t)sc.prim_shape.size(); d.prim_shape = sc.prim_shape.data(); d.prim_material = sc.prim_material.data(); d.prim_light = sc.prim_light.data();
    d.n_materials = (uint32_t)sc.materials.size(); d.materials = sc.materials.data();
    d.n_lights = (uint32_t)sc.lights.size(); d.lights = sc.lights.data();
    if (sc.env_w) { d.env_width = sc.env_w; d.env_height = sc.env_h; d.env_texels = sc.env_texels.data(); d.env_importance = sc.env_importance.data(); for (int k = 0; k < 3; ++k) d.env_power_lookup[k] = sc.env_power_lookup[k]; }
    d.max_node_prims = sc.max_node_prims; d.split_method = sc.split_method;
    for (size_t i = 0; i < sc.media_density_of.size(); ++i) sc.media[(size_t)sc.media_density_of[i]].density = sc.media_density[i].data();
    if (!sc.media.empty()) { d.n_media = (uint32_t)sc.media.size(); d.media = sc.media.data(); d.prim_medium_inside = sc.prim_med_in.data(); d.prim_medium_outside = sc.prim_med_out.data(); }
    if (!sc.instances.empty()) { d.n_objects = (uint32_t)sc.objects.size(); d.objects = sc.objects.data(); d.n_instances = (uint32_t)sc.instances.size(); d.instances = sc.instances.data(); d.n_top = (uint32_t)sc.top_refs.size(); d.top_refs = sc.top_refs.data(); }
    for (auto &t : sc.bss_tables) { PtBSSRDFTable e{}; e.n_rho = (uint32_t)t->n_rho; e.n_radius = (uint32_t)t->n_radius; e.rho_samples = t->rho_samples.data(); e.radius_samples = t->radius_samples.data(); e.profile = t->profile.data(); e.rhoeff = t->rhoeff.data(); e.profile_cdf = t->profile_cdf.data(); sc.bss_desc.push_back(e); }
    d.n_bssrdf_tables = (uint32_t)sc.bss_desc.size(); d.bssrdf_tables = sc.bss_desc.data();
    bool textured = false; for (const PtMaterial &m : sc.materials) for (int k = 0; k < 16; ++k) textured |= m.tex[k] >= 0;
    bool masks = false; for (int32_t a : sc.tri_alpha) masks |= a >= 0; bool smasks = false; for (int32_t a : sc.tri_shadow_alpha) smasks |= a >= 0;
    if (textured || masks || smasks) {
        d.n_textures = (uint32_t)sc.textures.size(); d.textures = sc.textures.data();
        for (Pyramid &p : sc.pyramids) { PtImage e{}; e.width = (uint32_t)p.width; e.height = (uint32_t)p.height; e.n_levels = (uint32_t)p.n_levels; e.channels = (uint32_t)p.channels; e.texels = p.texels.data(); sc.images.push_back(e); }
        d.n_images = (uint32_t)sc.images.size(); d.images = sc.images.data();
        if (!sc.images.empty()) { sc.ewa_lut = ewa_weight_lut(); d.ewa_weight_lut = sc.ewa_lut.data(); }
        if (masks) d.tri_alpha = sc.tri_alpha.data();
        if (smasks) d.tri_shadow_alpha = sc.tri_shadow_alpha.data();
    }
    // render parameters
    PtRenderParams &rp = sc.rp; rp = PtRenderParams{};
    rp.full_resolution[0] = sc.xres; rp.full_resolution[1] = sc.yres;
    int crop[4] = {(int)std::ceil((float)sc.xres * sc.crop[0]), (int)std::ceil((float)sc.yres * sc.crop[2]), (int)std::ceil((float)sc.xres * sc.crop[1]), (int)std::ceil((float)sc.yres * sc.crop[3])};   // film.rs:57-66
    for (int i = 0; i < 4; ++i) rp.cropped_pixel_bounds[i] = crop[i];
    filter_table(sc.filter, sc.filter_params, rp.filter_radius, rp.filter_table);
    rp.max_sample_luminance = sc.max_lum; rp.scale = sc.film_scale; rp.spp = (uint32_t)sc.spp;
    const float rx = rp.filter_radius[0], ry = rp.filter_radius[1];
    int sb[4] = {(int)std::floor((float)crop[0] + 0.5f - rx), (int)std::floor((float)crop[1] + 0.5f - ry), (int)std::ceil((float)crop[2] - 0.5f + rx), (int)std::ceil((float)crop[3] - 0.5f + ry)};   // film.rs:104-112
    for (int i = 0; i < 4; ++i) rp.sample_bounds[i] = sb[i];
    const ParamSet &cp = sc.camera_params;
    float so = cp.one_float("shutteropen", 0.0f), scl = cp.one_float("shutterclose", 1.0f); if (scl < so) std::swap(so, scl);
    const float frame = cp.one_float("frameaspectratio", (float)sc.xres / (float)sc.yres);
    float sw[4]; if (frame > 1.0f) { sw[0] = -frame; sw[1] = frame; sw[2] = -1.0f; sw[3] = 1.0f; } else { sw[0] = -1.0f; sw[1] = 1.0f; sw[2] = -1.0f / frame; sw[3] = 1.0f / frame; }
    if (const std::vector<float> *w = cp.floats("float", "screenwindow")) if (w->size() == 4) for (int i = 0; i < 4; ++i) sw[i] = (*w)[i];
    float fov = cp.one_float("fov", 90.0f); const float halffov = cp.one_float("halffov", -1.0f); if (halffov > 0.5f) fov = 2.0f * halffov;
    const Transform c2s = Transform::perspective(fov, 1e-2f, 1000.0f);
    const Transform s2r = Transform::scale((float)sc.xres, (float)sc.yres, 1.0f) * Transform::scale(1.0f / (sw[1] - sw[0]), 1.0f / (sw[2] - sw[3]), 1.0f) * Transform::translate(v3(-sw[0], -sw[3], 0.0f));
    const Transform r2c = c2s.inverse() * s2r.inverse();
    r2c.flat(rp.raster_to_camera); sc.camera_to_world.flat(rp.camera_to_world);
    rp.lens_radius = cp.one_float("lensradius", 0.0f); rp.focal_distance = cp.one_float("focaldistance", 1.0e30f); rp.shutter_open = so; rp.shutter_close = scl;
    rp.max_depth = (uint32_t)sc.maxdepth; rp.rr_threshold = sc.rr_threshold;
    rp.integrator = sc.volpath ? PT_INTEGRATOR_VOLPATH : PT_INTEGRATOR_PATH; rp.camera_medium = sc.camera_medium < 0 ? PT_NONE : (uint32_t)sc.camera_medium;
    if (!sc.has_pixel_bounds) for (int i = 0; i < 4; ++i) rp.pixel_bounds[i] = sb[i];
    else { const int *pb = sc.pixel_bounds; rp.pixel_bounds[0] = std::max(pb[0], sb[0]); rp.pixel_bounds[1] = std::max(pb[2], sb[1]); rp.pixel_bounds[2] = std::min(pb[1], sb[2]); rp.pixel_bounds[3] = std::min(pb[3], sb[3]); }   // path.rs:233-246
    rp.light_strategy = sc.strategy == "uniform" ? PT_LS_UNIFORM : sc.strategy == "power" ? PT_LS_POWER : PT_LS_SPATIAL;
    rp.tile_rank = 0; rp.tile_world = 1; rp.spp_per_pass = 0; rp.profile = 0;
    rp.sampler_type = sc.sampler == "halton" ? PT_SAMPLER_HALTON : PT_SAMPLER_SOBOL; rp.sample_at_pixel_center = (sc.sampler == "halton" && sc.sample_at_center) ? 1u : 0u;
}

}  // namespace fe

struct ptf_scene { fe::Scene sc; };

static int parse_text(const std::string &text, const std::string &base_dir, ptf_scene **out) {
    std::unique_ptr<ptf_scene> h(new ptf_scene());
    h->sc.base_dir = base_dir;
    fe::spectrum_search_dir() = base_dir;
    try {
        fe::Lexer lx(text);
        fe::Api api(h->sc);
        api.run(lx);
        fe::finish(h->sc, api);
    } catch (const std::exception &e) { fe::g_error = e.what(); return PT_ERR_INVALID_ARG; }
    *out = h.release();
    return PT_OK;
}

extern "C" {
// Parses a .pbrt file (pbrt_parse, pbrtparser.rs:26-33). File names inside resolve against the scene file's directory.
int ptf_parse_file(const char *path, ptf_scene **out) {
    if (!path || !out) { fe::g_error = "null argument"; return PT_ERR_INVALID_ARG; }
    std::ifstream f(path);
    if (!f) { fe::g_error = std::string("cannot open \"") + path + "\""; return PT_ERR_INVALID_ARG; }
    std::stringstream ss; ss << f.rdbuf();
    std::string p(path); size_t sl = p.find_last_of('/');
    return parse_text(ss.str(), sl == std::string::npos ? std::string() : p.substr(0, sl + 1), out);
}
int ptf_parse_string(const char *text, const char *base_dir, ptf_scene **out) {
    if (!text || !out) { fe::g_error = "null argument"; return PT_ERR_INVALID_ARG; }
    std::string bd = base_dir ? base_dir : ""; if (!bd.empty() && bd.back() != '/') bd += '/';
    return parse_text(text, bd, out);
}
const char *ptf_last_error(void) { return fe::g_error.c_str(); }
const PtSceneDesc *ptf_scene_desc(const ptf_scene *s) { return s ? &s->sc.desc : nullptr; }
const PtRenderParams *ptf_render_params(const ptf_scene *s) { return s ? &s->sc.rp : nullptr; }
const char *ptf_output_filename(const ptf_scene *s) { return s ? s->sc.filename.c_str() : ""; }
void ptf_scene_destroy(ptf_scene *s) { delete s; }
// rgb: width * height * 3 floats, top row first (the layout pt_film_resolve produces)
int ptf_write_pfm(const char *path, int width, int height, const float *rgb) {
    try { fe::write_pfm(path, width, height, rgb); } catch (const std::exception &e) { fe::g_error = e.what(); return PT_ERR_INVALID_ARG; }
    return PT_OK;
}
int ptf_write_image(const char *path, int width, int height, const float *rgb) {
    try { fe::write_image(path, width, height, rgb); } catch (const std::exception &e) { fe::g_error = e.what(); return PT_ERR_INVALID_ARG; }
    return PT_OK;
}
int ptf_read_image(const char *path, int *width, int *height, float *rgb, size_t capacity_floats) {
    try {
        const fe::Image im = fe::read_image(path);
        if (width) *width = im.w;
        if (height) *height = im.h;
        if (rgb) { if (capacity_floats < im.rgb.size()) { fe::g_error = "buffer too small"; return PT_ERR_INVALID_ARG; } std::memcpy(rgb, im.rgb.data(), im.rgb.size() * 4); }
    } catch (const std::exception &e) { fe::g_error = e.what(); return PT_ERR_INVALID_ARG; }
    return PT_OK;
}
}
