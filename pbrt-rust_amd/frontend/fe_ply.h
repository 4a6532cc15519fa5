// fe_ply.h -- PLY reader for Shape "plymesh" (shapes/plymesh.rs:24-160; the reference uses the ply-rs crate):
// ascii and binary_little_endian; vertex x y z [nx ny nz] [u v | s t | texture_u texture_v | texture_s texture_t];
// faces as a list property; quads split as (0,1,2),(3,0,2) (plymesh.rs:104-112), other polygons ignored.
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace fe {

struct PlyMesh { std::vector<float> P, N, UV; std::vector<uint32_t> indices; };

inline size_t ply_type_size(const std::string &t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    throw std::runtime_error("PLY: unknown property type " + t);
}
inline double ply_read_bin(const char *p, const std::string &t) {
    if (t == "char" || t == "int8") { int8_t v; std::memcpy(&v, p, 1); return v; }
    if (t == "uchar" || t == "uint8") { uint8_t v; std::memcpy(&v, p, 1); return v; }
    if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, p, 2); return v; }
    if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, p, 2); return v; }
    if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, p, 4); return v; }
    if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, p, 4); return v; }
    if (t == "float" || t == "float32") { float v; std::memcpy(&v, p, 4); return v; }
    double v; std::memcpy(&v, p, 8); return v;
}

inline PlyMesh read_ply(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("PLY file \"" + path + "\" not found");
    struct Prop { std::string name, type, count_type; bool list = false; };
    struct Elem { std::string name; size_t count = 0; std::vector<Prop> props; };
    std::vector<Elem> elems; bool ascii = false; std::string line;
    std::getline(f, line);
    if (line.substr(0, 3) != "ply") throw std::runtime_error("\"" + path + "\" is not a PLY file");
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ss(line); std::string w; ss >> w;
        if (w == "format") { std::string fm; ss >> fm; if (fm == "ascii") ascii = true; else if (fm != "binary_little_endian") throw std::runtime_error("PLY: format " + fm + " not supported"); }
        else if (w == "element") { Elem e; ss >> e.name >> e.count; elems.push_back(e); }
        else if (w == "property") {
            Prop p; std::string t; ss >> t;
            if (t == "list") { p.list = true; ss >> p.count_type >> p.type >> p.name; } else { p.type = t; ss >> p.name; }
            if (elems.empty()) throw std::runtime_error("PLY: property before element");
            elems.back().props.push_back(p);
        } else if (w == "end_header") break;
    }
    PlyMesh out; size_t vcount = 0;
    for (const Elem &e : elems) {
        const bool isv = e.name == "vertex", isf = e.name == "face";
        bool has_n = false, has_uv = false;
        if (isv) {
            vcount = e.count;
            for (const Prop &p : e.props) { if (p.name == "nx") has_n = true; if (p.name == "u" || p.name == "s" || p.name == "texture_u" || p.name == "texture_s") has_uv = true; }
            out.P.assign(3 * e.count, 0.0f); if (has_n) out.N.assign(3 * e.count, 0.0f); if (has_uv) out.UV.assign(2 * e.count, 0.0f);
        }
        for (size_t i = 0; i < e.count; ++i) {
            for (const Prop &p : e.props) {
                std::vector<double> vals;
                if (ascii) {
                    size_t n = 1; if (p.list) { double c; f >> c; n = (size_t)c; }
                    vals.resize(n); for (double &v : vals) f >> v;
                } else {
                    size_t n = 1; char buf[8];
                    if (p.list) { f.read(buf, (std::streamsize)ply_type_size(p.count_type)); n = (size_t)ply_read_bin(buf, p.count_type); }
                    vals.resize(n); const size_t sz = ply_type_size(p.type);
                    for (double &v : vals) { f.read(buf, (std::streamsize)sz); v = ply_read_bin(buf, p.type); }
                }
                if (!f) throw std::runtime_error("PLY file \"" + path + "\" is truncated");
                if (isv && !p.list) {
                    const float v = (float)vals[0];
                    if (p.name == "x") out.P[3 * i] = v; else if (p.name == "y") out.P[3 * i + 1] = v; else if (p.name == "z") out.P[3 * i + 2] = v;
                    else if (p.name == "nx") out.N[3 * i] = v; else if (p.name == "ny") out.N[3 * i + 1] = v; else if (p.name == "nz") out.N[3 * i + 2] = v;
                    else if (p.name == "u" || p.name == "s" || p.name == "texture_u" || p.name == "texture_s") out.UV[2 * i] = v;
                    else if (p.name == "v" || p.name == "t" || p.name == "texture_v" || p.name == "texture_t") out.UV[2 * i + 1] = v;
                } else if (isf && p.list && (p.name == "vertex_indices" || p.name == "vertex_index")) {
                    if (vals.size() == 3 || vals.size() == 4) {
                        out.indices.push_back((uint32_t)vals[0]); out.indices.push_back((uint32_t)vals[1]); out.indices.push_back((uint32_t)vals[2]);
                        if (vals.size() == 4) { out.indices.push_back((uint32_t)vals[3]); out.indices.push_back((uint32_t)vals[0]); out.indices.push_back((uint32_t)vals[2]); }
                    }
                }
            }
        }
    }
    if (vcount == 0 || out.indices.empty()) throw std::runtime_error("PLY file \"" + path + "\" is invalid! No face/vertex elements found");
    for (uint32_t ix : out.indices) if (ix >= vcount) throw std::runtime_error("PLY file \"" + path + "\": vertex index out of range");
    return out;
}

}  // namespace fe
