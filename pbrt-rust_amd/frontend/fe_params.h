// fe_params.h -- tokenizer and parameter lists of the .pbrt front end.
//   pbrtparser/lexer.rs:9-60 (token classes: directive keywords, "quoted strings", numbers, [ ], # comments),
//   pbrtparser/pbrtparser.rs:163-330 (param_item: "type name" declarations; int|integer, bool, float, point2, vector2,
//   point3|point, vector3|vector, normal, rgb|color, xyz, blackbody, spectrum (all kept as RGB, fe_spectrum.h), string, texture),
//   core/paramset.rs (find_one_*, find_* with defaults).
#pragma once
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "fe_spectrum.h"

namespace fe {

// directory that relative .spd file names of "spectrum" parameters resolve against (resolve_filename: the scene file's directory)
inline std::string &spectrum_search_dir() { static thread_local std::string d; return d; }

struct Token { enum Kind { Word, Str, Num, LBracket, RBracket, End } kind = End; std::string text; float num = 0; int line = 0; };

class Lexer {
public:
    explicit Lexer(const std::string &t) : s(t) {}
    Token next() {
        for (;;) {  // whitespace and # comments
            while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\r' || s[pos] == '\n' || s[pos] == '\f' || s[pos] == '\v')) { if (s[pos] == '\n') ++line; ++pos; }
            if (pos < s.size() && s[pos] == '#') { while (pos < s.size() && s[pos] != '\n') ++pos; continue; }
            break;
        }
        Token t; t.line = line;
        if (pos >= s.size()) return t;
        const char c = s[pos];
        if (c == '[') { ++pos; t.kind = Token::LBracket; return t; }
        if (c == ']') { ++pos; t.kind = Token::RBracket; return t; }
        if (c == '"') {
            size_t e = s.find('"', pos + 1);
            if (e == std::string::npos) throw std::runtime_error("line " + std::to_string(line) + ": unterminated string");
            t.kind = Token::Str; t.text = s.substr(pos + 1, e - pos - 1); pos = e + 1; return t;
        }
        if ((c >= '0' && c <= '9') || c == '+' || c == '-' || c == '.') {  // lexer.rs NUMBER regex
            char *endp = nullptr;
            const float v = std::strtof(s.c_str() + pos, &endp);
            if (endp != s.c_str() + pos) { t.kind = Token::Num; t.num = v; pos = (size_t)(endp - s.c_str()); return t; }
        }
        size_t e = pos;
        while (e < s.size() && ((s[e] >= 'A' && s[e] <= 'Z') || (s[e] >= 'a' && s[e] <= 'z'))) ++e;
        if (e == pos) throw std::runtime_error("line " + std::to_string(line) + ": unexpected character '" + std::string(1, c) + "'");
        t.kind = Token::Word; t.text = s.substr(pos, e - pos); pos = e; return t;
    }
    Token peek() { size_t p = pos; int l = line; Token t = next(); pos = p; line = l; return t; }
private:
    const std::string &s; size_t pos = 0; int line = 1;
};

struct Param { std::string type, name; std::vector<float> nums; std::vector<std::string> strs; bool looked_up = false; };

class ParamSet {
public:
    std::vector<Param> items;
    void add(const std::string &decl, std::vector<float> nums, std::vector<std::string> strs) {
        size_t sp = decl.find_first_of(" \t");
        if (sp == std::string::npos) throw std::runtime_error("parameter \"" + decl + "\" needs a type and a name");
        Param p; p.type = decl.substr(0, sp);
        size_t ns = decl.find_first_not_of(" \t", sp);
        p.name = decl.substr(ns, decl.find_first_of(" \t", ns) - ns);
        static const std::map<std::string, std::string> canon = {{"int", "int"}, {"integer", "int"}, {"bool", "bool"}, {"float", "float"}, {"vector2", "vector2"},
            {"vector3", "vector3"}, {"vector", "vector3"}, {"point2", "point2"}, {"point3", "point3"}, {"point", "point3"}, {"normal", "normal"},
            {"rgb", "rgb"}, {"color", "rgb"}, {"string", "string"}, {"texture", "texture"}, {"xyz", "xyz"}, {"blackbody", "blackbody"}, {"spectrum", "spectrum"}};
        auto it = canon.find(p.type);
        if (it == canon.end()) throw std::runtime_error("unknown parameter type " + p.type);   // pbrtparser.rs:182
        p.type = it->second;
        // Spectrum = RGBSpectrum: every spectral parameter type is stored as RGB, in ONE namespace with "rgb" (ParamSet::spectra), so a
        // later `find_one_spectrum(name)` sees it whatever type it was declared with (pbrtparser.rs:325-377, paramset.rs:145-246)
        if (p.type == "xyz") {
            nums.resize(nums.size() - nums.size() % 3);
            for (size_t i = 0; i + 2 < nums.size(); i += 3) { float rgb[3]; xyz_to_rgb(&nums[i], rgb); nums[i] = rgb[0]; nums[i + 1] = rgb[1]; nums[i + 2] = rgb[2]; }
            p.type = "rgb";
        } else if (p.type == "blackbody") {
            nums.resize(nums.size() - nums.size() % 2);
            std::vector<float> out;
            for (size_t i = 0; i + 1 < nums.size(); i += 2) { float rgb[3]; rgb_from_blackbody(nums[i], nums[i + 1], rgb); out.insert(out.end(), rgb, rgb + 3); }
            nums = std::move(out); p.type = "rgb";
        } else if (p.type == "spectrum") {
            std::vector<float> out;
            if (!strs.empty()) { for (const std::string &fn : strs) { float rgb[3]; rgb_from_spd_file((!fn.empty() && fn[0] == '/') ? fn : spectrum_search_dir() + fn, rgb); out.insert(out.end(), rgb, rgb + 3); } strs.clear(); }
            else {
                nums.resize(nums.size() - nums.size() % 2);
                std::vector<float> wl, v;
                for (size_t i = 0; i + 1 < nums.size(); i += 2) { wl.push_back(nums[i]); v.push_back(nums[i + 1]); }
                float rgb[3]; rgb_from_sampled(wl, v, rgb); out.assign(rgb, rgb + 3);
            }
            nums = std::move(out); p.type = "rgb";
        }
        const size_t arity = (p.type == "vector3" || p.type == "point3" || p.type == "normal" || p.type == "rgb") ? 3 : (p.type == "vector2" || p.type == "point2") ? 2 : 1;
        if (arity > 1) nums.resize(nums.size() - nums.size() % arity);   // excess values are dropped with a warning (pbrtparser.rs:214-330)
        p.nums = std::move(nums); p.strs = std::move(strs);
        for (size_t i = 0; i < items.size(); ++i) if (items[i].type == p.type && items[i].name == p.name) { items.erase(items.begin() + (long)i); break; }   // add_* erases an earlier item of the same kind and name
        items.push_back(std::move(p));
    }
    const Param *find(const std::string &type, const std::string &name) const {
        for (const Param &p : items) if (p.type == type && p.name == name) { const_cast<Param &>(p).looked_up = true; return &p; }
        return nullptr;
    }
    float one_float(const std::string &n, float d) const { const Param *p = find("float", n); return (p && p->nums.size() == 1) ? p->nums[0] : d; }
    int one_int(const std::string &n, int d) const { const Param *p = find("int", n); return (p && p->nums.size() == 1) ? (int)p->nums[0] : d; }   // `x as isize`
    bool one_bool(const std::string &n, bool d) const { const Param *p = find("bool", n); return (p && p->strs.size() == 1) ? p->strs[0] == "true" : d; }
    std::string one_string(const std::string &n, const std::string &d) const { const Param *p = find("string", n); return (p && p->strs.size() == 1) ? p->strs[0] : d; }
    std::string texture(const std::string &n) const { const Param *p = find("texture", n); return (p && p->strs.size() == 1) ? p->strs[0] : std::string(); }
    const std::vector<float> *floats(const std::string &type, const std::string &n) const { const Param *p = find(type, n); return p ? &p->nums : nullptr; }
    bool rgb(const std::string &n, float out[3]) const { const Param *p = find("rgb", n); if (!p || p->nums.size() < 3) return false; out[0] = p->nums[0]; out[1] = p->nums[1]; out[2] = p->nums[2]; return true; }
    bool vec3(const std::string &type, const std::string &n, float out[3]) const { const Param *p = find(type, n); if (!p || p->nums.size() < 3) return false; out[0] = p->nums[0]; out[1] = p->nums[1]; out[2] = p->nums[2]; return true; }
};

// Reads `"type name" value | [ values ]` pairs until the next directive keyword.
inline ParamSet read_params(Lexer &lx) {
    ParamSet ps;
    for (;;) {
        Token t = lx.peek();
        if (t.kind != Token::Str) break;
        lx.next();
        std::vector<float> nums; std::vector<std::string> strs;
        Token v = lx.next();
        if (v.kind == Token::LBracket) {
            for (;;) {
                Token e = lx.next();
                if (e.kind == Token::RBracket) break;
                if (e.kind == Token::Num) nums.push_back(e.num);
                else if (e.kind == Token::Str) strs.push_back(e.text);
                else if (e.kind == Token::Word && (e.text == "true" || e.text == "false")) strs.push_back(e.text);
                else throw std::runtime_error("line " + std::to_string(e.line) + ": bad value in parameter array");
            }
        } else if (v.kind == Token::Num) nums.push_back(v.num);
        else if (v.kind == Token::Str) strs.push_back(v.text);
        else throw std::runtime_error("line " + std::to_string(v.line) + ": parameter \"" + t.text + "\" has no value");
        ps.add(t.text, std::move(nums), std::move(strs));
    }
    return ps;
}

}  // namespace fe
