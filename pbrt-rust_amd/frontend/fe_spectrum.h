// fe_spectrum.h -- "xyz" / "blackbody" / "spectrum" parameter values converted to RGB on the host exactly as the reference's parameter
// sets do (Spectrum = RGBSpectrum): pbrtparser.rs:325-377 (add_xyz / add_blackbody / add_spectrum), core/paramset.rs:145-246,
// core/spectrum.rs:36-70 (black_body, black_body_normalized), :108-113 (from_xyz), :129-154 (RGBSpectrum::from_sampled), :411-434
// (sorted check, sort), :469-483 (interpolate_spectrum_samples), :485-493 (xyz_to_rgb), core/floatfile.rs (read_float_file).
// Quirks of the reference kept on purpose (a drop-in has to read the same scene the same way):
//   * from_sampled on unsorted wavelengths sorts (lambda, value) pairs but then recurses with the SORTED wavelengths and the
//     UNSORTED values (spectrum.rs:131-136);
//   * read_float_file pushes every parsed number twice (floatfile.rs:21-29), so an .spd file "l0 v0 l1 v1 ..." arrives as
//     wavelengths (l0, v0, l1, v1, ...) with the same list as values.
#pragma once
#include <algorithm>
#include <cmath>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "../../include/pt_cie_tables.h"

namespace fe {

inline const float *cie_x() { static const float t[PT_N_CIE_SAMPLES] = {PT_CIE_X_VALUES}; return t; }
inline const float *cie_y() { static const float t[PT_N_CIE_SAMPLES] = {PT_CIE_Y_VALUES}; return t; }
inline const float *cie_z() { static const float t[PT_N_CIE_SAMPLES] = {PT_CIE_Z_VALUES}; return t; }
inline const float *cie_lambda() { static const float t[PT_N_CIE_SAMPLES] = {PT_CIE_LAMBDA_VALUES}; return t; }

inline void xyz_to_rgb(const float xyz[3], float rgb[3]) {   // spectrum.rs:485-493
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}

inline void black_body(const float *lambda, int n, float t, float *le) {   // spectrum.rs:36-58 (f32 arithmetic as written)
    if (t <= 0.0f) { for (int i = 0; i < n; ++i) le[i] = 0.0f; return; }
    const float c = 299792458.0f, h = 6.62606957e-34f, kb = 1.3806488e-23f;
    for (int i = 0; i < n; ++i) {
        const float l = (float)((double)lambda[i] * (double)1.0e-9f);
        const float lambda5 = (l * l) * (l * l) * l;
        const float e = std::exp((h * c) / (l * kb * t));
        le[i] = (2.0f * h * c * c) / (lambda5 * (e - 1.0f));
    }
}
inline void black_body_normalized(const float *lambda, int n, float t, float *le) {   // spectrum.rs:60-70
    black_body(lambda, n, t, le);
    const float lambda_max = 2.8977721e-3f / t * 1.0e9f;
    float maxl = 0.0f;
    black_body(&lambda_max, 1, t, &maxl);
    for (int i = 0; i < n; ++i) le[i] /= maxl;
}

inline float interpolate_spectrum_samples(const float *lambda, const float *vals, int n, float l) {   // spectrum.rs:469-483
    if (l <= lambda[0]) return vals[0];
    if (l >= lambda[n - 1]) return vals[n - 1];
    // find_interval(n, |i| lambda[i] <= l) (pbrt.rs find_interval: clamp(first - 1, 0, n - 2))
    int first = 0, len = n;
    while (len > 0) { const int half = len >> 1, middle = first + half; if (lambda[middle] <= l) { first = middle + 1; len -= half + 1; } else len = half; }
    const int offset = std::min(std::max(first - 1, 0), n - 2);
    const float t = (l - lambda[offset]) / (lambda[offset + 1] - lambda[offset]);
    return (1.0f - t) * vals[offset] + t * vals[offset + 1];   // lerp (pbrt.rs:136-144)
}

inline void rgb_from_sampled(const std::vector<float> &lambda, const std::vector<float> &v, float rgb[3], int depth = 0) {   // spectrum.rs:129-154
    const int n = (int)lambda.size();
    if (n == 0) { rgb[0] = rgb[1] = rgb[2] = 0.0f; return; }
    bool sorted = true;
    for (int i = 0; i + 1 < n; ++i) if (lambda[i] > lambda[i + 1]) { sorted = false; break; }
    if (!sorted && depth == 0) {
        std::vector<std::pair<float, float>> sv(n);
        for (int i = 0; i < n; ++i) sv[i] = {lambda[i], v[i]};
        std::sort(sv.begin(), sv.end());   // sort_by(partial_cmp) over (lambda, value) tuples
        std::vector<float> sl(n);
        for (int i = 0; i < n; ++i) sl[i] = sv[i].first;
        rgb_from_sampled(sl, v, rgb, 1);   // the reference passes the sorted wavelengths with the UNSORTED values
        return;
    }
    float xyz[3] = {0.0f, 0.0f, 0.0f};
    for (int i = 0; i < PT_N_CIE_SAMPLES; ++i) {
        const float val = interpolate_spectrum_samples(lambda.data(), v.data(), n, cie_lambda()[i]);
        xyz[0] += val * cie_x()[i]; xyz[1] += val * cie_y()[i]; xyz[2] += val * cie_z()[i];
    }
    const float scale = (cie_lambda()[PT_N_CIE_SAMPLES - 1] - cie_lambda()[0]) / (PT_CIE_Y_INTEGRAL * (float)PT_N_CIE_SAMPLES);
    xyz[0] *= scale; xyz[1] *= scale; xyz[2] *= scale;
    xyz_to_rgb(xyz, rgb);
}

inline void rgb_from_blackbody(float temperature, float scale, float rgb[3]) {   // paramset.rs:163-180
    std::vector<float> lam(cie_lambda(), cie_lambda() + PT_N_CIE_SAMPLES), v(PT_N_CIE_SAMPLES);
    black_body_normalized(lam.data(), PT_N_CIE_SAMPLES, temperature, v.data());
    rgb_from_sampled(lam, v, rgb);
    for (int k = 0; k < 3; ++k) rgb[k] *= scale;
}

// floatfile.rs:9-37: whitespace-separated floats, lines starting with '#' skipped; every value lands in the list TWICE (the quirk
// noted above); a token that is not a number fails the whole file (the caller then uses a black spectrum, paramset.rs:215-222).
inline bool read_float_file(const std::string &path, std::vector<float> &values) {
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ts(line);
        std::string tok;
        while (ts >> tok) {
            char *endp = nullptr;
            const float val = std::strtof(tok.c_str(), &endp);
            if (endp == tok.c_str() || *endp != '\0') return false;
            values.push_back(val); values.push_back(val);
        }
    }
    return true;
}

// `"spectrum name" "file.spd"` (paramset.rs:199-246): one RGB per file name
inline void rgb_from_spd_file(const std::string &path, float rgb[3]) {
    std::vector<float> vals;
    if (!read_float_file(path, vals)) { rgb[0] = rgb[1] = rgb[2] = 0.0f; return; }
    std::vector<float> wl, v;
    for (size_t j = 0; j < vals.size() / 2; ++j) { wl.push_back(vals[2 * j]); v.push_back(vals[2 * j + 1]); }
    rgb_from_sampled(wl, v, rgb);
}

}  // namespace fe
