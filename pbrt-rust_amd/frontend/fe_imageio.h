// fe_imageio.h -- read_image (core/imageio.rs:18-40) for the formats a .pbrt scene can name: PFM, Radiance HDR, PNG, TGA.
// The reference delegates HDR / PNG / TGA decoding to the `image` crate (Cargo.lock: image 0.23.12, png 0.16.8), which is
// not vendored under /root/reference; what is restated here is the published file formats plus the reference's own
// conversion of the decoded pixels:
//   HDR  (imageio.rs:142-166): Rgbe8Pixel::to_hdr = c * 2^(e - 136) per channel, (0,0,0) when e == 0 (image crate; note:
//        no +0.5 as in Ward's original); new-style RLE, old-style repeat pixels and flat scanlines.
//   PNG / TGA (imageio.rs:338-357): to_rgb8 then v / 255 per channel (grey replicated, alpha dropped, palette expanded,
//        16-bit samples reduced to their high byte's rounding v * 255 / 65535).
//   EXR: refused with a message (the exr crate's codecs are not restated).
// zlib (system library, -lz) inflates PNG IDAT streams.
#pragma once
#include <cstdint>
#include <cstring>
#include <zlib.h>
#include "fe_image.h"

namespace fe {

inline std::vector<unsigned char> read_file(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("image \"" + path + "\" not found");
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

inline Image read_hdr(const std::string &path) {
    const std::vector<unsigned char> d = read_file(path);
    size_t p = 0;
    auto line = [&]() { std::string s; while (p < d.size() && d[p] != '\n') s.push_back((char)d[p++]); if (p < d.size()) p++; return s; };
    std::string first = line();
    if (first.rfind("#?", 0) != 0) throw std::runtime_error("\"" + path + "\": not a Radiance HDR file");
    for (;;) { if (p >= d.size()) throw std::runtime_error("\"" + path + "\": truncated HDR header"); if (line().empty()) break; }
    int w = 0, h = 0; char sy = 0, sx = 0, ay = 0, ax = 0;
    const std::string dims = line();
    if (std::sscanf(dims.c_str(), "%c%c %d %c%c %d", &sy, &ay, &h, &sx, &ax, &w) != 6 || ay != 'Y' || ax != 'X' || w <= 0 || h <= 0)
        throw std::runtime_error("\"" + path + "\": unsupported HDR orientation line \"" + dims + "\"");
    std::vector<unsigned char> rgbe((size_t)w * h * 4);
    for (int y = 0; y < h; ++y) {
        unsigned char *row = rgbe.data() + (size_t)y * w * 4;
        if (p + 4 > d.size()) throw std::runtime_error("HDR \"" + path + "\" is truncated");
        if (w >= 8 && w < 32768 && d[p] == 2 && d[p + 1] == 2 && (d[p + 2] & 0x80) == 0 && ((d[p + 2] << 8) | d[p + 3]) == w) {   // new-style RLE: 4 planes
            p += 4;
            for (int c = 0; c < 4; ++c)
                for (int x = 0; x < w;) {
                    if (p >= d.size()) throw std::runtime_error("HDR \"" + path + "\" is truncated");
                    int n = d[p++];
                    if (n > 128) { n -= 128; if (x + n > w || p >= d.size()) throw std::runtime_error("HDR \"" + path + "\": bad run"); const unsigned char v = d[p++]; for (int i = 0; i < n; ++i) row[(x++) * 4 + c] = v; }
                    else { if (n == 0 || x + n > w || p + n > d.size()) throw std::runtime_error("HDR \"" + path + "\": bad run"); for (int i = 0; i < n; ++i) row[(x++) * 4 + c] = d[p++]; }
                }
        } else {   // flat pixels with old-style (1,1,1,count) repeats
            int shift = 0;
            for (int x = 0; x < w;) {
                if (p + 4 > d.size()) throw std::runtime_error("HDR \"" + path + "\" is truncated");
                const unsigned char *px = d.data() + p; p += 4;
                if (px[0] == 1 && px[1] == 1 && px[2] == 1 && x > 0) {
                    const int n = (int)px[3] << shift;
                    for (int i = 0; i < n && x < w; ++i, ++x) std::memcpy(row + x * 4, row + (x - 1) * 4, 4);
                    shift += 8;
                } else { std::memcpy(row + x * 4, px, 4); x++; shift = 0; }
            }
        }
    }
    Image im; im.w = w; im.h = h; im.rgb.resize((size_t)w * h * 3);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int sy2 = sy == '-' ? y : h - 1 - y, sx2 = sx == '+' ? x : w - 1 - x;   // "-Y h +X w" is top-to-bottom, left-to-right
            const unsigned char *q = rgbe.data() + ((size_t)y * w + x) * 4;
            float *o = im.rgb.data() + ((size_t)sy2 * w + sx2) * 3;
            if (q[3] == 0) { o[0] = o[1] = o[2] = 0.0f; continue; }
            const float e = std::exp2((float)q[3] - 128.0f - 8.0f);
            o[0] = e * (float)q[0]; o[1] = e * (float)q[1]; o[2] = e * (float)q[2];
        }
    return im;
}

inline Image read_png(const std::string &path) {
    const std::vector<unsigned char> d = read_file(path);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) throw std::runtime_error("\"" + path + "\": not a PNG file");
    auto be32 = [&](size_t o) { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; };
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte;
    for (size_t p = 8; p + 12 <= d.size();) {
        const uint32_t len = be32(p); const std::string type((const char *)d.data() + p + 4, 4);
        if (p + 12 + len > d.size()) throw std::runtime_error("PNG \"" + path + "\" is truncated");
        const unsigned char *body = d.data() + p + 8;
        if (type == "IHDR") { w = be32(p + 8); h = be32(p + 12); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (type == "PLTE") plte.assign(body, body + len);
        else if (type == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (type == "IEND") break;
        p += 12 + len;
    }
    if (!w || !h) throw std::runtime_error("PNG \"" + path + "\": no IHDR");
    if (interlace) throw std::runtime_error("PNG \"" + path + "\": interlaced files are not supported");
    const int nch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!nch || (depth != 8 && depth != 16 && !(depth < 8 && (ctype == 0 || ctype == 3)))) throw std::runtime_error("PNG \"" + path + "\": unsupported colour type / depth");
    const size_t bpp_bits = (size_t)nch * depth, stride = (w * bpp_bits + 7) / 8, bpp = std::max<size_t>(1, bpp_bits / 8);
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) throw std::runtime_error("PNG \"" + path + "\": corrupt image data");
    std::vector<unsigned char> img(stride * h), zero(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {   // undo the scanline filters
        const unsigned char *in = raw.data() + (stride + 1) * y + 1, *up = y ? img.data() + stride * (y - 1) : zero.data();
        unsigned char *out = img.data() + stride * y;
        const int ft = raw[(stride + 1) * y];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? out[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a; else if (ft == 2) pred = b; else if (ft == 3) pred = (a + b) / 2;
            else if (ft == 4) { const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) throw std::runtime_error("PNG \"" + path + "\": bad filter type");
            out[i] = (unsigned char)(in[i] + pred);
        }
    }
    Image im; im.w = (int)w; im.h = (int)h; im.rgb.resize((size_t)w * h * 3);
    auto sample = [&](uint32_t y, uint32_t x, int ch) -> int {   // 8-bit value of one channel (to_rgb8)
        const unsigned char *row = img.data() + stride * y;
        if (depth == 8) return row[(size_t)x * nch + ch];
        if (depth == 16) { const int v = (row[((size_t)x * nch + ch) * 2] << 8) | row[((size_t)x * nch + ch) * 2 + 1]; return (v * 255 + 32767) / 65535; }
        const int per = 8 / depth, v = (row[x / per] >> (8 - depth - (x % per) * depth)) & ((1 << depth) - 1);
        return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
    };
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            int r, g, b;
            if (ctype == 3) { const size_t k = (size_t)sample(y, x, 0) * 3; if (k + 3 > plte.size()) throw std::runtime_error("PNG \"" + path + "\": palette index out of range"); r = plte[k]; g = plte[k + 1]; b = plte[k + 2]; }
            else if (nch <= 2) r = g = b = sample(y, x, 0);
            else { r = sample(y, x, 0); g = sample(y, x, 1); b = sample(y, x, 2); }
            float *o = im.rgb.data() + ((size_t)y * w + x) * 3;
            o[0] = (float)r / 255.0f; o[1] = (float)g / 255.0f; o[2] = (float)b / 255.0f;
        }
    return im;
}

inline Image read_tga(const std::string &path) {
    const std::vector<unsigned char> d = read_file(path);
    if (d.size() < 18) throw std::runtime_error("TGA \"" + path + "\" is truncated");
    const int id_len = d[0], cmap = d[1], type = d[2], w = d[12] | (d[13] << 8), h = d[14] | (d[15] << 8), bits = d[16], desc = d[17];
    if (cmap || (type != 2 && type != 3 && type != 10 && type != 11) || !(bits == 8 || bits == 24 || bits == 32) || w <= 0 || h <= 0) throw std::runtime_error("TGA \"" + path + "\": unsupported variant");
    const int bpp = bits / 8;
    std::vector<unsigned char> px((size_t)w * h * bpp);
    size_t p = 18 + (size_t)id_len;
    if (type < 8) { if (p + px.size() > d.size()) throw std::runtime_error("TGA \"" + path + "\" is truncated"); std::memcpy(px.data(), d.data() + p, px.size()); }
    else for (size_t o = 0; o < px.size();) {
        if (p >= d.size()) throw std::runtime_error("TGA \"" + path + "\" is truncated");
        const int hd = d[p++], n = (hd & 127) + 1;
        if (o + (size_t)n * bpp > px.size()) throw std::runtime_error("TGA \"" + path + "\": bad run");
        if (hd & 128) { if (p + bpp > d.size()) throw std::runtime_error("TGA \"" + path + "\" is truncated"); for (int i = 0; i < n; ++i) { std::memcpy(px.data() + o, d.data() + p, bpp); o += bpp; } p += bpp; }
        else { if (p + (size_t)n * bpp > d.size()) throw std::runtime_error("TGA \"" + path + "\" is truncated"); std::memcpy(px.data() + o, d.data() + p, (size_t)n * bpp); o += (size_t)n * bpp; p += (size_t)n * bpp; }
    }
    Image im; im.w = w; im.h = h; im.rgb.resize((size_t)w * h * 3);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int sy = (desc & 0x20) ? y : h - 1 - y, sx = (desc & 0x10) ? w - 1 - x : x;   // default origin: bottom left
            const unsigned char *q = px.data() + ((size_t)y * w + x) * bpp;
            float *o = im.rgb.data() + ((size_t)sy * w + sx) * 3;
            if (bpp == 1) o[0] = o[1] = o[2] = (float)q[0] / 255.0f;
            else { o[0] = (float)q[2] / 255.0f; o[1] = (float)q[1] / 255.0f; o[2] = (float)q[0] / 255.0f; }   // stored BGR(A)
        }
    return im;
}

// core/imageio.rs:18-40: dispatch on the file name extension
inline Image read_image(const std::string &path) {
    const size_t dot = path.find_last_of('.');
    const std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
    if (ext == "pfm") return read_pfm(path);
    if (ext == "hdr") return read_hdr(path);
    if (ext == "png" || ext == "PNG") return read_png(path);
    if (ext == "tga" || ext == "TGA") return read_tga(path);
    if (ext == "exr" || ext == "EXR") throw std::runtime_error("\"" + path + "\": OpenEXR files are not read by this front end (convert to .pfm or .hdr)");
    throw std::runtime_error("\"" + path + "\": unable to load image with this extension (imageio.rs:33-37)");
}

}  // namespace fe
