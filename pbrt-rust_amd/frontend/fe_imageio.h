// fe_imageio.h -- read_image (core/imageio.rs:18-40) for the formats a .pbrt scene can name: PFM, Radiance HDR, PNG, TGA.
// The reference delegates HDR / PNG / TGA decoding to the `image` crate (Cargo.lock: image 0.23.12, png 0.16.8), which is
// not vendored under /root/reference; what is restated here is the published file formats plus the reference's own
// conversion of the decoded pixels:
//   HDR  (imageio.rs:142-166): Rgbe8Pixel::to_hdr = c * 2^(e - 136) per channel, (0,0,0) when e == 0 (image crate; note:
//        no +0.5 as in Ward's original); new-style RLE, old-style repeat pixels and flat scanlines.
//   PNG / TGA (imageio.rs:338-357): to_rgb8 then v / 255 per channel (grey replicated, alpha dropped, palette expanded,
//        16-bit samples reduced to their high byte's rounding v * 255 / 65535).
//   EXR  (imageio.rs:68-114, `exr` crate 1.0.0): scan-line and tiled files, single- and multi-part, uncompressed / RLE / ZIPS / ZIP, HALF / FLOAT / UINT channels are
//        read; three uncompressed FLOAT channels are written.
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
    if ((uint64_t)w * (uint64_t)h > (1ull << 28)) throw std::runtime_error("\"" + path + "\": HDR image of " + std::to_string(w) + " x " + std::to_string(h) + " pixels is larger than 2^28 pixels");
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
                    if (shift > 24) throw std::runtime_error("HDR \"" + path + "\": bad run");   // a fifth consecutive repeat record: the count no longer fits 32 bits
                    const int64_t n = (int64_t)px[3] << shift;
                    for (int64_t i = 0; i < n && x < w; ++i, ++x) std::memcpy(row + x * 4, row + (x - 1) * 4, 4);
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
        if (type == "IHDR") {
            if (len != 13) throw std::runtime_error("PNG \"" + path + "\": IHDR chunk of " + std::to_string(len) + " bytes");
            w = be32(p + 8); h = be32(p + 12); depth = body[8]; ctype = body[9]; interlace = body[12];
            if ((uint64_t)w * (uint64_t)h > (1ull << 28)) throw std::runtime_error("PNG \"" + path + "\": larger than 2^28 pixels");
        }
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


// ---- OpenEXR (scan-line files; the `exr` crate 1.0.0 of Cargo.lock is not vendored: the published file layout is restated) ----
inline float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t bits;
    if (e == 0) { if (m == 0) bits = sign; else { int sh = 0; uint32_t mm = m; while (!(mm & 1024u)) { mm <<= 1; sh++; } bits = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 1023u) << 13); } }
    else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
    else bits = sign | ((e + 112u) << 23) | (m << 13);
    float f; std::memcpy(&f, &bits, 4); return f;
}
// read_image_exr (imageio.rs:68-97; the `exr` crate reads every flat image): R, G, B (or Y) channels -- HALF / FLOAT / UINT -- of the first flat
// part of a scan-line or TILED file (level 0 of a mip / rip map), single- or multi-part, with NO, RLE, ZIPS or ZIP compression. Deep data and the
// PIZ / PXR24 / B44 / DWA codecs are refused by name.
inline Image read_exr(const std::string &path) {
    const std::vector<unsigned char> d = read_file(path);
    auto need = [&](size_t p, size_t n) { if (p + n > d.size() || p + n < p) throw std::runtime_error("EXR \"" + path + "\" is truncated"); };
    auto i32 = [&](size_t p) { need(p, 4); int32_t v; std::memcpy(&v, d.data() + p, 4); return v; };
    need(0, 8);
    if (i32(0) != 20000630) throw std::runtime_error("\"" + path + "\": not an OpenEXR file");
    const int32_t version = i32(4);
    const bool multipart = (version & 0x1000) != 0;
    if (version & 0x800) throw std::runtime_error("EXR \"" + path + "\": deep data is not supported");   // (any file with a deep part sets this bit: none gets past here)
    struct Chan { std::string name; int type; };
    struct Part { std::vector<Chan> chans; int compression = -1; int32_t win[4] = {0, 0, -1, -1}; bool tiled = false, deep = false; uint32_t tile_w = 0, tile_h = 0; int level_mode = 0, round_up = 0; int32_t chunk_count = -1; };
    std::vector<Part> parts;
    size_t p = 8;
    for (;;) {   // one header per part; a multi-part file ends the list with an empty header
        Part pt; pt.tiled = !multipart && (version & 0x200) != 0;
        bool any = false;
        for (;;) {
            need(p, 1);
            if (d[p] == 0) { p++; break; }
            any = true;
            auto cstr = [&]() { std::string t; while (true) { need(p, 1); if (!d[p]) { p++; break; } t.push_back((char)d[p++]); } return t; };
            const std::string name = cstr(), type = cstr(); const int32_t size = i32(p); p += 4;
            if (size < 0) throw std::runtime_error("EXR \"" + path + "\": bad attribute size");
            need(p, (size_t)size);
            if (name == "channels") {   // (name NUL, type, pLinear + 3 reserved bytes, xSampling, ySampling) ..., NUL -- every read stays inside the attribute
                const size_t end = p + (size_t)size; size_t q = p;
                auto in_attr = [&](size_t at, size_t n) { if (at + n > end || at + n < at) throw std::runtime_error("EXR \"" + path + "\": channel list runs past its attribute"); };
                for (;;) {
                    in_attr(q, 1);
                    if (!d[q]) break;
                    Chan c;
                    for (;;) { in_attr(q, 1); if (!d[q]) break; c.name.push_back((char)d[q++]); }
                    q++;
                    in_attr(q, 16);
                    c.type = i32(q);
                    if (i32(q + 8) != 1 || i32(q + 12) != 1) throw std::runtime_error("EXR \"" + path + "\": subsampled channels are not supported");
                    q += 16; pt.chans.push_back(c);
                }
            }
            else if (name == "compression") { if (size < 1) throw std::runtime_error("EXR \"" + path + "\": empty compression attribute"); pt.compression = d[p]; }
            else if (name == "dataWindow") { if (size < 16) throw std::runtime_error("EXR \"" + path + "\": short dataWindow attribute"); for (int k = 0; k < 4; ++k) pt.win[k] = i32(p + 4 * k); }
            else if (name == "tiles" && size >= 9) { std::memcpy(&pt.tile_w, d.data() + p, 4); std::memcpy(&pt.tile_h, d.data() + p + 4, 4); pt.level_mode = d[p + 8] & 15; pt.round_up = d[p + 8] >> 4; }
            else if (name == "type" && size > 0) { const std::string t((const char *)d.data() + p, (size_t)size); pt.tiled = t == "tiledimage"; pt.deep = t.compare(0, 4, "deep") == 0; }
            else if (name == "chunkCount" && size == 4) pt.chunk_count = i32(p);
            p += (size_t)size;
        }
        if (!any) break;             // the empty header that ends a multi-part list
        parts.push_back(pt);
        if (!multipart) break;
    }
    if (parts.empty()) throw std::runtime_error("EXR \"" + path + "\": no header");
    // chunk counts of all parts (the offset tables follow the headers back to back)
    auto levels = [](int n, int round_up) { int l = 0; while (n > 1) { n = round_up ? (n + 1) / 2 : n / 2; ++l; } return l + 1; };
    auto level_size = [](int n, int l, int round_up) { for (int i = 0; i < l; ++i) n = std::max(1, round_up ? (n + 1) / 2 : n / 2); return n; };
    std::vector<size_t> n_chunks(parts.size());
    for (size_t k = 0; k < parts.size(); ++k) {
        const Part &pt = parts[k];
        const int64_t w64 = (int64_t)pt.win[2] - pt.win[0] + 1, h64 = (int64_t)pt.win[3] - pt.win[1] + 1;
        if (w64 <= 0 || h64 <= 0 || w64 > (1 << 26) || h64 > (1 << 26) || pt.chans.empty()) throw std::runtime_error("EXR \"" + path + "\": missing or absurd dataWindow / no channels");
        const int w = (int)w64, h = (int)h64;
        // a tiled part needs its tile description whether or not a chunkCount spares us the arithmetic (every part of a multi-part file carries one)
        if (pt.tiled && (pt.tile_w == 0 || pt.tile_h == 0 || pt.tile_w > (uint32_t)INT32_MAX || pt.tile_h > (uint32_t)INT32_MAX)) throw std::runtime_error("EXR \"" + path + "\": tiled part without a (sane) tile description");
        if (pt.chunk_count >= 0) { n_chunks[k] = (size_t)pt.chunk_count; continue; }
        if (!pt.tiled) { const int lpb = pt.compression == 3 ? 16 : (pt.compression == 4 || pt.compression == 6 ) ? 32 : (pt.compression == 5 || pt.compression == 7) ? 16 : 1; n_chunks[k] = (size_t)((h + lpb - 1) / lpb); continue; }
        auto tiles = [&](int lw, int lh) { return (size_t)((lw + (int)pt.tile_w - 1) / (int)pt.tile_w) * (size_t)((lh + (int)pt.tile_h - 1) / (int)pt.tile_h); };
        size_t n = 0;
        if (pt.level_mode == 0) n = tiles(w, h);
        else if (pt.level_mode == 1) { const int nl = levels(std::max(w, h), pt.round_up); for (int l = 0; l < nl; ++l) n += tiles(level_size(w, l, pt.round_up), level_size(h, l, pt.round_up)); }
        else { const int nx = levels(w, pt.round_up), ny = levels(h, pt.round_up); for (int ly = 0; ly < ny; ++ly) for (int lx = 0; lx < nx; ++lx) n += tiles(level_size(w, lx, pt.round_up), level_size(h, ly, pt.round_up)); }
        n_chunks[k] = n;
    }
    for (const Part &q : parts) if (q.deep) throw std::runtime_error("EXR \"" + path + "\": a part says it is deep although the version field does not: refused");
    const size_t use = 0;   // the first part (all parts are flat here)
    size_t table = p;
    for (size_t k = 0; k < use; ++k) table += 8 * n_chunks[k];
    const Part &pt = parts[use];
    const std::vector<Chan> &chans = pt.chans; const int compression = pt.compression; const int32_t *win = pt.win;
    const int w = (int)((int64_t)win[2] - win[0] + 1), h = (int)((int64_t)win[3] - win[1] + 1);   // (checked above)
    if (compression < 0 || compression > 3) {
        static const char *names[] = {"none", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB"};
        throw std::runtime_error("EXR \"" + path + "\": only uncompressed, RLE, ZIPS and ZIP data are supported (this file: " + (compression >= 0 && compression < 10 ? std::string(names[compression]) : std::to_string(compression)) + ")");
    }
    size_t px_bytes = 0; std::vector<size_t> chan_bytes;
    for (auto &c : chans) { chan_bytes.push_back(c.type == 1 ? 2 : 4); px_bytes += chan_bytes.back(); }
    int idx[3] = {-1, -1, -1};
    for (size_t c = 0; c < chans.size(); ++c) { if (chans[c].name == "R") idx[0] = (int)c; else if (chans[c].name == "G") idx[1] = (int)c; else if (chans[c].name == "B") idx[2] = (int)c; else if (chans[c].name == "Y" && idx[0] < 0) idx[0] = idx[1] = idx[2] = (int)c; }
    if (idx[0] < 0 || idx[1] < 0 || idx[2] < 0) throw std::runtime_error("EXR \"" + path + "\": no R, G, B (or Y) channels");
    Image im; im.w = w; im.h = h; im.rgb.assign((size_t)w * h * 3, 0.0f);
    std::vector<unsigned char> raw, tmp;
    // one chunk's pixel data (bw x bh pixels, line by line, channel by channel) -> raw
    auto unpack = [&](size_t src, size_t packed, size_t want) {
        need(src, packed);
        raw.resize(want);
        if (compression == 0 || packed == want) { std::memcpy(raw.data(), d.data() + src, std::min(want, packed)); return; }
        tmp.resize(want);
        if (compression == 1) {   // RLE: a count n < 0 is followed by -n literal bytes, n >= 0 by one byte to repeat n + 1 times
            size_t i = src, o = 0; const size_t end = src + packed;
            while (i < end) {
                const int n = (signed char)d[i++];
                if (n < 0) { const size_t c = (size_t)(-n); if (i + c > end || o + c > want) throw std::runtime_error("EXR \"" + path + "\": corrupt RLE block"); std::memcpy(tmp.data() + o, d.data() + i, c); i += c; o += c; }
                else { const size_t c = (size_t)n + 1; if (i >= end || o + c > want) throw std::runtime_error("EXR \"" + path + "\": corrupt RLE block"); std::memset(tmp.data() + o, d[i++], c); o += c; }
            }
            if (o != want) throw std::runtime_error("EXR \"" + path + "\": corrupt RLE block");
        } else {
            uLongf n = (uLongf)want;
            if (uncompress(tmp.data(), &n, d.data() + src, (uLong)packed) != Z_OK || n != want) throw std::runtime_error("EXR \"" + path + "\": corrupt ZIP block");
        }
        for (size_t i = 1; i < want; ++i) tmp[i] = (unsigned char)(tmp[i - 1] + tmp[i] - 128);     // undo the byte predictor
        const size_t half = (want + 1) / 2;                                                           // undo the even/odd split
        for (size_t i = 0; i < want; ++i) raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
    };
    auto store = [&](int x0, int y0, int bw, int bh) {
        const size_t line_bytes = px_bytes * (size_t)bw;
        for (int l = 0; l < bh; ++l)
            for (int c = 0; c < 3; ++c) {
                size_t off = 0; for (int k = 0; k < idx[c]; ++k) off += chan_bytes[k] * (size_t)bw;
                const Chan &ch = chans[idx[c]]; const unsigned char *src = raw.data() + line_bytes * (size_t)l + off;
                for (int x = 0; x < bw; ++x) {
                    float v;
                    if (ch.type == 1) { uint16_t hv; std::memcpy(&hv, src + 2 * x, 2); v = half_to_float(hv); }
                    else if (ch.type == 2) std::memcpy(&v, src + 4 * x, 4);
                    else { uint32_t u; std::memcpy(&u, src + 4 * x, 4); v = (float)u; }
                    im.rgb[((size_t)(y0 + l) * w + (x0 + x)) * 3 + c] = v;
                }
            }
    };
    const size_t part_field = multipart ? 4 : 0;
    if (!pt.tiled) {
        const int lines_per_block = compression == 3 ? 16 : 1, n_blocks = (h + lines_per_block - 1) / lines_per_block;
        for (int b = 0; b < n_blocks; ++b) {
            need(table + 8 * (size_t)b, 8);
            uint64_t off; std::memcpy(&off, d.data() + table + 8 * (size_t)b, 8);
            if (multipart && i32((size_t)off) != (int32_t)use) throw std::runtime_error("EXR \"" + path + "\": chunk of another part in this part's table");
            const size_t q = (size_t)off + part_field;
            const int y0 = i32(q) - win[1]; const int32_t packed = i32(q + 4);
            if (y0 < 0 || y0 >= h || packed < 0) throw std::runtime_error("EXR \"" + path + "\": bad scan-line block");
            const int nl = std::min(lines_per_block, h - y0);
            unpack(q + 8, (size_t)packed, px_bytes * (size_t)w * (size_t)nl);
            store(0, y0, w, nl);
        }
    } else {   // level (0, 0) comes first in the table of every level mode
        const int tw = (int)pt.tile_w, th = (int)pt.tile_h, nx = (w + tw - 1) / tw, ny = (h + th - 1) / th;
        for (int t = 0; t < nx * ny; ++t) {
            need(table + 8 * (size_t)t, 8);
            uint64_t off; std::memcpy(&off, d.data() + table + 8 * (size_t)t, 8);
            if (multipart && i32((size_t)off) != (int32_t)use) throw std::runtime_error("EXR \"" + path + "\": chunk of another part in this part's table");
            const size_t q = (size_t)off + part_field;
            const int tx = i32(q), ty = i32(q + 4), lx = i32(q + 8), ly = i32(q + 12); const int32_t packed = i32(q + 16);
            if (lx != 0 || ly != 0 || tx < 0 || ty < 0 || tx >= nx || ty >= ny || packed < 0) throw std::runtime_error("EXR \"" + path + "\": bad tile");
            const int x0 = tx * tw, y0 = ty * th, bw = std::min(tw, w - x0), bh = std::min(th, h - y0);
            unpack(q + 20, (size_t)packed, px_bytes * (size_t)bw * (size_t)bh);
            store(x0, y0, bw, bh);
        }
    }
    return im;
}
// write_image_exr (imageio.rs:99-114, `write_rgb_f32_file`): three FLOAT channels, scan lines, no compression
inline void write_exr(const std::string &path, int w, int h, const float *rgb_top_first) {
    std::vector<unsigned char> o;
    auto put = [&](const void *q, size_t n) { o.insert(o.end(), (const unsigned char *)q, (const unsigned char *)q + n); };
    auto i32 = [&](int32_t v) { put(&v, 4); };
    auto str = [&](const char *t) { put(t, std::strlen(t) + 1); };
    auto attr = [&](const char *name, const char *type, int32_t size) { str(name); str(type); i32(size); };
    i32(20000630); i32(2);
    attr("channels", "chlist", 3 * 18 + 1);
    for (const char *c : {"B", "G", "R"}) { str(c); i32(2); i32(0); i32(1); i32(1); }
    o.push_back(0);
    attr("compression", "compression", 1); o.push_back(0);
    attr("dataWindow", "box2i", 16); i32(0); i32(0); i32(w - 1); i32(h - 1);
    attr("displayWindow", "box2i", 16); i32(0); i32(0); i32(w - 1); i32(h - 1);
    attr("lineOrder", "lineOrder", 1); o.push_back(0);
    const float one = 1.0f, zero = 0.0f;
    attr("pixelAspectRatio", "float", 4); put(&one, 4);
    attr("screenWindowCenter", "v2f", 8); put(&zero, 4); put(&zero, 4);
    attr("screenWindowWidth", "float", 4); put(&one, 4);
    o.push_back(0);
    const size_t line = 8 + (size_t)w * 12; uint64_t off = o.size() + 8 * (uint64_t)h;
    for (int y = 0; y < h; ++y) { put(&off, 8); off += line; }
    std::vector<float> plane((size_t)w);
    for (int y = 0; y < h; ++y) {
        i32(y); i32((int32_t)((size_t)w * 12));
        for (int c = 2; c >= 0; --c) { for (int x = 0; x < w; ++x) plane[x] = rgb_top_first[((size_t)y * w + x) * 3 + c]; put(plane.data(), (size_t)w * 4); }   // B, G, R
    }
    FILE *fp = std::fopen(path.c_str(), "wb");
    if (!fp || std::fwrite(o.data(), 1, o.size(), fp) != o.size()) { if (fp) std::fclose(fp); throw std::runtime_error("cannot write \"" + path + "\""); }
    std::fclose(fp);
}

// write_image_png_tga (imageio.rs:359-381): clamp(255 * gamma_correct(v) + 0.5, 0, 255) as u8 (pbrt.rs:210-216)
inline unsigned char to_byte(float v) {
    const float g = v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f;
    const float s = 255.0f * g + 0.5f;
    return (unsigned char)(s < 0.0f || s != s ? 0.0f : (s > 255.0f ? 255.0f : s));
}
inline void write_png(const std::string &path, int w, int h, const float *rgb_top_first) {
    std::vector<unsigned char> lines(((size_t)w * 3 + 1) * h);
    for (int y = 0; y < h; ++y) { unsigned char *row = lines.data() + ((size_t)w * 3 + 1) * y; row[0] = 0; for (int i = 0; i < w * 3; ++i) row[1 + i] = to_byte(rgb_top_first[(size_t)y * w * 3 + i]); }
    uLongf zn = compressBound((uLong)lines.size()); std::vector<unsigned char> z(zn);
    if (compress2(z.data(), &zn, lines.data(), (uLong)lines.size(), 6) != Z_OK) throw std::runtime_error("cannot compress \"" + path + "\"");
    std::vector<unsigned char> o = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    auto be32 = [&](uint32_t v) { for (int s = 24; s >= 0; s -= 8) o.push_back((unsigned char)(v >> s)); };
    auto chunk = [&](const char *type, const unsigned char *body, size_t n) {
        be32((uint32_t)n); const size_t start = o.size(); o.insert(o.end(), type, type + 4); o.insert(o.end(), body, body + n);
        be32((uint32_t)crc32(0L, o.data() + start, (uInt)(n + 4)));
    };
    unsigned char ihdr[13] = {(unsigned char)(w >> 24), (unsigned char)(w >> 16), (unsigned char)(w >> 8), (unsigned char)w, (unsigned char)(h >> 24), (unsigned char)(h >> 16), (unsigned char)(h >> 8), (unsigned char)h, 8, 2, 0, 0, 0};
    chunk("IHDR", ihdr, 13); chunk("IDAT", z.data(), zn); chunk("IEND", nullptr, 0);
    FILE *fp = std::fopen(path.c_str(), "wb");
    if (!fp || std::fwrite(o.data(), 1, o.size(), fp) != o.size()) { if (fp) std::fclose(fp); throw std::runtime_error("cannot write \"" + path + "\""); }
    std::fclose(fp);
}
inline void write_tga(const std::string &path, int w, int h, const float *rgb_top_first) {
    std::vector<unsigned char> o = {0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, (unsigned char)w, (unsigned char)(w >> 8), (unsigned char)h, (unsigned char)(h >> 8), 24, 0x20};   // top-left origin
    for (size_t i = 0; i < (size_t)w * h; ++i) for (int c = 2; c >= 0; --c) o.push_back(to_byte(rgb_top_first[i * 3 + c]));
    FILE *fp = std::fopen(path.c_str(), "wb");
    if (!fp || std::fwrite(o.data(), 1, o.size(), fp) != o.size()) { if (fp) std::fclose(fp); throw std::runtime_error("cannot write \"" + path + "\""); }
    std::fclose(fp);
}
// write_image (imageio.rs:42-60): by extension
inline void write_image(const std::string &path, int w, int h, const float *rgb_top_first) {
    const size_t dot = path.find_last_of('.');
    const std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
    if (ext == "png") write_png(path, w, h, rgb_top_first);
    else if (ext == "tga") write_tga(path, w, h, rgb_top_first);
    else if (ext == "exr") write_exr(path, w, h, rgb_top_first);
    else if (ext == "pfm") write_pfm(path, w, h, rgb_top_first);
    else throw std::runtime_error("Unsupported file format \"" + ext + "\"");
}

// core/imageio.rs:18-40: dispatch on the file name extension
inline Image read_image(const std::string &path) {
    const size_t dot = path.find_last_of('.');
    const std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
    if (ext == "pfm") return read_pfm(path);
    if (ext == "hdr") return read_hdr(path);
    if (ext == "png" || ext == "PNG") return read_png(path);
    if (ext == "tga" || ext == "TGA") return read_tga(path);
    if (ext == "exr" || ext == "EXR") return read_exr(path);
    throw std::runtime_error("\"" + path + "\": unable to load image with this extension (imageio.rs:33-37)");
}

}  // namespace fe
