// mi355pbrt -- command-line renderer: the drop-in for `pbrt-rust scene.pbrt` on this back end (main.rs + api.rs:1715-1748).
//   mi355pbrt scene.pbrt [--outfile out.pfm] [--device N] [--spp N] [--quiet]
// Parses with libmi355front.so, renders with libmi355pt.so (HIP), writes the film in the format the Film's "filename"
// extension names (exr -- the reference's default "pbrt.exr" -- png, tga, pfm: core/imageio.rs:42-60).
#include "../../include/mi355front.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char **argv) {
    std::string scene, outfile; int device = 0, spp = 0; bool quiet = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--outfile" && i + 1 < argc) outfile = argv[++i];
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--spp" && i + 1 < argc) spp = std::atoi(argv[++i]);
        else if (a == "--quiet") quiet = true;
        else if (a[0] != '-') scene = a;
        else { std::fprintf(stderr, "usage: mi355pbrt scene.pbrt [--outfile out.pfm] [--device N] [--spp N] [--quiet]\n"); return 2; }
    }
    if (scene.empty()) { std::fprintf(stderr, "usage: mi355pbrt scene.pbrt [--outfile out.pfm] [--device N] [--spp N] [--quiet]\n"); return 2; }
    ptf_scene *fs = nullptr;
    if (ptf_parse_file(scene.c_str(), &fs) != PT_OK) { std::fprintf(stderr, "mi355pbrt: %s\n", ptf_last_error()); return 1; }
    PtRenderParams rp = *ptf_render_params(fs);
    if (spp > 0) rp.spp = (uint32_t)spp;
    if (outfile.empty()) {
        outfile = ptf_output_filename(fs);
    }
    if (pt_init(device) != PT_OK) { std::fprintf(stderr, "mi355pbrt: %s\n", pt_last_error()); return 1; }
    const auto t0 = std::chrono::steady_clock::now();
    pt_scene *sc = nullptr;
    if (pt_scene_create(ptf_scene_desc(fs), &sc) != PT_OK) { std::fprintf(stderr, "mi355pbrt: %s\n", pt_last_error()); return 1; }
    const auto t1 = std::chrono::steady_clock::now();
    const int w = rp.cropped_pixel_bounds[2] - rp.cropped_pixel_bounds[0], h = rp.cropped_pixel_bounds[3] - rp.cropped_pixel_bounds[1];
    std::vector<float> film((size_t)w * h * 4, 0.0f), rgb((size_t)w * h * 3);
    if (pt_render(sc, &rp, film.data(), 0) != PT_OK) { std::fprintf(stderr, "mi355pbrt: %s\n", pt_last_error()); return 1; }
    const auto t2 = std::chrono::steady_clock::now();
    pt_film_resolve(film.data(), (uint32_t)(w * h), rp.scale, rgb.data());
    if (ptf_write_image(outfile.c_str(), w, h, rgb.data()) != PT_OK) { std::fprintf(stderr, "mi355pbrt: %s\n", ptf_last_error()); return 1; }
    if (!quiet) {
        PtCounters c; pt_get_counters(sc, &c);
        const double ts = std::chrono::duration<double>(t1 - t0).count(), tr = std::chrono::duration<double>(t2 - t1).count();
        std::printf("%s: %dx%d, %u spp, %llu camera rays, scene %.2f s, render %.3f s (%.1f Msamples/s) -> %s\n", scene.c_str(), w, h, rp.spp,
                    (unsigned long long)c.camera_rays, ts, tr, (double)c.camera_rays / tr / 1e6, outfile.c_str());
    }
    pt_scene_destroy(sc); ptf_scene_destroy(fs);
    return 0;
}
