/* A plain C99 client of libmi355pt.so: the call sequence of INTEGRATION.md section 2 (pt_init, pt_scene_create, pt_render into a host film, pt_get_counters,
 * pt_film_resolve, pt_scene_destroy) with structs laid out by a C compiler from include/mi355pt.h -- what a bindgen-generated Rust module sees.
 * The scene's arrays come from files the test wrote (tests/test_c_client.py): argv[1] = directory. Prints the counters; writes film.bin / rgb.bin there. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mi355pt.h"

static void *slurp(const char *dir, const char *name, size_t *n_bytes) {
    char path[1024]; snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "rb"); if (!f) { if (n_bytes) *n_bytes = 0; return NULL; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    void *p = malloc(n > 0 ? (size_t)n : 1); if (n > 0 && fread(p, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(p); return NULL; }
    fclose(f); if (n_bytes) *n_bytes = (size_t)n; return p;
}
#define CHECK(call) do { int st_ = (call); if (st_ != PT_OK) { fprintf(stderr, "%s -> status %d: %s\n", #call, st_, pt_last_error()); return 10 + st_; } } while (0)

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const char *dir = argv[1]; size_t n;
    PtSceneDesc d; memset(&d, 0, sizeof d);
    d.P = (const float *)slurp(dir, "P.bin", &n); d.n_vertices = (uint32_t)(n / 12);
    d.indices = (const uint32_t *)slurp(dir, "indices.bin", &n); d.n_triangles = (uint32_t)(n / 12);
    d.tri_flags = (const uint8_t *)slurp(dir, "tri_flags.bin", &n);
    d.prim_shape = (const uint32_t *)slurp(dir, "prim_shape.bin", &n); d.n_prims = (uint32_t)(n / 4);
    d.prim_material = (const uint32_t *)slurp(dir, "prim_material.bin", &n);
    d.prim_light = (const uint32_t *)slurp(dir, "prim_light.bin", &n);
    d.materials = (const PtMaterial *)slurp(dir, "materials.bin", &n); d.n_materials = (uint32_t)(n / sizeof(PtMaterial));
    d.lights = (const PtLight *)slurp(dir, "lights.bin", &n); d.n_lights = (uint32_t)(n / sizeof(PtLight));
    d.max_node_prims = 4;
    PtRenderParams *rp = (PtRenderParams *)slurp(dir, "render_params.bin", &n);
    if (!d.P || !d.indices || !rp || n != sizeof(PtRenderParams)) { fprintf(stderr, "bad input (sizeof(PtRenderParams) = %zu, file = %zu)\n", sizeof(PtRenderParams), n); return 3; }
    CHECK(pt_init(0));
    pt_scene *scene = NULL;
    CHECK(pt_scene_create(&d, &scene));
    const uint32_t w = (uint32_t)(rp->cropped_pixel_bounds[2] - rp->cropped_pixel_bounds[0]), h = (uint32_t)(rp->cropped_pixel_bounds[3] - rp->cropped_pixel_bounds[1]);
    float *film = (float *)calloc((size_t)w * h * 4, 4), *rgb = (float *)calloc((size_t)w * h * 3, 4);
    CHECK(pt_render(scene, rp, film, 0));
    PtCounters c;
    CHECK(pt_get_counters(scene, &c));
    CHECK(pt_film_resolve(film, w * h, rp->scale, rgb));
    pt_scene_destroy(scene);
    printf("camera_rays %llu intersect_tests %llu shadow_tests %llu triangle_tests %llu film_splats %llu\n", (unsigned long long)c.camera_rays,
           (unsigned long long)c.intersect_tests, (unsigned long long)c.shadow_tests, (unsigned long long)c.triangle_tests, (unsigned long long)c.film_splats);
    char path[1024]; FILE *f;
    snprintf(path, sizeof path, "%s/film.bin", dir); f = fopen(path, "wb"); fwrite(film, 16, (size_t)w * h, f); fclose(f);
    snprintf(path, sizeof path, "%s/rgb.bin", dir); f = fopen(path, "wb"); fwrite(rgb, 12, (size_t)w * h, f); fclose(f);
    return 0;
}
