/* mi355front.h -- C ABI of the .pbrt scene-file front end (libmi355front.so).
 *
 * Replaces, on the host side, what `pbrt_parse` + `API` do in the reference before `integ.render(&scene)`
 * (pbrtparser/pbrtparser.rs:26-87, core/api.rs): it turns a .pbrt file into the PtSceneDesc / PtRenderParams that
 * libmi355pt.so consumes. Pure host C++ (no GPU code, no dependency on libmi355pt.so): a Rust host would keep its own
 * parser and only use include/mi355pt.h; this library is for hosts that have none (the `mi355pbrt` command line tool).
 * Supported subset and the directives that are rejected: see pbrt-rust_amd/frontend/frontend.cpp header. */
#ifndef MI355FRONT_H
#define MI355FRONT_H
#include <stddef.h>
#include "mi355pt.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ptf_scene ptf_scene;
/* Parse a scene file / an in-memory scene (file names resolve against base_dir). PT_OK or PT_ERR_INVALID_ARG + ptf_last_error(). */
int ptf_parse_file(const char *path, ptf_scene **out);
int ptf_parse_string(const char *text, const char *base_dir, ptf_scene **out);
const char *ptf_last_error(void);
/* Views into the parsed scene; valid until ptf_scene_destroy. */
const PtSceneDesc *ptf_scene_desc(const ptf_scene *scene);
const PtRenderParams *ptf_render_params(const ptf_scene *scene);
const char *ptf_output_filename(const ptf_scene *scene);   /* Film "string filename" */
void ptf_scene_destroy(ptf_scene *scene);
/* Film::write_image's PFM branch (core/imageio.rs:288-328): rgb = width*height*3 floats, top row first. */
int ptf_write_pfm(const char *path, int width, int height, const float *rgb);
/* write_image (core/imageio.rs:42-60): by extension -- exr (three FLOAT channels, uncompressed), png / tga (8-bit, gamma
 * encoded, imageio.rs:359-381), pfm. */
int ptf_write_image(const char *path, int width, int height, const float *rgb);
/* read_image (core/imageio.rs:18-40): pfm, hdr, png, tga, exr (scan-line / tiled / multi-part, NO/RLE/ZIPS/ZIP). Call with rgb == NULL to get the
 * size, then with a buffer of width*height*3 floats (top row first). */
int ptf_read_image(const char *path, int *width, int *height, float *rgb, size_t capacity_floats);
#ifdef __cplusplus
}
#endif
#endif
