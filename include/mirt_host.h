/*
 * mirt_host.h — C entry points of the host-side mirror of the reference's Scene / Camera API.
 *
 * The reference builds its scenes in Rust (src/scene.rs:11-275, src/camera.rs:96-139) and would
 * flatten them itself before calling mirt_scene_upload(). Where no Rust toolchain exists, the same
 * builders are provided in C++ (ray_tracing_amd/csrc/host/) and exposed here so that tests, the
 * benchmark and a command line can obtain the flattened scene of a named preset.
 */
#ifndef MIRT_HOST_H
#define MIRT_HOST_H
#include "mirt.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mirt_host_scene mirt_host_scene;

/* select_scene(name) (scene.rs:11-23), flattened. scene_seed seeds the builders' random draws
 * (unseeded in the reference); random_num is `num` of scene.rs:47 (11 there). earth_rgba is the
 * decoded static/earthmap.jpg, RGBA8, top row first, or NULL for the "file missing" colour. */
int mirt_host_select_scene(const char* name, uint32_t scene_seed, int random_num, const uint8_t* earth_rgba,
                           uint32_t earth_width, uint32_t earth_height, mirt_host_scene** out);
const mirt_scene_desc* mirt_host_scene_desc(const mirt_host_scene* s);
void mirt_host_scene_free(mirt_host_scene* s);

/* Camera::select_camera(aspect_ratio, scene) (camera.rs:96-139). */
int mirt_host_select_camera(float aspect_ratio, const char* scene, mirt_camera* out);

/* Renderer::new + multisample / max_depth + draw (renderer.rs:41-123, cli.rs:25-30) in one call:
 * renders scene `name` into the host RGBA8 frame (top row first). */
int mirt_host_draw(int device, const char* name, uint32_t width, uint32_t height, uint32_t multisample, uint32_t max_depth,
                   uint32_t seed, uint32_t scene_seed, int random_num, const uint8_t* earth_rgba, uint32_t earth_width,
                   uint32_t earth_height, uint8_t* frame, size_t frame_len);

/* image::save_buffer(path, &pixels, width, height, ColorType::Rgba8) of the image mode (cli.rs:31-38):
 * writes the RGBA8 frame (top row first) as a PNG file. */
int mirt_host_save_png(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height);

const char* mirt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
