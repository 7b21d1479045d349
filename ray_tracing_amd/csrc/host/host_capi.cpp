// C entry points of the host-side scene / camera / renderer mirror (include/mirt_host.h).
#include <memory>
#include <string>
#include "../../../include/mirt_host.h"
#include "png_writer.hpp"
#include "renderer.hpp"

struct mirt_host_scene {
    rt::FlatScene flat;
    mirt_scene_desc desc;
};

namespace {
thread_local std::string g_host_err;
rt::SceneOptions options(uint32_t scene_seed, int random_num, const uint8_t* earth, uint32_t ew, uint32_t eh) {
    rt::SceneOptions o;
    o.scene_seed = scene_seed;
    o.random_num = random_num;
    o.earthmap = rt::EarthMap{earth, ew, eh};
    return o;
}
}  // namespace

extern "C" {

const char* mirt_host_last_error(void) { return g_host_err.c_str(); }

int mirt_host_select_scene(const char* name, uint32_t scene_seed, int random_num, const uint8_t* earth_rgba, uint32_t ew, uint32_t eh,
                           mirt_host_scene** out) {
    if (!name || !out) { g_host_err = "mirt_host_select_scene: NULL argument"; return MIRT_ERR_INVALID; }
    try {
        rt::Scene sc = rt::select_scene(name, options(scene_seed, random_num, earth_rgba, ew, eh));
        auto h = std::make_unique<mirt_host_scene>();
        h->flat = sc.flatten();
        h->desc = h->flat.desc();
        *out = h.release();
        return MIRT_OK;
    } catch (const std::exception& e) {
        g_host_err = e.what();
        return MIRT_ERR_INVALID;
    } catch (...) {
        g_host_err = "mirt_host_select_scene: unknown exception";
        return MIRT_ERR_INVALID;
    }
}
const mirt_scene_desc* mirt_host_scene_desc(const mirt_host_scene* s) { return s ? &s->desc : nullptr; }
void mirt_host_scene_free(mirt_host_scene* s) { delete s; }

int mirt_host_select_camera(float aspect_ratio, const char* scene, mirt_camera* out) {
    if (!scene || !out) { g_host_err = "mirt_host_select_camera: NULL argument"; return MIRT_ERR_INVALID; }
    try {
        *out = rt::Camera::select_camera(aspect_ratio, scene).c;
        return MIRT_OK;
    } catch (const std::exception& e) {
        g_host_err = e.what();
        return MIRT_ERR_INVALID;
    } catch (...) {
        g_host_err = "mirt_host_select_camera: unknown exception";
        return MIRT_ERR_INVALID;
    }
}

int mirt_host_draw(int device, const char* name, uint32_t width, uint32_t height, uint32_t multisample, uint32_t max_depth,
                   uint32_t seed, uint32_t scene_seed, int random_num, const uint8_t* earth_rgba, uint32_t ew, uint32_t eh,
                   uint8_t* frame, size_t frame_len) {
    if (!name || !frame) { g_host_err = "mirt_host_draw: NULL argument"; return MIRT_ERR_INVALID; }
    try {
        rt::Scene sc = rt::select_scene(name, options(scene_seed, random_num, earth_rgba, ew, eh));
        rt::Camera cam = rt::Camera::select_camera((float)width / (float)height, name);  // cli.rs:25
        rt::Renderer r(width, height, sc, cam, device);
        r.multisample = multisample;
        r.max_depth = max_depth;
        r.seed = seed;
        r.draw(frame, frame_len);
        return MIRT_OK;
    } catch (const std::exception& e) {
        g_host_err = e.what();
        return MIRT_ERR_INVALID;
    } catch (...) {
        g_host_err = "mirt_host_draw: unknown exception";
        return MIRT_ERR_INVALID;
    }
}

int mirt_host_save_png(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height) {
    if (!path || !rgba) { g_host_err = "mirt_host_save_png: NULL argument"; return MIRT_ERR_INVALID; }
    try {
        rt::png_save_rgba8(path, rgba, width, height);
        return MIRT_OK;
    } catch (const std::exception& e) {
        g_host_err = e.what();
        return MIRT_ERR_INVALID;
    } catch (...) {
        g_host_err = "mirt_host_save_png: unknown exception";
        return MIRT_ERR_INVALID;
    }
}

}  // extern "C"
