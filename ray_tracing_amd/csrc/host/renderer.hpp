// Host-side mirror of the reference's Renderer (src/renderer.rs:24-123): same constructor
// arguments, same public knobs (multisample, max_depth, dirty), same draw(frame) contract — the
// body of draw() is the device render + resolve instead of the rayon row loop.
#pragma once
#include <stdexcept>
#include <string>
#include "scenes.hpp"

namespace rt {

class Renderer {
   public:
    uint32_t width, height;
    size_t multisample = 4;  // renderer.rs:50
    size_t max_depth = 10;   // renderer.rs:51
    bool dirty = true;       // renderer.rs:52
    uint32_t seed = 0;
    int device = 0;

    // renderer.rs:41-54
    Renderer(uint32_t w, uint32_t h, const Scene& scene, const Camera& camera, int device_ = 0)
        : width(w), height(h), device(device_), flat_(scene.flatten()), camera_(camera) {}
    ~Renderer() { if (ctx_) mirt_destroy(ctx_); }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    // renderer.rs:192-195 (scene / camera swap)
    void set_scene(const Scene& scene, const Camera& camera) {
        flat_ = scene.flatten();
        camera_ = camera;
        uploaded_ = false;
        dirty = true;
    }

    // renderer.rs:56-123: no-op unless dirty; frame must hold width*height*4 bytes (renderer.rs:65);
    // blocks until the RGBA8 frame (top row first) is complete.
    void draw(uint8_t* frame, size_t frame_len) {
        if (!dirty) return;
        dirty = false;
        if (frame_len / 4 != (size_t)width * height || frame_len % 4) throw std::invalid_argument("frame.len()/4 != width*height");
        if (!ctx_) check(mirt_create(device, &ctx_), nullptr);
        if (!uploaded_) {
            mirt_scene_desc d = flat_.desc();
            check(mirt_scene_upload(ctx_, &d), ctx_);
            uploaded_ = true;
        }
        check(mirt_camera_set(ctx_, &camera_.c), ctx_);
        check(mirt_frame_begin(ctx_, width, height, nullptr), ctx_);
        mirt_render_params p{};
        p.width = width; p.height = height;
        p.spp_begin = 0; p.spp_count = (uint32_t)multisample;
        p.max_depth = (uint32_t)max_depth; p.seed = seed;
        p.jitter = multisample != 1 ? 1u : 0u;  // renderer.rs:175
        p.kernel = MIRT_KERNEL_AUTO;
        check(mirt_render(ctx_, &p), ctx_);
        check(mirt_resolve_rgba8(ctx_, frame, frame_len), ctx_);
    }
    mirt_ctx* context() { return ctx_; }

   private:
    static void check(int rc, mirt_ctx* c) {
        if (rc != MIRT_OK) throw std::runtime_error(std::string("mirt: ") + mirt_last_error(c));
    }
    FlatScene flat_;
    Camera camera_;
    mirt_ctx* ctx_ = nullptr;
    bool uploaded_ = false;
};

}  // namespace rt
