// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Restates src/renderer.rs of the reference: Renderer::new (:41-54), draw (:56-123),
// float_to_rgb (:135-137), cal_norm_coords (:174-184), ray_color (:197-237).
#pragma once
#include <atomic>
#include <cmath>
#include <thread>
#include <vector>
#include "scene.hpp"

namespace orc {

struct Renderer {
    uint32_t width, height;
    Camera camera;
    Scene scene;
    size_t multisample = 4;  // renderer.rs:50
    size_t max_depth = 10;   // renderer.rs:51
    bool dirty = true;
    // Oracle-only controls
    uint32_t seed = 0;        // Philox counter word 3
    uint32_t spp_begin = 0;   // global index of the first sample (for sample-range splits)
    uint32_t spp_total = 0;   // samples per pixel of the whole render when this call renders a range of them (0: multisample)
    int threads = 1;
    Counters totals;

    Renderer(uint32_t w, uint32_t h, Scene s, Camera c) : width(w), height(h), camera(std::move(c)), scene(std::move(s)) {}

    // renderer.rs:135-137
    static uint8_t float_to_rgb(float num) {
        float c = std::fmin(std::fmax(num, 0.f), 0.999f);
        return (uint8_t)(std::sqrt(c) * 256.f);
    }

    // renderer.rs:174-184 — divides by (width-1)/(height-1); no jitter when multisample == 1.
    void cal_norm_coords(uint32_t x, uint32_t y, float out[2]) const {
        float x_offset = 0.f, y_offset = 0.f;
        if ((spp_total ? spp_total : multisample) != 1) {  // a range of a longer render jitters as the whole does
            x_offset = get_rand();
            y_offset = get_rand();
        }
        out[0] = ((float)x + x_offset) / (float)(width - 1);
        out[1] = ((float)y + y_offset) / (float)(height - 1);
    }

    // renderer.rs:197-237. `bounce` (0 for the camera ray) only positions the RNG.
    Color ray_color(const Ray& r, size_t depth, uint32_t pixel, uint32_t sample, uint32_t bounce) const {
        if (depth == 0) return V3(0, 0, 0);
        RngCtx& g = rng();
        g.begin_bounce(pixel, sample, bounce, seed);
        g.stream = STREAM_SCATTER;
        ++counters().segments;
        auto hit = scene.world->hit(r, 0.001f, INF);
        if (!hit) return scene.background;
        const HitRecord& rec = *hit;
        Color emitted = rec.material->emit(rec).value_or(V3(0, 0, 0));
        Color scattered_c(0, 0, 0);
        if (auto sr = rec.material->scatter(r, rec)) {
            Ray scattered;
            float pdf_val;
            if (sr->kind == ScatterKind::Diffuse) {
                std::unique_ptr<PDF> mixed;
                if (scene.lights->objects.empty()) {
                    mixed = std::move(sr->pdf);
                } else {
                    auto hp = std::make_unique<HittablePDF>(rec.point, scene.lights);
                    mixed = std::make_unique<MixPDF>(std::move(sr->pdf), std::move(hp));
                }
                scattered = Ray(rec.point, mixed->generate(), r.time);
                float cosine = std::fmax(dot(scattered.direction, rec.normal), 0.0001f);
                float pv = mixed->value(scattered.direction);
                pdf_val = pv / cosine * PI * 2.f;
            } else {
                scattered = sr->ray;
                pdf_val = 1.f;
            }
            Color rec_c = ray_color(scattered, depth - 1, pixel, sample, bounce + 1);
            scattered_c = cmul(sr->attenuation, rec_c) / pdf_val;
        }
        return emitted + scattered_c;
    }

    // One iteration of renderer.rs:100-108: a primary path, sanitised per channel
    // (NaN -> 0, +-inf -> 1).
    Color sample_pixel(uint32_t x, uint32_t y, uint32_t sample) const {
        uint32_t pixel = y * width + x;
        RngCtx& g = rng();
        g.begin_bounce(pixel, sample, 0, seed);
        g.stream = STREAM_CAMERA;
        float uv[2];
        cal_norm_coords(x, y, uv);
        Ray ray = camera.get_ray(uv[0], uv[1]);
        Color c = ray_color(ray, max_depth, pixel, sample, 0);
        for (int i = 0; i < 3; ++i) {
            float v = c[i];
            c[i] = std::isnan(v) ? 0.f : (std::isinf(v) ? 1.f : v);
        }
        return c;
    }

    // Sum of sanitised samples per pixel, `sums[(y*width + x)*3 + c]` with y = 0 at the image
    // BOTTOM (the reference's y, renderer.rs:97). Rows are handed out dynamically to worker
    // threads, as rayon does with par_chunks_exact_mut (renderer.rs:72).
    void render_sums(float* sums) {
        std::atomic<uint32_t> next_row{0};
        std::vector<Counters> per_thread((size_t)threads);
        auto worker = [&](int tid) {
            counters() = Counters();
            for (;;) {
                uint32_t y = next_row.fetch_add(1);
                if (y >= height) break;
                for (uint32_t x = 0; x < width; ++x) {
                    Color acc(0, 0, 0);
                    for (size_t s = 0; s < multisample; ++s) acc = acc + sample_pixel(x, y, spp_begin + (uint32_t)s);
                    float* o = sums + ((size_t)y * width + x) * 3;
                    o[0] = acc.x; o[1] = acc.y; o[2] = acc.z;
                }
            }
            per_thread[(size_t)tid] = counters();
        };
        if (threads <= 1) {
            worker(0);
        } else {
            std::vector<std::thread> pool;
            for (int t = 0; t < threads; ++t) pool.emplace_back(worker, t);
            for (auto& t : pool) t.join();
        }
        totals = Counters();
        for (auto& c : per_thread) {
            totals.segments += c.segments; totals.sphere_tests += c.sphere_tests;
            totals.rect_tests += c.rect_tests; totals.aabb_tests += c.aabb_tests;
        }
    }

    // renderer.rs:56-123 — RGBA8 frame, top row first: renderer row y is frame row height-1-y.
    void draw(uint8_t* frame, size_t frame_len) {
        if (!dirty) return;
        dirty = false;
        if (frame_len / 4 != (size_t)width * height) throw std::runtime_error("frame size");
        std::vector<float> sums((size_t)width * height * 3);
        render_sums(sums.data());
        for (uint32_t y = 0; y < height; ++y)
            for (uint32_t x = 0; x < width; ++x) {
                const float* s = &sums[((size_t)y * width + x) * 3];
                uint8_t* px = frame + ((size_t)(height - 1 - y) * width + x) * 4;
                for (int c = 0; c < 3; ++c) px[c] = float_to_rgb(s[c] / (float)multisample);
                px[3] = 0xff;
            }
    }
};

}  // namespace orc
