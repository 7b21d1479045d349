// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Restates src/rand_gen.rs:7-90 of the reference. The reference draws from rand::random::<f32>()
// (rand 0.8.5, not vendored: thread-local ChaCha12, top 24 bits * 2^-24 in [0,1)). That generator is
// unseeded and cannot be reproduced; what the rest of the renderer depends on is only the
// distribution of get_rand() and the ORDER of draws inside each sampler, both kept here.
// The source of uniform bits is Philox4x32-10 (Salmon et al. 2011, Random123), addressed as
//   key = (pixel index, global sample index)
//   ctr = (block, bounce, stream, seed)
// where a "stream" is a sequential run of draws: STREAM_CAMERA (pixel jitter, lens disk, shutter
// time), STREAM_SCATTER (everything Material::scatter and PDF::generate draw at one bounce) and
// STREAM_MEDIUM + m (the single free-path draw of constant medium m at one bounce). Keeping the
// medium draw on its own stream makes the result independent of the order objects are visited in.
#pragma once
#include <cstdint>
#include <vector>
#include "vec.hpp"

namespace orc {

inline void philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

inline float u32_to_unit_f32(uint32_t v) { return (float)(v >> 8) * (1.0f / 16777216.0f); }

enum : uint32_t { STREAM_CAMERA = 0, STREAM_SCATTER = 1, STREAM_MEDIUM = 2 };

// One sequential stream: draw i is word (i & 3) of Philox block (i >> 2).
struct RngStream {
    uint32_t key[2] = {0, 0};
    uint32_t ctr123[3] = {0, 0, 0};
    uint32_t next = 0;
    uint32_t buf[4];
    uint32_t buf_block = 0xffffffffu;
    float draw() {
        uint32_t block = next >> 2;
        if (block != buf_block) {
            uint32_t ctr[4] = {block, ctr123[0], ctr123[1], ctr123[2]};
            philox4x32_10(key, ctr, buf);
            buf_block = block;
        }
        return u32_to_unit_f32(buf[next++ & 3]);
    }
};

// Per-thread draw context. The renderer positions it (pixel, sample, bounce); samplers call
// get_rand() exactly where the reference calls it.
struct RngCtx {
    uint32_t pixel = 0, sample = 0, bounce = 0, seed = 0;
    uint32_t stream = STREAM_CAMERA;
    std::vector<RngStream> streams;  // indexed by stream id, reset at every bounce
    uint64_t draws = 0;

    void begin_bounce(uint32_t pixel_, uint32_t sample_, uint32_t bounce_, uint32_t seed_) {
        pixel = pixel_; sample = sample_; bounce = bounce_; seed = seed_;
        for (auto& s : streams) { s.next = 0; s.buf_block = 0xffffffffu; }
    }
    RngStream& get(uint32_t id) {
        if (streams.size() <= id) streams.resize(id + 1);
        RngStream& s = streams[id];
        s.key[0] = pixel; s.key[1] = sample;
        s.ctr123[0] = bounce; s.ctr123[1] = id; s.ctr123[2] = seed;
        return s;
    }
    float draw() { ++draws; return get(stream).draw(); }
};

inline RngCtx& rng() {
    static thread_local RngCtx ctx;
    return ctx;
}

// RAII switch of the active stream (used by ConstantMedium::hit).
struct StreamScope {
    uint32_t saved;
    explicit StreamScope(uint32_t id) : saved(rng().stream) { rng().stream = id; }
    ~StreamScope() { rng().stream = saved; }
};

// rand_gen.rs:7-17
inline float get_rand() { return rng().draw(); }
// rand_gen.rs:20-23
inline float get_rand_range(float min, float max) { return min + (max - min) * get_rand(); }
// rand_gen.rs:26-28 (Rust `as i32` truncates toward zero)
inline int get_rand_int_range(int min, int max) { return (int)get_rand_range((float)min, (float)max); }
// rand_gen.rs:31-33
inline size_t get_rand_usize_range(size_t min, size_t max) {
    float v = get_rand_range((float)min, (float)max);
    return v <= 0.f ? 0 : (size_t)v;  // Rust float->usize saturates at 0
}
// rand_gen.rs:36-42 (array literal: x, then y, then z)
inline V3 get_rand_vec3_range(float min, float max) {
    float x = get_rand_range(min, max);
    float y = get_rand_range(min, max);
    float z = get_rand_range(min, max);
    return V3(x, y, z);
}
// rand_gen.rs:45-52
inline V3 rand_vec3_in_unit_sphere() {
    for (;;) {
        V3 p = get_rand_vec3_range(-1.f, 1.f);
        if (norm_squared(p) < 1.f) return p;
    }
}
// rand_gen.rs:55-57
inline V3 rand_vec3_on_unit_sphere() { return normalize(rand_vec3_in_unit_sphere()); }
// rand_gen.rs:70-78
inline V3 rand_vec3_in_unit_disk() {
    for (;;) {
        float x = get_rand_range(-1.f, 1.f);
        float y = get_rand_range(-1.f, 1.f);
        V3 p(x, y, 0.f);
        if (norm_squared(p) >= 1.f) continue;
        return p;
    }
}
// rand_gen.rs:80-90 — r2 is drawn from [0, 0.9): the lobe is truncated at z >= sqrt(0.1).
inline V3 random_cosine_direction() {
    float r1 = get_rand();
    float r2 = get_rand_range(0.f, 0.9f);
    float z = std::sqrt(1.f - r2);
    float phi = 2.f * PI * r1;
    float r2_sq = std::sqrt(r2);
    float x = std::cos(phi) * r2_sq;
    float y = std::sin(phi) * r2_sq;
    return V3(x, y, z);
}

}  // namespace orc
