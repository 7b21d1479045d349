// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Restates src/texture.rs:7-103 of the reference.
#pragma once
#include <algorithm>
#include <memory>
#include <vector>
#include "perlin.hpp"

namespace orc {

using Color = V3;

// texture.rs:7-9
struct Texture {
    virtual ~Texture() = default;
    virtual Color value(const float uv[2], V3 p) const = 0;
};
using SharedTexture = std::shared_ptr<Texture>;

// texture.rs:11-25
struct SolidColor : Texture {
    Color color;
    explicit SolidColor(Color c) : color(c) {}
    Color value(const float*, V3) const override { return color; }
};

// texture.rs:27-47 — constructor argument order is (even, odd).
struct CheckerTexture : Texture {
    SharedTexture odd, even;
    CheckerTexture(Color even_c, Color odd_c)
        : odd(std::make_shared<SolidColor>(odd_c)), even(std::make_shared<SolidColor>(even_c)) {}
    Color value(const float uv[2], V3 p) const override {
        float sines = 1.0f;
        for (int i = 0; i < 3; ++i) sines = sines * std::sin(10.f * p[i]);
        return (sines < 0.f ? odd : even)->value(uv, p);
    }
};

// texture.rs:49-69 — marble phase uses p.z and turbulence of the UNSCALED point.
struct NoiseTexture : Texture {
    Perlin noise;
    float scale;
    explicit NoiseTexture(float s) : noise(), scale(s) {}
    Color value(const float*, V3 p) const override {
        float g = 0.5f * (1.f + std::sin(scale * p.z + 10.f * noise.turb(p)));
        return V3(1.f * g, 1.f * g, 1.f * g);
    }
};

// texture.rs:71-103. The reference decodes static/earthmap.jpg with the `image` crate (0.24.2,
// not vendored) and flips it vertically (:84). The oracle is handed decoded RGBA8 texels, top row
// first as stored in the file, and applies the same flip here. A missing image yields (0,1,1).
struct ImageTexture : Texture {
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> rgba;  // after the vertical flip
    ImageTexture() = default;
    ImageTexture(const uint8_t* texels_top_first, uint32_t w_, uint32_t h_) : w(w_), h(h_) {
        if (texels_top_first && w && h) {
            rgba.resize((size_t)w * h * 4);
            for (uint32_t y = 0; y < h; ++y)
                std::copy(texels_top_first + (size_t)(h - 1 - y) * w * 4, texels_top_first + (size_t)(h - y) * w * 4,
                          rgba.begin() + (size_t)y * w * 4);
        }
    }
    Color value(const float uv[2], V3) const override {
        if (rgba.empty()) return V3(0.f, 1.f, 1.f);
        uint32_t dims[2] = {w, h}, xy[2];
        for (int i = 0; i < 2; ++i) {
            float c = std::min(std::max(uv[i], 0.f), 1.f);
            xy[i] = std::min((uint32_t)(c * (float)dims[i]), dims[i] - 1);
        }
        const uint8_t* px = &rgba[((size_t)xy[1] * w + xy[0]) * 4];
        const float s = 1.f / 255.f;
        return V3((float)px[0] * s, (float)px[1] * s, (float)px[2] * s);
    }
};

}  // namespace orc
