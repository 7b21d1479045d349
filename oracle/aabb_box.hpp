// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Restates AxisAlignedBoundingBox of src/aabb.rs:8-61 of the reference. Included from ray.hpp.
#pragma once
#include <algorithm>
#include <optional>

namespace orc {

struct AABB {
    V3 minimum, maximum;
    AABB() = default;
    AABB(V3 mn, V3 mx) : minimum(mn), maximum(mx) {}

    // aabb.rs:18-41
    static std::optional<AABB> surrounding_box(const std::optional<AABB>& b0, const std::optional<AABB>& b1) {
        if (b0 && b1) {
            V3 small(std::min(b0->minimum.x, b1->minimum.x), std::min(b0->minimum.y, b1->minimum.y),
                     std::min(b0->minimum.z, b1->minimum.z));
            V3 big(std::max(b0->maximum.x, b1->maximum.x), std::max(b0->maximum.y, b1->maximum.y),
                   std::max(b0->maximum.z, b1->maximum.z));
            return AABB(small, big);
        }
        if (b0) return b0;
        if (b1) return b1;
        return std::nullopt;
    }

    // aabb.rs:43-60 — each axis is tested against the ORIGINAL [t_min, t_max] (the narrowed
    // interval is shadowed, not carried to the next axis), so the test is looser than a true slab
    // test. It never rejects a box the ray really enters.
    bool hit(const Ray& ray, float t_min, float t_max) const {
        ++counters().aabb_tests;
        for (int i = 0; i < 3; ++i) {
            float a = ray.origin[i], b = ray.direction[i], mn = minimum[i], mx = maximum[i];
            float inv_b = 1.f / b;
            float t0 = (mn - a) * inv_b;
            float t1 = (mx - a) * inv_b;
            if (inv_b < 0.f) std::swap(t0, t1);
            // f32::max / f32::min return the non-NaN operand
            float lo = std::fmax(t_min, t0), hi = std::fmin(t_max, t1);
            if (!(hi > lo)) return false;
        }
        return true;
    }
};

}  // namespace orc
