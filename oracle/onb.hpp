// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Restates src/onb.rs:3-50 of the reference.
#pragma once
#include "vec.hpp"

namespace orc {

struct ONB {
    V3 u, v, w;  // columns of the reference's Rotation3 (onb.rs:45)

    // onb.rs:38-49. v = w x a and u = w x v are NOT normalised: |u| = |v| = sin(angle(w, a)),
    // so the basis is orthonormal only when w is perpendicular to a.
    static ONB build_from_w(V3 w) {
        V3 x_axis(1, 0, 0);
        V3 a = std::fabs(dot(w, x_axis)) > 0.9f ? V3(0, 1, 0) : x_axis;
        ONB o;
        o.v = cross(w, a);
        o.u = cross(w, o.v);
        o.w = w;
        return o;
    }
    // onb.rs:34-36 — matrix * vector, column-major accumulation, no renormalisation.
    V3 local_dir(V3 c) const { return (u * c.x + v * c.y) + w * c.z; }
};

}  // namespace orc
