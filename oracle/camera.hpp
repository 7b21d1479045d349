// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Restates src/camera.rs:5-139 of the reference.
#pragma once
#include <cmath>
#include <string>
#include "geo.hpp"

namespace orc {

struct Camera {
    V3 origin, horizontal, vertical, lower_left_corner;
    float aspect_ratio, vfov;
    V3 w, u, v, vup;
    float len_radius, focus_dist, time0, time1;
    // Oracle-only switch for the deterministic first-hit check: lens offset forced to 0 and the
    // shutter time forced to `fixed_time`, with no draws consumed.
    bool fixed = false;
    float fixed_time = 0.f;

    // camera.rs:29-69 — u = vup x w and v = w x u are not normalised.
    Camera(V3 lookfrom, V3 direction, V3 vup_, float vfov_, float aspect, float aperture, float focus, float t0, float t1) {
        float theta = degree_to_radian(vfov_);
        float h = std::tan(theta / 2.f);
        float viewport_height = 2.f * h;
        float viewport_width = viewport_height * aspect;
        w = -direction;
        u = cross(vup_, w);
        v = cross(w, u);
        horizontal = (focus * viewport_width) * u;
        vertical = (focus * viewport_height) * v;
        origin = lookfrom;
        lower_left_corner = ((origin - horizontal / 2.f) - vertical / 2.f) - w * focus;
        vfov = vfov_; vup = vup_; len_radius = aperture / 2.f; focus_dist = focus;
        time0 = t0; time1 = t1; aspect_ratio = aspect;
    }

    // camera.rs:86-94 — the disk rejection loop runs even when the aperture is 0.
    Ray get_ray(float s, float t) const {
        if (fixed) {
            return Ray(origin, normalize(((lower_left_corner + s * horizontal) + t * vertical) - origin), fixed_time);
        }
        V3 rd = len_radius * rand_vec3_in_unit_disk();
        V3 offset = u * rd.x + v * rd.y;
        V3 dir = normalize((((lower_left_corner + s * horizontal) + t * vertical) - origin) - offset);
        float time = get_rand_range(time0, time1);
        return Ray(origin + offset, dir, time);
    }

    // camera.rs:96-139
    static Camera select_camera(float aspect_ratio, const std::string& scene) {
        float aperture = 0.1f;
        const float dist_to_focus = 10.f;
        float vfov = 20.f;
        V3 vup(0, 1, 0);
        V3 lookfrom = V3(13.f, 1.5f, 3.f) * 2.f;
        V3 lookat(0.f, 2.f, 0.f);
        V3 direction = lookat - lookfrom;
        if (scene == "cornell" || scene == "smoke") {
            lookfrom = V3(278.f, 278.f, -800.f);
            lookat = V3(278.f, 278.f, 0.f);
            direction = lookat - lookfrom;
            vfov = 40.0f;
            aperture = 0.f;
        } else if (scene == "simplelight") {
        } else if (scene == "final") {
            lookfrom = V3(478.f, 278.f, -600.f);
            lookat = V3(278.f, 278.f, 0.f);
            direction = lookat - lookfrom;
            vfov = 40.f;
        } else {
            lookfrom = V3(13.f, 2.f, 3.f);
            direction = V3(-13.f, -2.f, -3.f);
        }
        return Camera(lookfrom, normalize(direction), vup, vfov, aspect_ratio, aperture, dist_to_focus, 0.f, 1.f);
    }
};

}  // namespace orc
