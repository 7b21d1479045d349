// Host-side mirror of the reference's scene builders (src/scene.rs:11-275) and camera presets
// (src/camera.rs:96-139): same names, same constants, same order of random draws. The reference
// draws from an unseeded generator; here the draws come from BuildRng(scene_seed).
#pragma once
#include <string>
#include "scene_tree.hpp"

namespace rt {

struct EarthMap {  // decoded static/earthmap.jpg: RGBA8, top row first; null = file missing
    const uint8_t* rgba = nullptr;
    uint32_t width = 0, height = 0;
};

struct SceneOptions {
    uint32_t scene_seed = 1;
    int random_num = 11;  // scene.rs:47
    EarthMap earthmap;
};

namespace scenes {

inline SharedHittable rect(SharedMaterial m, float k, float a0, float a1, float b0, float b1, uint32_t axis) {
    return AxisAlignedRect::create(std::move(m), k, {a0, a1}, {b0, b1}, axis);
}

// scene.rs:76-103
inline SharedHittable create_random_sphere(int ai, int bi, BuildRng& g) {
    float a = (float)ai, b = (float)bi;
    float mat = g.get_rand();
    float cx = a + 0.9f * g.get_rand();
    float cz = b + 0.9f * g.get_rand();
    RGB center{cx, 0.2f, cz};
    float dx = center[0] - 4.f, dy = center[1] - 0.2f, dz = center[2] - 0.f;
    if (std::sqrt((dx * dx + dy * dy) + dz * dz) > 0.9f) {
        if (mat < 0.8f) {
            RGB a1 = g.get_rand_vec3_range(0.f, 1.f), a2 = g.get_rand_vec3_range(0.f, 1.f);
            RGB albedo{a1[0] * a2[0], a1[1] * a2[1], a1[2] * a2[2]};
            RGB center2{center[0] + 0.f, center[1] + g.get_rand_range(0.f, 0.5f), center[2] + 0.f};
            return Sphere::create_moving(center, center2, 0.f, 1.f, 0.2f, Lambertian::from_color(albedo));
        } else if (mat < 0.95f) {
            RGB albedo = g.get_rand_vec3_range(0.5f, 1.f);
            float fuzz = g.get_rand_range(0.f, 0.5f);
            return Sphere::create(center, 0.2f, Metal::create(albedo, fuzz));
        }
        return Sphere::create(center, 0.2f, Dielectric::create(1.5f));
    }
    return nullptr;
}

// scene.rs:46-74
inline Scene create_random_scene(const std::string& name, int num, BuildRng& g) {
    std::vector<SharedHittable> objects;
    for (int a = -num; a < num; ++a)
        for (int b = -num; b < num; ++b)
            if (auto s = create_random_sphere(a, b, g)) objects.push_back(s);
    auto ground = Lambertian::create(CheckerTexture::create({0.2f, 0.3f, 0.1f}, {0.9f, 0.9f, 0.9f}));
    auto material1 = Dielectric::create(1.5f);
    auto material2 = Lambertian::from_color({0.4f, 0.2f, 0.1f});
    auto material3 = Metal::create({0.7f, 0.6f, 0.5f}, 0.f);
    objects.push_back(Sphere::create({0.f, -1000.f, 0.f}, 1000.f, ground));
    objects.push_back(Sphere::create({-4.f, 1.f, 0.f}, 1.f, material2));
    objects.push_back(Sphere::create({4.f, 1.f, 0.f}, 1.f, material3));
    return Scene({Sphere::create({0.f, 1.f, 0.f}, 1.f, material1)}, objects, {0.7f, 0.8f, 1.f}, name);
}
// scene.rs:106-114
inline Scene two_spheres(const std::string& name) {
    auto mat = Lambertian::create(CheckerTexture::create({0.2f, 0.3f, 0.1f}, {0.9f, 0.9f, 0.9f}));
    return Scene({}, {Sphere::create({0.f, -10.f, 0.f}, 10.f, mat), Sphere::create({0.f, 10.f, 0.f}, 10.f, mat)}, {0.7f, 0.8f, 1.f}, name);
}
// scene.rs:116-125
inline Scene two_perlin_spheres(const std::string& name, BuildRng& g) {
    auto pertex = NoiseTexture::create(4.f, g);
    return Scene({}, {Sphere::create({0.f, -1000.f, 0.f}, 1000.f, Lambertian::create(pertex)), Sphere::create({0.f, 2.f, 0.f}, 2.f, Lambertian::create(pertex))},
                 {0.7f, 0.8f, 1.f}, name);
}
// scene.rs:127-131
inline Scene earth(const std::string& name, const EarthMap& em) {
    auto tex = ImageTexture::from_rgba8(em.rgba, em.width, em.height);
    return Scene({}, {Sphere::create({0.f, 0.f, 0.f}, 2.f, Lambertian::create(tex))}, {0.7f, 0.8f, 1.f}, name);
}
// scene.rs:133-146
inline Scene simplelight(const std::string& name, BuildRng& g) {
    auto pertex = NoiseTexture::create(4.f, g);
    auto diff_light = DiffuseLight::from_color({4.f, 4.f, 4.f});
    return Scene({}, {Sphere::create({0.f, -1000.f, 0.f}, 1000.f, Lambertian::create(pertex)), Sphere::create({0.f, 2.f, 0.f}, 2.f, Lambertian::create(pertex)),
                      rect(diff_light, -4.f, 3.f, 1.f, 5.f, 3.f, MIRT_AXIS_XY)},
                 {0.f, 0.f, 0.f}, name);
}

inline SharedHittable rotated_box(RGB mx, const SharedMaterial& m, float deg, RGB off) {
    return Translation::create(RotationY::create(Cuboid::create({0.f, 0.f, 0.f}, mx, m), deg), off);
}

// scene.rs:148-180
inline Scene cornell_box(const std::string& label) {
    auto red = Lambertian::from_color({0.65f, 0.05f, 0.05f});
    auto white = Lambertian::from_color({0.73f, 0.73f, 0.73f});
    auto green = Lambertian::from_color({0.12f, 0.45f, 0.15f});
    auto light = DiffuseLight::from_color({15.f, 15.f, 15.f});
    const float length = 555.f;
    auto glass = Dielectric::create(1.5f);
    std::vector<SharedHittable> lights = {FlipFace::create(rect(light, length - 1.f, 213.f, 227.f, 343.f, 332.f, MIRT_AXIS_XZ)),
                                          Sphere::create({190.f, 90.f, 190.f}, 90.f, glass)};
    std::vector<SharedHittable> objects = {rect(green, length, 0.f, 0.f, length, length, MIRT_AXIS_YZ),
                                           rect(red, 0.f, 0.f, 0.f, length, length, MIRT_AXIS_YZ),
                                           rect(white, 0.f, 0.f, 0.f, length, length, MIRT_AXIS_XZ),
                                           rect(white, length, 0.f, 0.f, length, length, MIRT_AXIS_XZ),
                                           rect(white, length, 0.f, 0.f, length, length, MIRT_AXIS_XY),
                                           rotated_box({165.f, 330.f, 165.f}, white, 15.f, {265.f, 0.f, 295.f})};
    return Scene(lights, objects, {0.f, 0.f, 0.f}, label);
}
// scene.rs:182-221
inline Scene cornell_smoke(const std::string& label) {
    auto red = Lambertian::from_color({0.65f, 0.05f, 0.05f});
    auto white = Lambertian::from_color({0.73f, 0.73f, 0.73f});
    auto green = Lambertian::from_color({0.12f, 0.45f, 0.15f});
    auto light = DiffuseLight::from_color({7.f, 7.f, 7.f});
    const float length = 555.f;
    std::vector<SharedHittable> lights = {FlipFace::create(rect(light, length - 1.f, 113.f, 127.f, 443.f, 432.f, MIRT_AXIS_XZ))};
    std::vector<SharedHittable> objects = {
        rect(green, length, 0.f, 0.f, length, length, MIRT_AXIS_YZ),
        rect(red, 0.f, 0.f, 0.f, length, length, MIRT_AXIS_YZ),
        rect(white, 0.f, 0.f, 0.f, length, length, MIRT_AXIS_XZ),
        rect(white, length, 0.f, 0.f, length, length, MIRT_AXIS_XZ),
        rect(white, length, 0.f, 0.f, length, length, MIRT_AXIS_XY),
        ConstantMedium::create_c(rotated_box({165.f, 330.f, 165.f}, white, 15.f, {265.f, 0.f, 295.f}), 0.01f, {0.f, 0.f, 0.f}),
        ConstantMedium::create_c(rotated_box({165.f, 165.f, 165.f}, white, -18.f, {130.f, 0.f, 65.f}), 0.01f, {1.f, 1.f, 1.f})};
    return Scene(lights, objects, {0.f, 0.f, 0.f}, label);
}
// scene.rs:223-275
inline Scene final_scene(const std::string& label, const EarthMap& em, BuildRng& g) {
    auto ground = Lambertian::from_color({0.48f, 0.83f, 0.53f});
    const int boxes_per_side = 20;
    const float w = 100.f;
    std::vector<SharedHittable> boxes;
    for (int i = 0; i < boxes_per_side; ++i)
        for (int j = 0; j < boxes_per_side; ++j) {
            RGB xyz0{-1000.f + (float)i * w, 0.f, -1000.f + (float)j * w};
            RGB xyz1{xyz0[0] + w, g.get_rand_range(1.f, 101.f), xyz0[2] + w};
            boxes.push_back(Cuboid::create(xyz0, xyz1, ground));
        }
    auto boxes_bvh = BVHNode::create(boxes);
    auto light = DiffuseLight::from_color({15.f, 15.f, 15.f});
    auto xz = FlipFace::create(rect(light, 554.f, 123.f, 147.f, 423.f, 412.f, MIRT_AXIS_XZ));
    RGB center1{400.f, 400.f, 200.f};
    RGB center2{center1[0] + 1.f * 30.f, center1[1] + 0.f * 30.f, center1[2] + 0.f * 30.f};
    auto moving_sphere = Sphere::create_moving(center1, center2, 0.f, 1.f, 50.f, Lambertian::from_color({0.7f, 0.3f, 0.1f}));
    auto dielectric_sphere = Sphere::create({260.f, 150.f, 45.f}, 50.f, Dielectric::create(1.5f));
    auto metal_sphere = Sphere::create({0.f, 150.f, 145.f}, 50.f, Metal::create({0.8f, 0.8f, 0.9f}, 1.f));
    auto boundary = Sphere::create({360.f, 150.f, 145.f}, 70.f, Dielectric::create(1.5f));
    auto medium1 = ConstantMedium::create_c(boundary, 0.2f, {0.2f, 0.4f, 0.9f});
    auto boundary1 = Sphere::create({0.f, 0.f, 0.f}, 500.f, Dielectric::create(1.5f));
    auto medium2 = ConstantMedium::create_c(boundary1, 0.00001f, {1.f, 1.f, 1.f});
    auto earth_s = Sphere::create({400.f, 200.f, 400.f}, 100.f, Lambertian::create(ImageTexture::from_rgba8(em.rgba, em.width, em.height)));
    auto pertext = NoiseTexture::create(0.1f, g);
    auto perlin_s = Sphere::create({220.f, 280.f, 300.f}, 80.f, Lambertian::create(pertext));
    auto white = Lambertian::from_color({0.73f, 0.73f, 0.73f});
    std::vector<SharedHittable> boxes2;
    for (int i = 0; i < 1000; ++i) boxes2.push_back(Sphere::create(g.get_rand_vec3_range(0.f, 165.f), 10.f, white));
    auto b2 = Translation::create(RotationY::create(BVHNode::create(boxes2), 15.f), {-100.f, 270.f, 395.f});
    return Scene({xz, dielectric_sphere}, {boxes_bvh, moving_sphere, metal_sphere, medium1, medium2, earth_s, perlin_s, b2}, {0.f, 0.f, 0.f}, label);
}
}  // namespace scenes

// scene.rs:11-23
inline Scene select_scene(const std::string& name, const SceneOptions& opt = SceneOptions()) {
    BuildRng g(opt.scene_seed);
    if (name == "random") return scenes::create_random_scene(name, opt.random_num, g);
    if (name == "2psp") return scenes::two_perlin_spheres(name, g);
    if (name == "earth") return scenes::earth(name, opt.earthmap);
    if (name == "simplelight") return scenes::simplelight(name, g);
    if (name == "cornell") return scenes::cornell_box(name);
    if (name == "final") return scenes::final_scene(name, opt.earthmap, g);
    if (name == "smoke") return scenes::cornell_smoke(name);
    return scenes::two_spheres(name);
}

// camera.rs:5-69, 96-139
struct Camera {
    mirt_camera c{};
    float aspect_ratio = 1.f, vfov = 0.f, focus_dist = 0.f;
    float w[3]{}, vup[3]{};

    static void cross3(const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
    }
    // camera.rs:29-69 — u = vup x w and v = w x u are not normalised.
    static Camera create(const float lookfrom[3], const float direction[3], const float vup_[3], float vfov_, float aspect, float aperture,
                         float focus, float t0, float t1) {
        Camera cam;
        float theta = vfov_ / 180.0f * PI;
        float h = std::tan(theta / 2.f);
        float viewport_height = 2.f * h;
        float viewport_width = viewport_height * aspect;
        for (int a = 0; a < 3; ++a) { cam.w[a] = -direction[a]; cam.vup[a] = vup_[a]; }
        cross3(vup_, cam.w, cam.c.u);
        cross3(cam.w, cam.c.u, cam.c.v);
        for (int a = 0; a < 3; ++a) {
            cam.c.horizontal[a] = (focus * viewport_width) * cam.c.u[a];
            cam.c.vertical[a] = (focus * viewport_height) * cam.c.v[a];
            cam.c.origin[a] = lookfrom[a];
            cam.c.lower_left_corner[a] = ((cam.c.origin[a] - cam.c.horizontal[a] / 2.f) - cam.c.vertical[a] / 2.f) - cam.w[a] * focus;
        }
        cam.c.lens_radius = aperture / 2.f;
        cam.c.time0 = t0; cam.c.time1 = t1;
        cam.aspect_ratio = aspect; cam.vfov = vfov_; cam.focus_dist = focus;
        return cam;
    }
    // camera.rs:96-139
    static Camera select_camera(float aspect_ratio, const std::string& scene) {
        float aperture = 0.1f;
        const float dist_to_focus = 10.f;
        float vfov = 20.f;
        const float vup[3] = {0.f, 1.f, 0.f};
        float lookfrom[3] = {13.f * 2.f, 1.5f * 2.f, 3.f * 2.f};
        float lookat[3] = {0.f, 2.f, 0.f};
        float direction[3] = {lookat[0] - lookfrom[0], lookat[1] - lookfrom[1], lookat[2] - lookfrom[2]};
        auto set = [&](float fx, float fy, float fz, float ax, float ay, float az) {
            lookfrom[0] = fx; lookfrom[1] = fy; lookfrom[2] = fz;
            lookat[0] = ax; lookat[1] = ay; lookat[2] = az;
            for (int a = 0; a < 3; ++a) direction[a] = lookat[a] - lookfrom[a];
        };
        if (scene == "cornell" || scene == "smoke") {
            set(278.f, 278.f, -800.f, 278.f, 278.f, 0.f);
            vfov = 40.0f;
            aperture = 0.f;
        } else if (scene == "simplelight") {
        } else if (scene == "final") {
            set(478.f, 278.f, -600.f, 278.f, 278.f, 0.f);
            vfov = 40.f;
        } else {
            lookfrom[0] = 13.f; lookfrom[1] = 2.f; lookfrom[2] = 3.f;
            direction[0] = -13.f; direction[1] = -2.f; direction[2] = -3.f;
        }
        float n = std::sqrt((direction[0] * direction[0] + direction[1] * direction[1]) + direction[2] * direction[2]);
        float dir[3] = {direction[0] / n, direction[1] / n, direction[2] / n};
        return create(lookfrom, dir, vup, vfov, aspect_ratio, aperture, dist_to_focus, 0.f, 1.f);
    }
};

}  // namespace rt
