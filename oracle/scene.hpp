// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Restates src/scene.rs:11-295 of the reference (all eight scene builders and Scene::new).
// The reference's builders draw from an unseeded generator, so `random`, `final` and every Perlin
// table differ from run to run there. Here the same draws, in the same order, come from a seeded
// build stream (key = (scene_seed, BUILD_KEY1)); BVH split-axis draws use a second build stream
// (aabb.hpp) so that scene contents do not depend on tree shapes.
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "aabb.hpp"
#include "camera.hpp"
#include "geo.hpp"

namespace orc {

constexpr uint32_t BUILD_KEY1 = 0x5CE9E000u;

struct ImageData {
    const uint8_t* texels = nullptr;  // RGBA8, top row first
    uint32_t w = 0, h = 0;
};

// scene.rs:277-295
struct Scene {
    std::shared_ptr<HittableList> lights;
    std::shared_ptr<HittableList> world;
    Color background;
    std::string label;
    std::vector<ConstantMedium*> media;
    int num_leaf_ids = 0;

    // scene.rs:285-294 — the lights list is appended to the world as its LAST element.
    Scene(std::vector<SharedHittable> light_objs, std::vector<SharedHittable> world_objs, Color bg, const std::string& name)
        : background(bg), label(name) {
        lights = std::make_shared<HittableList>(std::move(light_objs));
        world_objs.push_back(lights);
        world = std::make_shared<HittableList>(std::move(world_objs));
        assign_ids();
    }

    // Oracle-only: number leaf primitives in world order (spheres, rects and constant media count
    // one each; a BVH is numbered in the order of the object list it was built from), then the
    // boundary primitives of the media in medium order.
    void assign_ids() {
        int next = 0;
        number(world.get(), next, true);
        for (size_t m = 0; m < media.size(); ++m) number(media[m]->boundary.get(), next, false);
        num_leaf_ids = next;
    }
    void number(Hittable* h, int& next, bool collect_media) {
        if (auto* s = dynamic_cast<Sphere*>(h)) { if (s->prim_id < 0) s->prim_id = next++; }
        else if (auto* r = dynamic_cast<AxisAlignedRect*>(h)) { if (r->prim_id < 0) r->prim_id = next++; }
        else if (auto* m = dynamic_cast<ConstantMedium*>(h)) {
            if (m->prim_id < 0) {
                m->prim_id = next++;
                if (collect_media) { m->medium_index = (uint32_t)media.size(); media.push_back(m); }
            }
        }
        else if (auto* l = dynamic_cast<HittableList*>(h)) { for (auto& o : l->objects) number(o.get(), next, collect_media); }
        else if (auto* c = dynamic_cast<Cuboid*>(h)) number(c->sides.get(), next, collect_media);
        else if (auto* t = dynamic_cast<Translation*>(h)) number(t->obj.get(), next, collect_media);
        else if (auto* ro = dynamic_cast<RotationY*>(h)) number(ro->obj.get(), next, collect_media);
        else if (auto* f = dynamic_cast<FlipFace*>(h)) number(f->obj.get(), next, collect_media);
        else if (auto* b = dynamic_cast<BVHNode*>(h)) { for (auto& o : b->input_order) number(o.get(), next, collect_media); }
    }
};

inline SharedHittable sphere(V3 c, float r, SharedMaterial m) { return std::make_shared<Sphere>(c, r, std::move(m)); }
inline SharedHittable rect(SharedMaterial m, float k, float a0, float a1, float b0, float b1, AlignedAxis ax) {
    float c0[2] = {a0, a1}, c1[2] = {b0, b1};
    return std::make_shared<AxisAlignedRect>(std::move(m), k, c0, c1, ax);
}
inline SharedMaterial lambert_tex(SharedTexture t) { return std::make_shared<Lambertian>(std::move(t)); }

// scene.rs:76-103
inline SharedHittable create_random_sphere(int ai, int bi) {
    float a = (float)ai, b = (float)bi;
    float mat = get_rand();
    float cx = a + 0.9f * get_rand();
    float cz = b + 0.9f * get_rand();
    V3 center(cx, 0.2f, cz);
    if (norm(center - V3(4.f, 0.2f, 0.f)) > 0.9f) {
        if (mat < 0.8f) {
            V3 a1 = get_rand_vec3_range(0.f, 1.f);
            V3 a2 = get_rand_vec3_range(0.f, 1.f);
            SharedMaterial material = Lambertian::from_color(cmul(a1, a2));
            V3 center2 = center + V3(0.f, get_rand_range(0.f, 0.5f), 0.f);
            return std::make_shared<Sphere>(center, center2, 0.f, 1.f, 0.2f, material);
        } else if (mat < 0.95f) {
            V3 albedo = get_rand_vec3_range(0.5f, 1.f);
            float fuzz = get_rand_range(0.f, 0.5f);
            return sphere(center, 0.2f, std::make_shared<Metal>(albedo, fuzz));
        } else {
            return sphere(center, 0.2f, std::make_shared<Dielectric>(1.5f));
        }
    }
    return nullptr;
}

// scene.rs:46-74; `num` is 11 in the reference (:47) and a parameter here for the scaled scene.
inline Scene create_random_scene(const std::string& name, int num) {
    std::vector<SharedHittable> objects;
    for (int a = -num; a < num; ++a)
        for (int b = -num; b < num; ++b)
            if (auto s = create_random_sphere(a, b)) objects.push_back(s);
    auto material_ground = lambert_tex(std::make_shared<CheckerTexture>(V3(0.2f, 0.3f, 0.1f), V3::repeat(0.9f)));
    auto material1 = std::make_shared<Dielectric>(1.5f);
    auto material2 = Lambertian::from_color(V3(0.4f, 0.2f, 0.1f));
    auto material3 = std::make_shared<Metal>(V3(0.7f, 0.6f, 0.5f), 0.f);
    objects.push_back(sphere(V3(0.f, -1000.f, 0.f), 1000.f, material_ground));
    objects.push_back(sphere(V3(-4.f, 1.f, 0.f), 1.f, material2));
    objects.push_back(sphere(V3(4.f, 1.f, 0.f), 1.f, material3));
    std::vector<SharedHittable> lights = {sphere(V3(0.f, 1.f, 0.f), 1.f, material1)};
    return Scene(lights, objects, V3(0.7f, 0.8f, 1.f), name);
}

// scene.rs:106-114
inline Scene two_spheres(const std::string& name) {
    auto mat = lambert_tex(std::make_shared<CheckerTexture>(V3(0.2f, 0.3f, 0.1f), V3::repeat(0.9f)));
    return Scene({}, {sphere(V3(0.f, -10.f, 0.f), 10.f, mat), sphere(V3(0.f, 10.f, 0.f), 10.f, mat)}, V3(0.7f, 0.8f, 1.f), name);
}
// scene.rs:116-125
inline Scene two_perlin_spheres(const std::string& name) {
    auto pertex = std::make_shared<NoiseTexture>(4.f);
    return Scene({}, {sphere(V3(0.f, -1000.f, 0.f), 1000.f, lambert_tex(pertex)), sphere(V3(0.f, 2.f, 0.f), 2.f, lambert_tex(pertex))},
                 V3(0.7f, 0.8f, 1.f), name);
}
// scene.rs:127-131
inline Scene earth(const std::string& name, const ImageData& img) {
    auto tex = std::make_shared<ImageTexture>(img.texels, img.w, img.h);
    return Scene({}, {sphere(V3(0, 0, 0), 2.f, lambert_tex(tex))}, V3(0.7f, 0.8f, 1.f), name);
}
// scene.rs:133-146 — the rect is declared with k = -4, c0 = (3,1), c1 = (5,3) on XY.
inline Scene simplelight(const std::string& name) {
    auto pertex = std::make_shared<NoiseTexture>(4.f);
    auto diff_light = DiffuseLight::from_color(V3::repeat(4.f));
    return Scene({}, {sphere(V3(0.f, -1000.f, 0.f), 1000.f, lambert_tex(pertex)), sphere(V3(0.f, 2.f, 0.f), 2.f, lambert_tex(pertex)),
                      rect(diff_light, -4.f, 3.f, 1.f, 5.f, 3.f, AlignedAxis::XY)},
                 V3(0, 0, 0), name);
}

inline SharedHittable rotated_box(V3 mx, const SharedMaterial& m, float deg, V3 off) {
    SharedHittable cuboid = std::make_shared<Cuboid>(V3(0, 0, 0), mx, m);
    cuboid = std::make_shared<RotationY>(cuboid, deg);
    return std::make_shared<Translation>(cuboid, off);
}

// scene.rs:148-180
inline Scene cornell_box(const std::string& label) {
    auto red = Lambertian::from_color(V3(0.65f, 0.05f, 0.05f));
    auto white = Lambertian::from_color(V3::repeat(0.73f));
    auto green = Lambertian::from_color(V3(0.12f, 0.45f, 0.15f));
    auto light = DiffuseLight::from_color(V3::repeat(15.f));
    const float length = 555.f;
    auto glass = std::make_shared<Dielectric>(1.5f);
    std::vector<SharedHittable> lights = {
        std::make_shared<FlipFace>(rect(light, length - 1.f, 213.f, 227.f, 343.f, 332.f, AlignedAxis::XZ)),
        sphere(V3(190.f, 90.f, 190.f), 90.f, glass)};
    std::vector<SharedHittable> objects = {
        rect(green, length, 0.f, 0.f, length, length, AlignedAxis::YZ),
        rect(red, 0.f, 0.f, 0.f, length, length, AlignedAxis::YZ),
        rect(white, 0.f, 0.f, 0.f, length, length, AlignedAxis::XZ),
        rect(white, length, 0.f, 0.f, length, length, AlignedAxis::XZ),
        rect(white, length, 0.f, 0.f, length, length, AlignedAxis::XY),
        rotated_box(V3(165.f, 330.f, 165.f), white, 15.f, V3(265.f, 0.f, 295.f))};
    return Scene(lights, objects, V3(0, 0, 0), label);
}

// scene.rs:182-221
inline Scene cornell_smoke(const std::string& label) {
    auto red = Lambertian::from_color(V3(0.65f, 0.05f, 0.05f));
    auto white = Lambertian::from_color(V3::repeat(0.73f));
    auto green = Lambertian::from_color(V3(0.12f, 0.45f, 0.15f));
    auto light = DiffuseLight::from_color(V3::repeat(7.f));
    const float length = 555.f;
    std::vector<SharedHittable> lights = {
        std::make_shared<FlipFace>(rect(light, length - 1.f, 113.f, 127.f, 443.f, 432.f, AlignedAxis::XZ))};
    std::vector<SharedHittable> objects = {
        rect(green, length, 0.f, 0.f, length, length, AlignedAxis::YZ),
        rect(red, 0.f, 0.f, 0.f, length, length, AlignedAxis::YZ),
        rect(white, 0.f, 0.f, 0.f, length, length, AlignedAxis::XZ),
        rect(white, length, 0.f, 0.f, length, length, AlignedAxis::XZ),
        rect(white, length, 0.f, 0.f, length, length, AlignedAxis::XY),
        std::make_shared<ConstantMedium>(rotated_box(V3(165.f, 330.f, 165.f), white, 15.f, V3(265.f, 0.f, 295.f)), 0.01f,
                                         std::make_shared<SolidColor>(V3(0, 0, 0))),
        std::make_shared<ConstantMedium>(rotated_box(V3(165.f, 165.f, 165.f), white, -18.f, V3(130.f, 0.f, 65.f)), 0.01f,
                                         std::make_shared<SolidColor>(V3(1, 1, 1)))};
    return Scene(lights, objects, V3(0, 0, 0), label);
}

// scene.rs:223-275. Draw order: 400 box heights; the Perlin tables of the marble texture; 1000
// sphere centres. (`boxes2` holds spheres, not boxes.) The boundary sphere of the first medium is
// not itself part of the world.
inline Scene final_scene(const std::string& label, const ImageData& img) {
    auto ground = Lambertian::from_color(V3(0.48f, 0.83f, 0.53f));
    const int boxes_per_side = 20;
    const float w = 100.f;
    std::vector<SharedHittable> boxes;
    for (int i = 0; i < boxes_per_side; ++i)
        for (int j = 0; j < boxes_per_side; ++j) {
            V3 xyz0(-1000.f + (float)i * w, 0.f, -1000.f + (float)j * w);
            V3 xyz1(xyz0.x + w, get_rand_range(1.f, 101.f), xyz0.z + w);
            boxes.push_back(std::make_shared<Cuboid>(xyz0, xyz1, ground));
        }
    SharedHittable boxes_bvh = BVHNode::build(boxes, 0.f, 1.f);

    auto light = DiffuseLight::from_color(V3::repeat(15.f));
    SharedHittable xz = std::make_shared<FlipFace>(rect(light, 554.f, 123.f, 147.f, 423.f, 412.f, AlignedAxis::XZ));

    V3 center1(400.f, 400.f, 200.f);
    V3 center2 = center1 + V3(1.f, 0.f, 0.f) * 30.f;
    SharedHittable moving_sphere =
        std::make_shared<Sphere>(center1, center2, 0.f, 1.f, 50.f, Lambertian::from_color(V3(0.7f, 0.3f, 0.1f)));
    SharedHittable dielectric_sphere = sphere(V3(260.f, 150.f, 45.f), 50.f, std::make_shared<Dielectric>(1.5f));
    SharedHittable metal_sphere = sphere(V3(0.f, 150.f, 145.f), 50.f, std::make_shared<Metal>(V3(0.8f, 0.8f, 0.9f), 1.f));

    SharedHittable boundary = sphere(V3(360.f, 150.f, 145.f), 70.f, std::make_shared<Dielectric>(1.5f));
    SharedHittable medium1 = std::make_shared<ConstantMedium>(boundary, 0.2f, std::make_shared<SolidColor>(V3(0.2f, 0.4f, 0.9f)));
    SharedHittable boundary1 = sphere(V3(0, 0, 0), 500.f, std::make_shared<Dielectric>(1.5f));
    SharedHittable medium2 = std::make_shared<ConstantMedium>(boundary1, 0.00001f, std::make_shared<SolidColor>(V3(1, 1, 1)));

    SharedHittable earth_s = sphere(V3(400.f, 200.f, 400.f), 100.f, lambert_tex(std::make_shared<ImageTexture>(img.texels, img.w, img.h)));
    auto pertext = std::make_shared<NoiseTexture>(0.1f);
    SharedHittable perlin_s = sphere(V3(220.f, 280.f, 300.f), 80.f, lambert_tex(pertext));

    auto white = Lambertian::from_color(V3::repeat(0.73f));
    std::vector<SharedHittable> boxes2;
    for (int i = 0; i < 1000; ++i) boxes2.push_back(sphere(get_rand_vec3_range(0.f, 165.f), 10.f, white));
    SharedHittable b2 = BVHNode::build(boxes2, 0.f, 1.f);
    b2 = std::make_shared<RotationY>(b2, 15.f);
    b2 = std::make_shared<Translation>(b2, V3(-100.f, 270.f, 395.f));

    std::vector<SharedHittable> lights = {xz, dielectric_sphere};
    std::vector<SharedHittable> world = {boxes_bvh, moving_sphere, metal_sphere, medium1, medium2, earth_s, perlin_s, b2};
    return Scene(lights, world, V3(0, 0, 0), label);
}

struct SceneParams {
    uint32_t scene_seed = 1;
    int random_num = 11;
    ImageData earthmap;
};

// Two scenes the reference does not have, for known-answer tests only (tests/test_analytic_kats.py works their
// expected radiance out by hand from material.rs:86-100, 194-199 and geo.rs:411-434): a lone fuzzy-metal sphere and a
// lone ball of scattering medium, each under a uniform sky, so that every path's result is a product of albedos
// times the sky, decided only by the draws.
inline Scene kat_metal_sphere(const std::string& name) {
    return Scene({}, {sphere(V3(0.f, 0.f, 0.f), 1.f, std::make_shared<Metal>(V3(0.8f, 0.6f, 0.4f), 0.5f))}, V3(0.5f, 0.75f, 1.f), name);
}
inline Scene kat_medium_ball(const std::string& name) {
    auto glass = std::make_shared<Dielectric>(1.5f);   // the boundary's material is never used (geo.rs:393-435)
    return Scene({}, {std::make_shared<ConstantMedium>(sphere(V3(0.f, 0.f, 0.f), 2.f, glass), 0.7f, std::make_shared<SolidColor>(V3(0.5f, 0.75f, 0.25f)))},
                 V3(1.f, 0.5f, 0.25f), name);
}

// scene.rs:11-23
inline Scene select_scene(const std::string& name, const SceneParams& p) {
    RngCtx& r = rng();
    r.streams.clear();
    r.begin_bounce(p.scene_seed, BUILD_KEY1, 0, 0);
    r.stream = BUILD_STREAM_CONTENT;
    if (name == "random") return create_random_scene(name, p.random_num);
    if (name == "2psp") return two_perlin_spheres(name);
    if (name == "earth") return earth(name, p.earthmap);
    if (name == "simplelight") return simplelight(name);
    if (name == "cornell") return cornell_box(name);
    if (name == "final") return final_scene(name, p.earthmap);
    if (name == "smoke") return cornell_smoke(name);
    if (name == "kat_metal_sphere") return kat_metal_sphere(name);
    if (name == "kat_medium_ball") return kat_medium_ball(name);
    return two_spheres(name);
}

}  // namespace orc
