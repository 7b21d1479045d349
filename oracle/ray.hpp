// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Restates src/ray.rs of the reference: Ray (:7-25), Hittable (:32-47), HitRecord (:50-81),
// HittableList (:83-155). HittableList::new_bvh (:102-113) has no caller in the reference and is
// not restated.
#pragma once
#include <memory>
#include <optional>
#include <vector>
#include "rand_gen.hpp"
#include "vec.hpp"

namespace orc {

struct Material;
struct AABB;

// ray.rs:7-25. `direction` is typed UnitVector3 in the reference but is not always unit length
// (ONB::local_dir, Metal::reflect); nothing renormalises it.
struct Ray {
    V3 origin;
    V3 direction;
    float time = 0.f;
    Ray() : origin(0, 0, 0), direction(0, 1, 0), time(0) {}
    Ray(V3 o, V3 d, float t) : origin(o), direction(d), time(t) {}
    V3 at(float t) const { return origin + t * direction; }
};

// ray.rs:50-70. The reference's default record allocates a placeholder Lambertian (:67); that
// material is always overwritten before use, so the oracle stores a null pointer instead.
// `prim_id` is an oracle-only tag (index of the leaf primitive) used by the first-hit checks.
struct HitRecord {
    V3 point{0, 0, 0};
    V3 normal{0, 1, 0};
    float t = 3.40282347e+38f;  // f32::MAX
    float uv[2] = {0.f, 0.f};
    bool front_face = false;
    const Material* material = nullptr;
    int prim_id = -1;
    // ray.rs:73-80
    void set_face_normal(const Ray& ray, V3 outward_normal) {
        front_face = dot(ray.direction, outward_normal) < 0.f;
        normal = front_face ? outward_normal : -outward_normal;
    }
};

// Work counters (oracle-only instrumentation; thread-local, summed by the renderer).
struct Counters {
    uint64_t segments = 0, sphere_tests = 0, rect_tests = 0, aabb_tests = 0;
};
inline Counters& counters() {
    static thread_local Counters c;
    return c;
}

// ray.rs:32-47
struct Hittable {
    virtual ~Hittable() = default;
    virtual std::optional<HitRecord> hit(const Ray& ray, float t_min, float t_max) const = 0;
    virtual std::optional<AABB> bounding_box(float time0, float time1) const = 0;
    virtual float pdf_val(V3 /*origin*/, V3 /*v*/) const { return 0.f; }
    virtual V3 random(V3 /*origin*/) const { return V3(1, 0, 0); }
    virtual std::shared_ptr<Hittable> get_one() const { return nullptr; }
};
using SharedHittable = std::shared_ptr<Hittable>;

}  // namespace orc

#include "aabb_box.hpp"

namespace orc {

// ray.rs:83-155
struct HittableList : Hittable {
    std::vector<SharedHittable> objects;
    explicit HittableList(std::vector<SharedHittable> objs) : objects(std::move(objs)) {}

    // ray.rs:123-133 — linear scan with a shrinking t_max. Leaf tests accept t == t_max, so of two
    // objects hit at exactly the same t the LATER one in the list wins.
    std::optional<HitRecord> hit(const Ray& ray, float t_min, float t_max) const override {
        std::optional<HitRecord> hit_temp;
        float closest_so_far = t_max;
        for (const auto& object : objects) {
            if (auto new_hit = object->hit(ray, t_min, closest_so_far)) {
                closest_so_far = new_hit->t;
                hit_temp = new_hit;
            }
        }
        return hit_temp;
    }
    // ray.rs:135-140
    std::optional<AABB> bounding_box(float time0, float time1) const override {
        std::optional<AABB> acc;
        for (const auto& cur : objects) acc = AABB::surrounding_box(acc, cur->bounding_box(time0, time1));
        return acc;
    }
    // ray.rs:149-151 — arithmetic mean of the members' pdfs.
    float pdf_val(V3 origin, V3 v) const override {
        float sum = 0.f;
        for (const auto& x : objects) sum += x->pdf_val(origin, v);
        return sum / (float)objects.size();
    }
    // ray.rs:152-154 — member floor(n * xi).
    V3 random(V3 origin) const override {
        return objects[get_rand_usize_range(0, objects.size())]->random(origin);
    }
};

}  // namespace orc
