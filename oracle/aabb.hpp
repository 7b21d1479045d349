// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Restates BVHNode of src/aabb.rs:63-135 of the reference.
#pragma once
#include <algorithm>
#include <stdexcept>
#include "ray.hpp"

namespace orc {

enum : uint32_t { BUILD_STREAM_CONTENT = 0, BUILD_STREAM_BVH = 1 };

struct BVHNode : Hittable {
    SharedHittable left, right;
    AABB bbox;
    // Oracle-only: the object list in the caller's order, kept so leaf ids can be numbered
    // independently of the (random) tree shape.
    std::vector<SharedHittable> input_order;

    // aabb.rs:102-108
    static float box_compare(const SharedHittable& a, int axis) {
        auto b = a->bounding_box(0.f, 0.f);
        if (!b) throw std::runtime_error("No bounding_box!");
        return b->minimum[axis];
    }

    // aabb.rs:71-100. The split axis is a random draw per node (:73). In the reference it comes
    // from the same unseeded generator as the scene contents; here it is drawn on its own build
    // stream so that scene CONTENT draws do not depend on the tree shape.
    BVHNode(const std::vector<SharedHittable>& objects_in, size_t begin, size_t end, float time0, float time1) {
        std::vector<SharedHittable> objects(objects_in.begin() + begin, objects_in.begin() + end);
        int axis;
        {
            StreamScope s(BUILD_STREAM_BVH);
            axis = get_rand_int_range(0, 3);
        }
        size_t obj_span = objects.size();
        if (obj_span == 1) {
            left = right = objects[0];
        } else if (obj_span == 2) {
            if (box_compare(objects[0], axis) < box_compare(objects[1], axis)) {
                left = objects[0]; right = objects[1];
            } else {
                left = objects[1]; right = objects[0];
            }
        } else {
            // slice::sort_by is a stable sort
            std::stable_sort(objects.begin(), objects.end(), [axis](const SharedHittable& x, const SharedHittable& y) {
                return box_compare(x, axis) < box_compare(y, axis);
            });
            size_t mid = obj_span / 2;
            left = std::make_shared<BVHNode>(objects, 0, mid, time0, time1);
            right = std::make_shared<BVHNode>(objects, mid, obj_span, time0, time1);
        }
        auto b = AABB::surrounding_box(left->bounding_box(time0, time1), right->bounding_box(time0, time1));
        if (!b) throw std::runtime_error("No bounding_box");
        bbox = *b;
    }

    static std::shared_ptr<BVHNode> build(const std::vector<SharedHittable>& objects, float time0, float time1) {
        auto n = std::make_shared<BVHNode>(objects, 0, objects.size(), time0, time1);
        n->input_order = objects;
        return n;
    }

    // aabb.rs:112-126 — left first, then right with t_max cut to the left hit: right wins ties.
    std::optional<HitRecord> hit(const Ray& ray, float t_min, float t_max) const override {
        if (!bbox.hit(ray, t_min, t_max)) return std::nullopt;
        auto hit_left = left->hit(ray, t_min, t_max);
        if (hit_left) {
            auto hit_right = right->hit(ray, t_min, hit_left->t);
            return hit_right ? hit_right : hit_left;
        }
        return right->hit(ray, t_min, t_max);
    }
    std::optional<AABB> bounding_box(float, float) const override { return bbox; }
};

}  // namespace orc
