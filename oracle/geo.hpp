// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Restates src/geo.rs:14-474 of the reference: Sphere, AxisAlignedRect, Cuboid, Translation,
// RotationY, ConstantMedium, FlipFace.
#pragma once
#include <cmath>
#include <memory>
#include "material.hpp"
#include "onb.hpp"
#include "ray.hpp"

namespace orc {

// camera.rs:25-27
inline float degree_to_radian(float degree) { return degree / 180.0f * PI; }

// geo.rs:14-150
struct Sphere : Hittable {
    V3 center0, center1;
    float radius;
    SharedMaterial material;
    float time0 = 0.f, time1 = 0.f;
    bool moving = false;
    int prim_id = -1;

    Sphere(V3 c, float r, SharedMaterial m) : center0(c), center1(c), radius(r), material(std::move(m)) {}
    Sphere(V3 c0, V3 c1, float t0, float t1, float r, SharedMaterial m)
        : center0(c0), center1(c1), radius(r), material(std::move(m)), time0(t0), time1(t1), moving(true) {}

    // geo.rs:62-69
    V3 get_center(float time) const {
        if (moving) return center0 + (time - time0) / (time1 - time0) * (center1 - center0);
        return center0;
    }
    // geo.rs:71-75
    static void get_sphere_uv(V3 p, float uv[2]) {
        float theta = std::acos(-p.y);
        float phi = std::atan2(-p.z, p.x) + PI;
        uv[0] = phi / (2.f * PI);
        uv[1] = theta / PI;
    }
    // geo.rs:77-85
    V3 random_to_sphere(float d) const {
        float r1 = get_rand();
        float r2 = get_rand();
        float z = 1.f + r2 * (std::sqrt(1.f - radius * radius / d) - 1.f);
        float phi = 2.f * PI * r1;
        float x = std::cos(phi) * std::sqrt(1.f - z * z);
        float y = std::sin(phi) * std::sqrt(1.f - z * z);
        return V3(x, y, z);
    }
    // geo.rs:89-114 — a = |d|^2 (d may be non-unit); both range ends inclusive.
    std::optional<HitRecord> hit(const Ray& ray, float t_min, float t_max) const override {
        ++counters().sphere_tests;
        V3 oc = ray.origin - get_center(ray.time);
        float a = norm_squared(ray.direction);
        float half_b = dot(oc, ray.direction);
        float c = norm_squared(oc) - radius * radius;
        float discriminant = half_b * half_b - a * c;
        if (discriminant < 0.f) return std::nullopt;
        float sqrtd = std::sqrt(discriminant);
        float root = (-half_b - sqrtd) / a;
        if (root < t_min || root > t_max) {
            root = (-half_b + sqrtd) / a;
            if (root < t_min || root > t_max) return std::nullopt;
        }
        HitRecord rec;
        rec.t = root;
        rec.point = ray.at(root);
        V3 outward_normal = (rec.point - get_center(ray.time)) / radius;
        get_sphere_uv(outward_normal, rec.uv);
        rec.set_face_normal(ray, normalize(outward_normal));
        rec.material = material.get();
        rec.prim_id = prim_id;
        return rec;
    }
    // geo.rs:116-130
    std::optional<AABB> bounding_box(float t0, float t1) const override {
        V3 offset = V3::repeat(radius);
        if (moving) {
            V3 c0 = get_center(t0), c1 = get_center(t1);
            return AABB::surrounding_box(AABB(c0 - offset, c0 + offset), AABB(c1 - offset, c1 + offset));
        }
        return AABB(center0 - offset, center0 + offset);
    }
    // geo.rs:136-143 — probe ray at time 0; uses center0; NaN when origin is inside the sphere.
    float pdf_val(V3 origin, V3 v) const override {
        if (hit(Ray(origin, v, 0.f), 0.001f, INF)) {
            float cos_theta_max = std::sqrt(1.f - radius * radius / norm_squared(center0 - origin));
            float solid_angle = 2.f * PI * (1.f - cos_theta_max);
            return 1.f / solid_angle;
        }
        return 0.f;
    }
    // geo.rs:144-149
    V3 random(V3 origin) const override {
        V3 direction = center0 - origin;
        float dis_sq = norm_squared(direction);
        ONB uvw = ONB::build_from_w(normalize(direction));
        return uvw.local_dir(random_to_sphere(dis_sq));
    }
};

// geo.rs:152-167
enum class AlignedAxis { XY = 0, XZ = 1, YZ = 2 };
inline void axis_indexes(AlignedAxis a, int idx[3]) {
    idx[0] = 0; idx[1] = 1; idx[2] = 2;
    switch (a) {
        case AlignedAxis::XY: idx[0] = 0; idx[1] = 1; idx[2] = 2; break;
        case AlignedAxis::XZ: idx[0] = 0; idx[1] = 2; idx[2] = 1; break;
        case AlignedAxis::YZ: idx[0] = 1; idx[1] = 2; idx[2] = 0; break;
    }
}

// geo.rs:168-243
struct AxisAlignedRect : Hittable {
    SharedMaterial material;
    float k;
    V3 p0{0, 0, 0}, p1{0, 0, 0};
    AlignedAxis axis;
    int prim_id = -1;

    // geo.rs:178-192 — only the stored corner points are padded by 1e-4; the hit plane has zero
    // thickness.
    AxisAlignedRect(SharedMaterial m, float k_, const float c0[2], const float c1[2], AlignedAxis ax)
        : material(std::move(m)), k(k_), axis(ax) {
        int am[3];
        axis_indexes(ax, am);
        if (!(c0[0] < c1[0] && c0[1] < c1[1])) throw std::runtime_error("rect extents");
        for (int i = 0; i < 2; ++i) { p0[am[i]] = c0[i]; p1[am[i]] = c1[i]; }
        p0[am[2]] = k - 0.0001f;
        p1[am[2]] = k + 0.0001f;
    }
    // geo.rs:196-216
    std::optional<HitRecord> hit(const Ray& ray, float t_min, float t_max) const override {
        ++counters().rect_tests;
        int id[3];
        axis_indexes(axis, id);
        int xi = id[0], yi = id[1], zi = id[2];
        float t = (k - ray.origin[zi]) / ray.direction[zi];
        if (t < t_min || t > t_max) return std::nullopt;
        V3 xyz = ray.at(t);
        float x = xyz[xi], y = xyz[yi];
        if (x < p0[xi] || x > p1[xi] || y < p0[yi] || y > p1[yi]) return std::nullopt;
        HitRecord rec;
        V3 upper = xyz - p0, lower = p1 - p0;
        rec.uv[0] = upper[xi] / lower[xi];
        rec.uv[1] = upper[yi] / lower[yi];
        V3 outward_normal(0, 0, 0);
        outward_normal[zi] = 1.f;
        rec.set_face_normal(ray, outward_normal);
        rec.material = material.get();
        rec.point = xyz;
        rec.t = t;
        rec.prim_id = prim_id;
        return rec;
    }
    // geo.rs:218-221
    std::optional<AABB> bounding_box(float, float) const override { return AABB(p0, p1); }
    // geo.rs:223-233 — probe ray at time 0.2; |v|^2 is not assumed to be 1.
    float pdf_val(V3 origin, V3 v) const override {
        if (auto rec = hit(Ray(origin, v, 0.2f), 0.001f, INF)) {
            int id[3];
            axis_indexes(axis, id);
            V3 diag = p1 - p0;
            float area = std::fabs(diag[id[0]] * diag[id[1]]);
            float dis_squared = rec->t * rec->t * norm_squared(v);
            float cosine = std::fabs(dot(v, rec->normal));
            return dis_squared / (cosine * area);
        }
        return 0.f;
    }
    // geo.rs:234-242
    V3 random(V3 origin) const override {
        int id[3];
        axis_indexes(axis, id);
        V3 xyz(0, 0, 0);
        xyz[id[0]] = get_rand_range(p0[id[0]], p1[id[0]]);
        xyz[id[1]] = get_rand_range(p0[id[1]], p1[id[1]]);
        xyz[id[2]] = (p0[id[2]] + p1[id[2]]) / 2.f;
        return normalize(xyz - origin);
    }
};

// geo.rs:245-289 — six rects in the order XY(kmin,kmax), XZ(kmin,kmax), YZ(kmin,kmax).
struct Cuboid : Hittable {
    V3 cuboid_min, cuboid_max;
    std::shared_ptr<HittableList> sides;
    Cuboid(V3 mn, V3 mx, const SharedMaterial& material) : cuboid_min(mn), cuboid_max(mx) {
        std::vector<SharedHittable> v;
        for (AlignedAxis axis : {AlignedAxis::XY, AlignedAxis::XZ, AlignedAxis::YZ}) {
            int am[3];
            axis_indexes(axis, am);
            float c0[2] = {mn[am[0]], mn[am[1]]};
            float c1[2] = {mx[am[0]], mx[am[1]]};
            float ks[2] = {mn[am[2]], mx[am[2]]};
            for (float kk : ks) v.push_back(std::make_shared<AxisAlignedRect>(material, kk, c0, c1, axis));
        }
        sides = std::make_shared<HittableList>(std::move(v));
    }
    std::optional<HitRecord> hit(const Ray& ray, float t_min, float t_max) const override {
        return sides->hit(ray, t_min, t_max);
    }
    std::optional<AABB> bounding_box(float, float) const override { return AABB(cuboid_min, cuboid_max); }
    std::shared_ptr<Hittable> get_one() const override { return sides->get_one(); }
};

// geo.rs:291-330 — pdf_val / random are forwarded WITHOUT applying the offset.
struct Translation : Hittable {
    SharedHittable obj;
    V3 offset;
    Translation(SharedHittable o, V3 off) : obj(std::move(o)), offset(off) {}
    std::optional<HitRecord> hit(const Ray& ray, float t_min, float t_max) const override {
        Ray moved(ray.origin - offset, ray.direction, ray.time);
        auto rec = obj->hit(moved, t_min, t_max);
        if (rec) rec->point += offset;
        return rec;
    }
    std::optional<AABB> bounding_box(float t0, float t1) const override {
        auto b = obj->bounding_box(t0, t1);
        if (b) return AABB(b->minimum + offset, b->maximum + offset);
        return std::nullopt;
    }
    float pdf_val(V3 origin, V3 v) const override { return obj->pdf_val(origin, v); }
    V3 random(V3 origin) const override { return obj->random(origin); }
};

// geo.rs:332-391. nalgebra's Rotation3::from_axis_angle(y, theta) is [[c,0,s],[0,1,0],[-s,0,c]];
// its inverse is the transpose. No pdf_val / random (trait defaults apply).
struct RotationY : Hittable {
    SharedHittable obj;
    float sin_t, cos_t;
    std::optional<AABB> bbox;
    V3 rot(V3 p) const { return V3(cos_t * p.x + sin_t * p.z, p.y, -sin_t * p.x + cos_t * p.z); }
    V3 inv_rot(V3 p) const { return V3(cos_t * p.x - sin_t * p.z, p.y, sin_t * p.x + cos_t * p.z); }
    // geo.rs:340-368
    RotationY(SharedHittable o, float degree) : obj(std::move(o)) {
        float radians = degree_to_radian(degree);
        sin_t = std::sin(radians);
        cos_t = std::cos(radians);
        V3 mn = V3::repeat(INF), mx = V3::repeat(-INF);
        if (auto b = obj->bounding_box(0.f, 1.f)) {
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j)
                    for (int kk = 0; kk < 2; ++kk) {
                        V3 vec((float)i, (float)j, (float)kk);
                        V3 xyz = rot(cmul(vec, b->maximum) + cmul(V3::repeat(1.f) - vec, b->minimum));
                        for (int a = 0; a < 3; ++a) {
                            mn[a] = std::fmin(mn[a], xyz[a]);
                            mx[a] = std::fmax(mx[a], xyz[a]);
                        }
                    }
            bbox = AABB(mn, mx);
        }
    }
    // geo.rs:372-385
    std::optional<HitRecord> hit(const Ray& ray, float t_min, float t_max) const override {
        Ray rot_ray(inv_rot(ray.origin), inv_rot(ray.direction), ray.time);
        auto rec = obj->hit(rot_ray, t_min, t_max);
        if (rec) {
            rec->point = rot(rec->point);
            rec->normal = rot(rec->normal);
        }
        return rec;
    }
    std::optional<AABB> bounding_box(float, float) const override { return bbox; }
};

// geo.rs:393-440
struct ConstantMedium : Hittable {
    SharedHittable boundary;
    SharedMaterial phase_function;
    float neg_inv_density;
    int prim_id = -1;
    uint32_t medium_index = 0;  // selects the RNG stream of the free-path draw
    ConstantMedium(SharedHittable b, float density, SharedTexture albedo)
        : boundary(std::move(b)), phase_function(std::make_shared<Isotropic>(std::move(albedo))),
          neg_inv_density(-1.f / density) {}
    // geo.rs:413-435 — entry/exit found with an unbounded t range; distances are in units of the
    // ray parameter (no |direction| factor); the record keeps the default normal (+y),
    // front_face = false and uv = 0.
    std::optional<HitRecord> hit(const Ray& ray, float t_min, float t_max) const override {
        auto rec1 = boundary->hit(ray, -INF, INF);
        if (!rec1) return std::nullopt;
        auto rec2 = boundary->hit(ray, rec1->t + 0.0001f, INF);
        if (!rec2) return std::nullopt;
        rec1->t = std::fmax(std::fmax(t_min, rec1->t), 0.f);
        rec2->t = std::fmin(t_max, rec2->t);
        if (rec1->t >= rec2->t) return std::nullopt;
        float distance_inside_boundary = rec2->t - rec1->t;
        float hit_distance;
        {
            StreamScope s(STREAM_MEDIUM + medium_index);
            hit_distance = neg_inv_density * std::log(get_rand());
        }
        if (hit_distance > distance_inside_boundary) return std::nullopt;
        HitRecord rec;
        rec.t = rec1->t + hit_distance;
        rec.point = ray.at(rec.t);
        rec.material = phase_function.get();
        rec.prim_id = prim_id;
        return rec;
    }
    std::optional<AABB> bounding_box(float t0, float t1) const override { return boundary->bounding_box(t0, t1); }
};

// geo.rs:442-474 — toggles front_face only; the normal is left as it is.
struct FlipFace : Hittable {
    SharedHittable obj;
    explicit FlipFace(SharedHittable o) : obj(std::move(o)) {}
    std::optional<HitRecord> hit(const Ray& ray, float t_min, float t_max) const override {
        auto rec = obj->hit(ray, t_min, t_max);
        if (rec) rec->front_face = !rec->front_face;
        return rec;
    }
    std::optional<AABB> bounding_box(float t0, float t1) const override { return obj->bounding_box(t0, t1); }
    float pdf_val(V3 origin, V3 v) const override { return obj->pdf_val(origin, v); }
    V3 random(V3 origin) const override { return obj->random(origin); }
};

}  // namespace orc
