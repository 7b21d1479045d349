// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Restates src/material.rs:10-200 of the reference. Material::scattering_pdf (:15-17, :63-66) is
// never called by the renderer and is not restated.
#pragma once
#include <cmath>
#include <memory>
#include <optional>
#include "pdf.hpp"
#include "texture.hpp"

namespace orc {

// material.rs:19-36
enum class ScatterKind { Specular, Diffuse, ISO };
struct ScatterRecord {
    ScatterKind kind;
    Ray ray;                      // Specular / ISO
    std::unique_ptr<PDF> pdf;     // Diffuse
    Color attenuation;
};

// material.rs:10-18
struct Material {
    virtual ~Material() = default;
    virtual std::optional<ScatterRecord> scatter(const Ray& ray_in, const HitRecord& rec) const = 0;
    virtual std::optional<Color> emit(const HitRecord&) const { return std::nullopt; }
};
using SharedMaterial = std::shared_ptr<Material>;

// material.rs:37-67
struct Lambertian : Material {
    SharedTexture albedo;
    explicit Lambertian(SharedTexture a) : albedo(std::move(a)) {}
    static SharedMaterial from_color(Color c) { return std::make_shared<Lambertian>(std::make_shared<SolidColor>(c)); }
    std::optional<ScatterRecord> scatter(const Ray&, const HitRecord& rec) const override {
        ScatterRecord s;
        s.kind = ScatterKind::Diffuse;
        s.pdf = std::make_unique<CosinePDF>(rec.normal);
        s.attenuation = albedo->value(rec.uv, rec.point);
        return s;
    }
};

// material.rs:75-77 — v need not be unit length; the result is not renormalised.
inline V3 reflect(V3 v, V3 n) { return v - (2.f * dot(n, v)) * n; }

// material.rs:69-101 — the fuzz vector is drawn even when fuzz == 0; absorbed when the fuzzed
// direction points into the surface.
struct Metal : Material {
    Color albedo;
    float fuzz;
    Metal(Color a, float f) : albedo(a), fuzz(std::fmin(f, 1.f)) {}
    std::optional<ScatterRecord> scatter(const Ray& ray_in, const HitRecord& rec) const override {
        V3 reflected = reflect(ray_in.direction, rec.normal);
        V3 scatter_dir = reflected + fuzz * rand_vec3_in_unit_sphere();
        if (dot(rec.normal, scatter_dir) > 0.f) {
            ScatterRecord s;
            s.kind = ScatterKind::Specular;
            s.ray = Ray(rec.point, normalize(scatter_dir), ray_in.time);
            s.attenuation = albedo;
            return s;
        }
        return std::nullopt;
    }
};

// material.rs:103-152
struct Dielectric : Material {
    float index_of_refraction;
    explicit Dielectric(float ior) : index_of_refraction(ior) {}
    // material.rs:114-119
    static V3 refract(V3 incident, V3 normal, float index_ratio) {
        float cos_theta = std::fmin(-dot(incident, normal), 1.f);
        V3 r_out_perp = index_ratio * (incident + cos_theta * normal);
        V3 r_out_parallel = (-std::sqrt(std::fabs(1.f - norm_squared(r_out_perp)))) * normal;
        return normalize(r_out_perp + r_out_parallel);
    }
    // material.rs:121-125
    static float reflectance(float cosine, float ref_idx) {
        float r0 = (1.f - ref_idx) / (1.f + ref_idx);
        r0 = r0 * r0;
        return r0 + (1.f - r0) * std::pow(1.f - cosine, 5.f);
    }
    // material.rs:129-151 — the uniform draw happens only when refraction is possible (`||`
    // short-circuits, :139).
    std::optional<ScatterRecord> scatter(const Ray& ray_in, const HitRecord& rec) const override {
        float refraction_ratio = rec.front_face ? 1.f / index_of_refraction : index_of_refraction;
        float cos_theta = std::fmin(-dot(ray_in.direction, rec.normal), 1.f);
        float sin_theta = std::sqrt(1.f - cos_theta * cos_theta);
        bool cannot_refract = refraction_ratio * sin_theta > 1.f;
        V3 direction;
        if (cannot_refract || reflectance(cos_theta, refraction_ratio) > get_rand())
            direction = reflect(ray_in.direction, rec.normal);
        else
            direction = refract(ray_in.direction, rec.normal, refraction_ratio);
        ScatterRecord s;
        s.kind = ScatterKind::Specular;
        s.ray = Ray(rec.point, direction, ray_in.time);
        s.attenuation = V3::repeat(1.f);
        return s;
    }
};

// material.rs:154-176 — emits only toward the side flagged front_face.
struct DiffuseLight : Material {
    SharedTexture texture;
    explicit DiffuseLight(SharedTexture t) : texture(std::move(t)) {}
    static SharedMaterial from_color(Color c) { return std::make_shared<DiffuseLight>(std::make_shared<SolidColor>(c)); }
    std::optional<ScatterRecord> scatter(const Ray&, const HitRecord&) const override { return std::nullopt; }
    std::optional<Color> emit(const HitRecord& rec) const override {
        if (rec.front_face) return texture->value(rec.uv, rec.point);
        return std::nullopt;
    }
};

// material.rs:178-200
struct Isotropic : Material {
    SharedTexture albedo;
    explicit Isotropic(SharedTexture a) : albedo(std::move(a)) {}
    std::optional<ScatterRecord> scatter(const Ray& ray_in, const HitRecord& rec) const override {
        ScatterRecord s;
        s.kind = ScatterKind::ISO;
        s.ray = Ray(rec.point, rand_vec3_on_unit_sphere(), ray_in.time);
        s.attenuation = albedo->value(rec.uv, rec.point);
        return s;
    }
};

}  // namespace orc
