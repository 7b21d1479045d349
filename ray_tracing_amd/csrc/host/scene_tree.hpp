// Host-side mirror of the reference's scene API: the object tree a caller builds with
// Sphere / AxisAlignedRect / Cuboid / RotationY / Translation / ConstantMedium / FlipFace / BVHNode /
// HittableList (src/geo.rs, src/aabb.rs, src/ray.rs), its materials and textures (src/material.rs,
// src/texture.rs, src/perlin.rs), Scene::new (src/scene.rs:285-294) — and the flattening of that
// tree into the plain arrays of mirt_scene_desc. Nothing here intersects rays: the tree only
// records what the caller declared; tracing happens on the device.
#pragma once
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/mirt.h"

namespace rt {

using RGB = std::array<float, 3>;
constexpr float PI = 3.14159265358979323846f;

// ---- seeded stand-in for the reference's unseeded rand::random (rand_gen.rs:7-42) used by the
// scene builders: sequential draws of Philox4x32-10 with key (scene_seed, 0x5CE9E000),
// counter (block, 0, 0, 0).
class BuildRng {
   public:
    explicit BuildRng(uint32_t scene_seed) : seed_(scene_seed) {}
    float get_rand() {
        uint32_t block = next_ >> 2;
        if (block != have_) {
            uint32_t key[2] = {seed_, 0x5CE9E000u}, ctr[4] = {block, 0u, 0u, 0u};
            mirt_philox4x32_10(key, ctr, buf_);
            have_ = block;
        }
        uint32_t w = buf_[next_++ & 3u];
        return (float)(w >> 8) * (1.0f / 16777216.0f);
    }
    float get_rand_range(float mn, float mx) { return mn + (mx - mn) * get_rand(); }
    size_t get_rand_usize_range(size_t mn, size_t mx) {
        float v = get_rand_range((float)mn, (float)mx);
        return v <= 0.f ? 0 : (size_t)v;
    }
    RGB get_rand_vec3_range(float mn, float mx) {
        float x = get_rand_range(mn, mx), y = get_rand_range(mn, mx), z = get_rand_range(mn, mx);
        return {x, y, z};
    }

   private:
    uint32_t seed_, next_ = 0, have_ = 0xffffffffu;
    uint32_t buf_[4];
};

// ---- textures (texture.rs) --------------------------------------------------------------------
struct Perlin {  // perlin.rs:8-22, 73-86
    float grad[256][3];
    uint8_t perm[3][256];
    explicit Perlin(BuildRng& g) {
        for (int i = 0; i < 256; ++i) {
            RGB v = g.get_rand_vec3_range(-1.f, 1.f);
            float n = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
            for (int a = 0; a < 3; ++a) grad[i][a] = v[a] / n;
        }
        for (int a = 0; a < 3; ++a) {
            for (int i = 0; i < 256; ++i) perm[a][i] = (uint8_t)i;
            for (int i = 255; i >= 1; --i) {
                size_t target = g.get_rand_usize_range(0, (size_t)i);
                std::swap(perm[a][i], perm[a][target]);
            }
        }
    }
};

struct Texture {
    uint32_t kind = MIRT_TEX_SOLID;
    RGB color{0, 0, 0}, color2{0, 0, 0};
    float scale = 0.f;
    std::shared_ptr<Perlin> perlin;
    // image: RGBA8 rows as indexed by ImageTexture::value, i.e. after the flip of texture.rs:84
    std::vector<uint8_t> texels;
    uint32_t img_w = 0, img_h = 0;
};
using SharedTexture = std::shared_ptr<Texture>;

struct SolidColor {
    static SharedTexture create(RGB c) { auto t = std::make_shared<Texture>(); t->color = c; return t; }
};
struct CheckerTexture {  // texture.rs:33: (even, odd)
    static SharedTexture create(RGB even, RGB odd) {
        auto t = std::make_shared<Texture>();
        t->kind = MIRT_TEX_CHECKER; t->color = even; t->color2 = odd;
        return t;
    }
};
struct NoiseTexture {  // texture.rs:55-60
    static SharedTexture create(float scale, BuildRng& g) {
        auto t = std::make_shared<Texture>();
        t->kind = MIRT_TEX_NOISE; t->scale = scale; t->perlin = std::make_shared<Perlin>(g);
        return t;
    }
};
struct ImageTexture {
    // texture.rs:82-86: decoded RGBA8 (top row first, as an image decoder returns it) is flipped
    // vertically. Null / empty input is the "file missing" case of texture.rs:78,99-101.
    static SharedTexture from_rgba8(const uint8_t* top_first, uint32_t w, uint32_t h) {
        auto t = std::make_shared<Texture>();
        t->kind = MIRT_TEX_IMAGE;
        if (top_first && w && h) {
            t->img_w = w; t->img_h = h;
            t->texels.resize((size_t)w * h * 4);
            for (uint32_t y = 0; y < h; ++y)
                std::copy(top_first + (size_t)(h - 1 - y) * w * 4, top_first + (size_t)(h - y) * w * 4, t->texels.begin() + (size_t)y * w * 4);
        }
        return t;
    }
};

// ---- materials (material.rs) ------------------------------------------------------------------
struct Material {
    uint32_t kind = MIRT_MAT_LAMBERTIAN;
    SharedTexture texture;
    RGB albedo{0, 0, 0};
    float param = 0.f;
};
using SharedMaterial = std::shared_ptr<Material>;

struct Lambertian {
    static SharedMaterial create(SharedTexture t) { auto m = std::make_shared<Material>(); m->texture = std::move(t); return m; }
    static SharedMaterial from_color(RGB c) { return create(SolidColor::create(c)); }
};
struct Metal {  // material.rs:78-83: fuzz = min(fuzz, 1)
    static SharedMaterial create(RGB albedo, float fuzz) {
        auto m = std::make_shared<Material>();
        m->kind = MIRT_MAT_METAL; m->albedo = albedo; m->param = std::fmin(fuzz, 1.f);
        return m;
    }
};
struct Dielectric {  // material.rs:108-113
    static SharedMaterial create(float ior) {
        if (!(ior > 0.f)) throw std::invalid_argument("index_of_refraction must be > 0");
        auto m = std::make_shared<Material>();
        m->kind = MIRT_MAT_DIELECTRIC; m->param = ior;
        return m;
    }
};
struct DiffuseLight {
    static SharedMaterial create(SharedTexture t) { auto m = std::make_shared<Material>(); m->kind = MIRT_MAT_DIFFUSE_LIGHT; m->texture = std::move(t); return m; }
    static SharedMaterial from_color(RGB c) { return create(SolidColor::create(c)); }
};
struct Isotropic {
    static SharedMaterial create(SharedTexture t) { auto m = std::make_shared<Material>(); m->kind = MIRT_MAT_ISOTROPIC; m->texture = std::move(t); return m; }
};

// ---- hittables (geo.rs, aabb.rs, ray.rs) --------------------------------------------------------
struct Hittable {
    enum Kind { SPHERE, RECT, CUBOID, TRANSLATION, ROTATION_Y, MEDIUM, FLIP_FACE, BVH, LIST } kind;
    // sphere
    float center0[3]{}, center1[3]{}, radius = 0, time0 = 0, time1 = 0;
    bool moving = false;
    // rect
    uint32_t axis = 0;
    float k = 0, lo[2]{}, hi[2]{};
    // cuboid
    float cmin[3]{}, cmax[3]{};
    // translation / rotation
    float offset[3]{}, degrees = 0;
    // medium
    float neg_inv_density = 0;
    SharedMaterial material;  // leaf material; medium: the Isotropic phase function
    std::vector<std::shared_ptr<Hittable>> children;  // wrappers: 1 child; cuboid: 6 rects; bvh/list: members
    explicit Hittable(Kind k_) : kind(k_) {}
};
using SharedHittable = std::shared_ptr<Hittable>;

struct Sphere {
    static SharedHittable create(RGB c, float r, SharedMaterial m) {  // geo.rs:26-41
        auto h = std::make_shared<Hittable>(Hittable::SPHERE);
        for (int a = 0; a < 3; ++a) h->center0[a] = h->center1[a] = c[a];
        h->radius = r; h->material = std::move(m);
        return h;
    }
    static SharedHittable create_moving(RGB c0, RGB c1, float t0, float t1, float r, SharedMaterial m) {  // geo.rs:42-61
        if (t0 == t1) throw std::invalid_argument("moving sphere: time0 == time1");
        auto h = std::make_shared<Hittable>(Hittable::SPHERE);
        for (int a = 0; a < 3; ++a) { h->center0[a] = c0[a]; h->center1[a] = c1[a]; }
        h->time0 = t0; h->time1 = t1; h->radius = r; h->moving = true; h->material = std::move(m);
        return h;
    }
};
struct AxisAlignedRect {  // geo.rs:178-192
    static SharedHittable create(SharedMaterial m, float k, std::array<float, 2> c0, std::array<float, 2> c1, uint32_t axis) {
        if (!(c0[0] < c1[0] && c0[1] < c1[1])) throw std::invalid_argument("rect extents");
        auto h = std::make_shared<Hittable>(Hittable::RECT);
        h->material = std::move(m); h->k = k; h->axis = axis;
        h->lo[0] = c0[0]; h->lo[1] = c0[1]; h->hi[0] = c1[0]; h->hi[1] = c1[1];
        return h;
    }
};
struct Cuboid {  // geo.rs:252-275: XY(kmin,kmax), XZ(kmin,kmax), YZ(kmin,kmax)
    static SharedHittable create(RGB mn, RGB mx, const SharedMaterial& m) {
        auto h = std::make_shared<Hittable>(Hittable::CUBOID);
        for (int a = 0; a < 3; ++a) { h->cmin[a] = mn[a]; h->cmax[a] = mx[a]; }
        static const int idx[3][3] = {{0, 1, 2}, {0, 2, 1}, {1, 2, 0}};
        for (uint32_t axis = 0; axis < 3; ++axis) {
            const int* am = idx[axis];
            for (float kk : {mn[am[2]], mx[am[2]]})
                h->children.push_back(AxisAlignedRect::create(m, kk, {mn[am[0]], mn[am[1]]}, {mx[am[0]], mx[am[1]]}, axis));
        }
        return h;
    }
};
struct Translation {  // geo.rs:296-303
    static SharedHittable create(SharedHittable obj, RGB off) {
        auto h = std::make_shared<Hittable>(Hittable::TRANSLATION);
        for (int a = 0; a < 3; ++a) h->offset[a] = off[a];
        h->children.push_back(std::move(obj));
        return h;
    }
};
struct RotationY {  // geo.rs:340-368
    static SharedHittable create(SharedHittable obj, float degree) {
        auto h = std::make_shared<Hittable>(Hittable::ROTATION_Y);
        h->degrees = degree;
        h->children.push_back(std::move(obj));
        return h;
    }
};
struct ConstantMedium {  // geo.rs:400-409
    static SharedHittable create(SharedHittable boundary, float density, SharedTexture albedo) {
        auto h = std::make_shared<Hittable>(Hittable::MEDIUM);
        h->neg_inv_density = -1.f / density;
        h->material = Isotropic::create(std::move(albedo));
        h->children.push_back(std::move(boundary));
        return h;
    }
    static SharedHittable create_c(SharedHittable boundary, float density, RGB color) {
        return create(std::move(boundary), density, SolidColor::create(color));
    }
};
struct FlipFace {  // geo.rs:447-451
    static SharedHittable create(SharedHittable obj) {
        auto h = std::make_shared<Hittable>(Hittable::FLIP_FACE);
        h->children.push_back(std::move(obj));
        return h;
    }
};
struct BVHNode {  // aabb.rs:71: the member list; the library builds the hierarchy at upload
    static SharedHittable create(std::vector<SharedHittable> objects) {
        if (objects.empty()) throw std::invalid_argument("BVH over no objects");
        auto h = std::make_shared<Hittable>(Hittable::BVH);
        h->children = std::move(objects);
        return h;
    }
};
struct HittableList {
    static SharedHittable create(std::vector<SharedHittable> objects) {
        auto h = std::make_shared<Hittable>(Hittable::LIST);
        h->children = std::move(objects);
        return h;
    }
};

// ---- Scene (scene.rs:277-295) and its flattening ------------------------------------------------
struct FlatScene {
    std::vector<mirt_prim> prims;
    std::vector<mirt_entry> world;
    std::vector<uint32_t> lights;
    std::vector<mirt_material> materials;
    std::vector<mirt_texture> textures;
    std::vector<mirt_perlin> perlins;
    std::vector<mirt_image> images;
    std::vector<mirt_instance> instances;
    std::vector<SharedTexture> keep_alive;  // owners of the texel buffers `images` points into
    float background[3]{};
    mirt_scene_desc desc() const {
        mirt_scene_desc d{};
        d.prims = prims.data(); d.n_prims = (uint32_t)prims.size();
        d.world = world.data(); d.n_world = (uint32_t)world.size();
        d.lights = lights.data(); d.n_lights = (uint32_t)lights.size();
        d.materials = materials.data(); d.n_materials = (uint32_t)materials.size();
        d.textures = textures.data(); d.n_textures = (uint32_t)textures.size();
        d.perlins = perlins.data(); d.n_perlins = (uint32_t)perlins.size();
        d.images = images.data(); d.n_images = (uint32_t)images.size();
        d.instances = instances.data(); d.n_instances = (uint32_t)instances.size();
        for (int a = 0; a < 3; ++a) d.background[a] = background[a];
        return d;
    }
};

struct Scene {
    SharedHittable lights;  // HittableList
    SharedHittable world;   // HittableList, lights appended last (scene.rs:287)
    RGB background;
    std::string label;
    Scene(std::vector<SharedHittable> light_objs, std::vector<SharedHittable> world_objs, RGB bg, std::string name)
        : background(bg), label(std::move(name)) {
        lights = HittableList::create(std::move(light_objs));
        world_objs.push_back(lights);
        world = HittableList::create(std::move(world_objs));
    }
    FlatScene flatten() const;
};

// ---- flattening ---------------------------------------------------------------------------------
namespace detail {

struct Xform {  // Translation(RotationY(.)) collected on the way down
    bool has_rot = false, has_off = false;
    float degrees = 0, off[3] = {0, 0, 0};
};

class Flattener {
   public:
    explicit Flattener(FlatScene& out) : f_(out) {
        f_.instances.push_back(mirt_instance{0.f, 1.f, {0.f, 0.f, 0.f}});  // identity, never applied
    }

    void run(const Scene& sc) {
        for (int a = 0; a < 3; ++a) f_.background[a] = sc.background[a];
        // world list in order; the lights list is its last element
        for (const auto& obj : sc.world->children) {
            if (obj == sc.lights) {
                for (const auto& l : obj->children) {
                    size_t before = f_.prims.size();
                    top_level(l);
                    if (f_.prims.size() != before + 1 || f_.prims[before].kind == MIRT_PRIM_MEDIUM || f_.prims[before].instance)
                        throw std::invalid_argument("a light must be a single un-transformed sphere or rect (FlipFace allowed)");
                    f_.lights.push_back((uint32_t)before);
                }
            } else {
                top_level(obj);
            }
        }
        // boundary primitives of the media go after all world primitives, in medium order
        for (auto& pm : pending_media_) {
            uint32_t first = (uint32_t)f_.prims.size();
            leaves(pm.boundary, Xform{}, false, /*allow_medium=*/false);
            f_.prims[pm.prim].boundary_first = first;
            f_.prims[pm.prim].boundary_count = (uint32_t)f_.prims.size() - first;
        }
    }

   private:
    struct PendingMedium { uint32_t prim; SharedHittable boundary; };
    FlatScene& f_;
    std::map<const Material*, uint32_t> mat_ids_;
    std::map<const Texture*, uint32_t> tex_ids_;
    std::map<const Perlin*, uint32_t> perlin_ids_;
    std::vector<PendingMedium> pending_media_;

    uint32_t texture_id(const SharedTexture& t) {
        if (!t) throw std::invalid_argument("material without texture");
        auto it = tex_ids_.find(t.get());
        if (it != tex_ids_.end()) return it->second;
        mirt_texture q{};
        q.kind = t->kind;
        for (int a = 0; a < 3; ++a) { q.color[a] = t->color[a]; q.color2[a] = t->color2[a]; }
        q.scale = t->scale;
        if (t->kind == MIRT_TEX_NOISE) {
            auto pit = perlin_ids_.find(t->perlin.get());
            if (pit == perlin_ids_.end()) {
                mirt_perlin p{};
                std::memcpy(p.grad, t->perlin->grad, sizeof(p.grad));
                std::memcpy(p.perm, t->perlin->perm, sizeof(p.perm));
                pit = perlin_ids_.emplace(t->perlin.get(), (uint32_t)f_.perlins.size()).first;
                f_.perlins.push_back(p);
            }
            q.perlin = pit->second;
        } else if (t->kind == MIRT_TEX_IMAGE) {
            q.image = (uint32_t)f_.images.size();
            f_.images.push_back(mirt_image{t->img_w, t->img_h, t->texels.empty() ? nullptr : t->texels.data()});
            f_.keep_alive.push_back(t);
        }
        uint32_t id = (uint32_t)f_.textures.size();
        f_.textures.push_back(q);
        tex_ids_[t.get()] = id;
        return id;
    }
    uint32_t material_id(const SharedMaterial& m) {
        if (!m) throw std::invalid_argument("primitive without material");
        auto it = mat_ids_.find(m.get());
        if (it != mat_ids_.end()) return it->second;
        mirt_material q{};
        q.kind = m->kind;
        q.param = m->param;
        for (int a = 0; a < 3; ++a) q.albedo[a] = m->albedo[a];
        if (m->kind == MIRT_MAT_LAMBERTIAN || m->kind == MIRT_MAT_DIFFUSE_LIGHT || m->kind == MIRT_MAT_ISOTROPIC) q.texture = texture_id(m->texture);
        uint32_t id = (uint32_t)f_.materials.size();
        f_.materials.push_back(q);
        mat_ids_[m.get()] = id;
        return id;
    }
    uint32_t instance_id(const Xform& x) {
        if (!x.has_rot && !x.has_off) return 0;
        float radians = x.degrees / 180.0f * PI;  // camera.rs:25-27
        mirt_instance in{};
        in.sin_theta = x.has_rot ? std::sin(radians) : 0.f;
        in.cos_theta = x.has_rot ? std::cos(radians) : 1.f;
        for (int a = 0; a < 3; ++a) in.offset[a] = x.off[a];
        for (uint32_t i = 1; i < f_.instances.size(); ++i)
            if (std::memcmp(&f_.instances[i], &in, sizeof(in)) == 0) return i;
        f_.instances.push_back(in);
        return (uint32_t)f_.instances.size() - 1;
    }

    // Peel Translation / RotationY wrappers; only Translation(RotationY(x)), Translation(x) and
    // RotationY(x) are representable (every use in scene.rs has one of these forms).
    static const Hittable* peel(const Hittable* h, Xform& x) {
        for (;;) {
            if (h->kind == Hittable::TRANSLATION) {
                if (x.has_rot || x.has_off) throw std::invalid_argument("unsupported transform nesting (Translation inside a transform)");
                x.has_off = true;
                for (int a = 0; a < 3; ++a) x.off[a] = h->offset[a];
            } else if (h->kind == Hittable::ROTATION_Y) {
                if (x.has_rot) throw std::invalid_argument("unsupported transform nesting (RotationY inside RotationY)");
                x.has_rot = true;
                x.degrees = h->degrees;
            } else {
                return h;
            }
            h = h->children[0].get();
        }
    }

    void emit_leaf(const Hittable* h, const Xform& x, bool flip) {
        mirt_prim p{};
        p.material = material_id(h->material);
        p.instance = instance_id(x);
        p.flags = flip ? MIRT_PRIM_FLIP_FACE : 0u;
        if (h->kind == Hittable::SPHERE) {
            p.kind = MIRT_PRIM_SPHERE;
            for (int a = 0; a < 3; ++a) { p.center0[a] = h->center0[a]; p.center1[a] = h->center1[a]; }
            p.radius = h->radius; p.time0 = h->time0; p.time1 = h->time1;
            if (h->moving) p.flags |= MIRT_PRIM_MOVING;
        } else {
            p.kind = MIRT_PRIM_RECT;
            p.axis = h->axis; p.k = h->k;
            p.lo[0] = h->lo[0]; p.lo[1] = h->lo[1]; p.hi[0] = h->hi[0]; p.hi[1] = h->hi[1];
        }
        f_.prims.push_back(p);
    }

    // Emit every leaf primitive under `obj` (depth-first, list order).
    void leaves(const SharedHittable& obj, Xform x, bool flip, bool allow_medium) {
        const Hittable* h = peel(obj.get(), x);
        switch (h->kind) {
            case Hittable::SPHERE:
            case Hittable::RECT: emit_leaf(h, x, flip); break;
            case Hittable::CUBOID:
            case Hittable::LIST:
                for (const auto& c : h->children) leaves(c, x, flip, allow_medium);
                break;
            case Hittable::FLIP_FACE: leaves(h->children[0], x, !flip, allow_medium); break;
            case Hittable::MEDIUM: {
                if (!allow_medium) throw std::invalid_argument("a medium cannot bound another medium");
                if (x.has_rot || x.has_off) throw std::invalid_argument("transform a medium's boundary, not the medium");
                mirt_prim p{};
                p.kind = MIRT_PRIM_MEDIUM;
                p.material = material_id(h->material);
                p.neg_inv_density = h->neg_inv_density;
                pending_media_.push_back(PendingMedium{(uint32_t)f_.prims.size(), h->children[0]});
                f_.prims.push_back(p);
                break;
            }
            case Hittable::BVH: throw std::invalid_argument("BVHNode is only supported as a direct world member (optionally under Translation/RotationY)");
            default: throw std::invalid_argument("unknown hittable");
        }
    }

    void top_level(const SharedHittable& obj) {
        Xform x;
        const Hittable* h = peel(obj.get(), x);
        if (h->kind == Hittable::BVH) {
            mirt_entry e{};
            e.kind = MIRT_ENTRY_BVH;
            e.first_prim = (uint32_t)f_.prims.size();
            e.instance = instance_id(x);
            uint32_t per = 0;
            for (const auto& m : h->children) {
                size_t before = f_.prims.size();
                leaves(m, Xform{}, false, false);
                uint32_t n = (uint32_t)(f_.prims.size() - before);
                if (per == 0) per = n;
                if (n != per || n == 0) throw std::invalid_argument("BVH members must flatten to the same number of primitives each");
            }
            e.prims_per_object = per;
            e.prim_count = (uint32_t)f_.prims.size() - e.first_prim;
            f_.world.push_back(e);
            return;
        }
        uint32_t first = (uint32_t)f_.prims.size();
        leaves(obj, Xform{}, false, true);
        uint32_t n = (uint32_t)f_.prims.size() - first;
        if (n == 0) return;
        if (!f_.world.empty() && f_.world.back().kind == MIRT_ENTRY_LIST &&
            f_.world.back().first_prim + f_.world.back().prim_count == first) {
            f_.world.back().prim_count += n;  // consecutive list members share one entry
        } else {
            mirt_entry e{};
            e.kind = MIRT_ENTRY_LIST;
            e.first_prim = first; e.prim_count = n; e.prims_per_object = 1;
            f_.world.push_back(e);
        }
    }
};
}  // namespace detail

inline FlatScene Scene::flatten() const {
    FlatScene f;
    detail::Flattener fl(f);
    fl.run(*this);
    return f;
}

}  // namespace rt
