// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// C entry points over the CPU restatement of the reference renderer, for the test-suite, the
// smoke check and the cpu_baseline leg of the benchmark (loaded with ctypes). Parity status: the
// reference ships no tests, golden images or known-answer vectors, and cannot be compiled here
// (Rust toolchain absent), so this oracle is pinned only by the analytic known answers derived
// from the reference's scene and camera constants (tests/golden/cornell_first_hit_kat.json) and
// by the numeric properties of its samplers — "parity unpinned" by any reference-produced output.
#include <cstring>
#include <memory>
#include <string>
#include "renderer.hpp"

using namespace orc;

namespace {
struct Handle {
    std::unique_ptr<Renderer> r;
    std::vector<uint8_t> earth;
    uint32_t ew = 0, eh = 0;
};
thread_local std::string g_err;
}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

void orc_philox4x32_10(const uint32_t* key2, const uint32_t* ctr4, uint32_t* out4) { philox4x32_10(key2, ctr4, out4); }

// Renderer::new (renderer.rs:41) over select_scene (scene.rs:11) and Camera::select_camera
// (camera.rs:96) with aspect = width / height (cli.rs:25).
void* orc_renderer_create(const char* scene_name, uint32_t width, uint32_t height, uint32_t scene_seed, int random_num,
                          const uint8_t* earth_rgba_top_first, uint32_t ew, uint32_t eh) {
    try {
        auto h = std::make_unique<Handle>();
        SceneParams p;
        p.scene_seed = scene_seed;
        p.random_num = random_num;
        if (earth_rgba_top_first && ew && eh) {
            h->earth.assign(earth_rgba_top_first, earth_rgba_top_first + (size_t)ew * eh * 4);
            h->ew = ew; h->eh = eh;
            p.earthmap = ImageData{h->earth.data(), ew, eh};
        }
        std::string name(scene_name);
        Scene scene = select_scene(name, p);
        Camera cam = Camera::select_camera((float)width / (float)height, name);
        h->r = std::make_unique<Renderer>(width, height, std::move(scene), std::move(cam));
        return h.release();
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}

void orc_renderer_destroy(void* hv) { delete static_cast<Handle*>(hv); }

int orc_num_leaf_ids(void* hv) { return static_cast<Handle*>(hv)->r->scene.num_leaf_ids; }
int orc_num_lights(void* hv) { return (int)static_cast<Handle*>(hv)->r->scene.lights->objects.size(); }

// origin, lower_left_corner, horizontal, vertical, u, v (3 floats each), len_radius, time0, time1
void orc_camera_get(void* hv, float* out21) {
    const Camera& c = static_cast<Handle*>(hv)->r->camera;
    const V3* vs[6] = {&c.origin, &c.lower_left_corner, &c.horizontal, &c.vertical, &c.u, &c.v};
    for (int i = 0; i < 6; ++i) { out21[i * 3] = vs[i]->x; out21[i * 3 + 1] = vs[i]->y; out21[i * 3 + 2] = vs[i]->z; }
    out21[18] = c.len_radius; out21[19] = c.time0; out21[20] = c.time1;
}

// Sum of sanitised samples [spp_begin, spp_begin + spp_count) per pixel; sums[(y*W + x)*3 + c],
// y = 0 at the image bottom. counters4 = {segments, sphere tests, rect tests, aabb tests}.
// Samples per pixel of the whole render of which later orc_render_sums calls render ranges (decides the pixel
// jitter, renderer.rs:175); 0 = every call stands for itself.
void orc_set_total_spp(void* hv, uint32_t spp_total) { static_cast<Handle*>(hv)->r->spp_total = spp_total; }

int orc_render_sums(void* hv, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth, uint32_t seed, int threads,
                    float* sums, uint64_t* counters4) {
    try {
        Renderer& r = *static_cast<Handle*>(hv)->r;
        r.multisample = spp_count; r.max_depth = max_depth; r.seed = seed; r.spp_begin = spp_begin; r.threads = threads;
        r.render_sums(sums);
        if (counters4) {
            counters4[0] = r.totals.segments; counters4[1] = r.totals.sphere_tests;
            counters4[2] = r.totals.rect_tests; counters4[3] = r.totals.aabb_tests;
        }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// Renderer::draw (renderer.rs:56): RGBA8, top row first.
int orc_draw(void* hv, uint32_t spp, uint32_t max_depth, uint32_t seed, int threads, uint8_t* frame, size_t frame_len) {
    try {
        Renderer& r = *static_cast<Handle*>(hv)->r;
        r.multisample = spp; r.max_depth = max_depth; r.seed = seed; r.spp_begin = 0; r.threads = threads; r.dirty = true;
        r.draw(frame, frame_len);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// world.hit(ray, 0.001, inf) (renderer.rs:201) for n rays; rays = n x {origin 3, direction 3,
// time}. Ray i positions the RNG at (pixel = i, sample = 0, bounce = 0) for the media draws.
// aov (optional) = n x 12: point 3, normal 3, front_face, u, v, then the hit material's
// attenuation (Lambertian / Isotropic texture value, Metal albedo, 1 for Dielectric) or, for a
// DiffuseLight, its emission.
int orc_first_hit(void* hv, const float* rays, size_t n, uint32_t seed, int32_t* prim_id, float* t_out, float* aov) {
    try {
        Renderer& r = *static_cast<Handle*>(hv)->r;
        for (size_t i = 0; i < n; ++i) {
            const float* q = rays + i * 7;
            Ray ray(V3(q[0], q[1], q[2]), V3(q[3], q[4], q[5]), q[6]);
            RngCtx& g = rng();
            g.begin_bounce((uint32_t)i, 0, 0, seed);
            g.stream = STREAM_SCATTER;
            auto hit = r.scene.world->hit(ray, 0.001f, INF);
            prim_id[i] = hit ? hit->prim_id : -1;
            t_out[i] = hit ? hit->t : INF;
            if (aov) {
                float* a = aov + i * 12;
                std::memset(a, 0, 12 * sizeof(float));
                if (hit) {
                    a[0] = hit->point.x; a[1] = hit->point.y; a[2] = hit->point.z;
                    a[3] = hit->normal.x; a[4] = hit->normal.y; a[5] = hit->normal.z;
                    a[6] = hit->front_face ? 1.f : 0.f; a[7] = hit->uv[0]; a[8] = hit->uv[1];
                    Color c(0, 0, 0);
                    const Material* m = hit->material;
                    if (auto* l = dynamic_cast<const Lambertian*>(m)) c = l->albedo->value(hit->uv, hit->point);
                    else if (auto* is = dynamic_cast<const Isotropic*>(m)) c = is->albedo->value(hit->uv, hit->point);
                    else if (auto* me = dynamic_cast<const Metal*>(m)) c = me->albedo;
                    else if (dynamic_cast<const Dielectric*>(m)) c = V3(1, 1, 1);
                    else if (auto* dl = dynamic_cast<const DiffuseLight*>(m)) c = dl->texture->value(hit->uv, hit->point);
                    a[9] = c.x; a[10] = c.y; a[11] = c.z;
                }
            }
        }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// Fixed camera rays: Camera::get_ray (camera.rs:86) with the lens offset forced to 0 and the time
// forced to `time`, at s = x/(W-1), t = y/(H-1) (the multisample == 1 branch of
// renderer.rs:175-183). rays = W*H x 7, index y*W + x.
void orc_camera_rays(void* hv, float time, float* rays) {
    Renderer& r = *static_cast<Handle*>(hv)->r;
    Camera cam = r.camera;
    cam.fixed = true;
    cam.fixed_time = time;
    for (uint32_t y = 0; y < r.height; ++y)
        for (uint32_t x = 0; x < r.width; ++x) {
            float s = ((float)x + 0.f) / (float)(r.width - 1);
            float t = ((float)y + 0.f) / (float)(r.height - 1);
            Ray ray = cam.get_ray(s, t);
            float* q = rays + ((size_t)y * r.width + x) * 7;
            q[0] = ray.origin.x; q[1] = ray.origin.y; q[2] = ray.origin.z;
            q[3] = ray.direction.x; q[4] = ray.direction.y; q[5] = ray.direction.z; q[6] = ray.time;
        }
}

// One radiance sample of ray_color (renderer.rs:197) for explicit rays, un-sanitised;
// ray i uses (pixel = i, sample). out = n x 3.
int orc_ray_color(void* hv, const float* rays, size_t n, uint32_t sample, uint32_t max_depth, uint32_t seed, float* out) {
    try {
        Renderer& r = *static_cast<Handle*>(hv)->r;
        r.seed = seed;
        for (size_t i = 0; i < n; ++i) {
            const float* q = rays + i * 7;
            Ray ray(V3(q[0], q[1], q[2]), V3(q[3], q[4], q[5]), q[6]);
            Color c = r.ray_color(ray, max_depth, (uint32_t)i, sample, 0);
            out[i * 3] = c.x; out[i * 3 + 1] = c.y; out[i * 3 + 2] = c.z;
        }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// Sampler probes (rand_gen.rs:45-90). which: 0 = random_cosine_direction, 1 = in_unit_sphere,
// 2 = in_unit_disk, 3 = on_unit_sphere, 4 = get_rand (x only). Draw i uses key (i, 0).
void orc_sampler(int which, size_t n, uint32_t seed, float* out3) {
    for (size_t i = 0; i < n; ++i) {
        RngCtx& g = rng();
        g.begin_bounce((uint32_t)i, 0, 0, seed);
        g.stream = STREAM_SCATTER;
        V3 v(0, 0, 0);
        switch (which) {
            case 0: v = random_cosine_direction(); break;
            case 1: v = rand_vec3_in_unit_sphere(); break;
            case 2: v = rand_vec3_in_unit_disk(); break;
            case 3: v = rand_vec3_on_unit_sphere(); break;
            default: v.x = get_rand(); break;
        }
        out3[i * 3] = v.x; out3[i * 3 + 1] = v.y; out3[i * 3 + 2] = v.z;
    }
}

// Light-list probes: HittableList::pdf_val (ray.rs:149) and ::random (ray.rs:152) of scene.lights,
// and the mixture of renderer.rs:207-221 (dir, pdf value, cosine-pdf value) at a surface point.
float orc_lights_pdf(void* hv, const float* origin, const float* dir) {
    Renderer& r = *static_cast<Handle*>(hv)->r;
    return r.scene.lights->pdf_val(V3(origin[0], origin[1], origin[2]), V3(dir[0], dir[1], dir[2]));
}
void orc_lights_random(void* hv, const float* origin, size_t n, uint32_t seed, float* out3) {
    Renderer& r = *static_cast<Handle*>(hv)->r;
    for (size_t i = 0; i < n; ++i) {
        RngCtx& g = rng();
        g.begin_bounce((uint32_t)i, 0, 0, seed);
        g.stream = STREAM_SCATTER;
        V3 v = r.scene.lights->random(V3(origin[0], origin[1], origin[2]));
        out3[i * 3] = v.x; out3[i * 3 + 1] = v.y; out3[i * 3 + 2] = v.z;
    }
}
float orc_cosine_pdf_value(const float* w, const float* dir) {
    CosinePDF p(V3(w[0], w[1], w[2]));
    return p.value(V3(dir[0], dir[1], dir[2]));
}
void orc_onb(const float* w, float* uvw9) {
    ONB o = ONB::build_from_w(V3(w[0], w[1], w[2]));
    const V3* vs[3] = {&o.u, &o.v, &o.w};
    for (int i = 0; i < 3; ++i) { uvw9[i * 3] = vs[i]->x; uvw9[i * 3 + 1] = vs[i]->y; uvw9[i * 3 + 2] = vs[i]->z; }
}
uint8_t orc_float_to_rgb(float v) { return Renderer::float_to_rgb(v); }

// Leaf primitives in id order, 12 floats each: {kind (0 sphere, 1 rect, 2 medium), ...}
//   sphere: center0 3, radius, center1 3, moving, time0, time1
//   rect:   axis, k, lo0, lo1, hi0, hi1, flip-face seen on the way down
//   medium: neg_inv_density
// Used to check that an independently written scene builder declares the same scene.
static void dump_walk(const Hittable* h, bool flip, float* out) {
    if (auto* s = dynamic_cast<const Sphere*>(h)) {
        float* o = out + (size_t)s->prim_id * 12;
        o[0] = 0; o[1] = s->center0.x; o[2] = s->center0.y; o[3] = s->center0.z; o[4] = s->radius;
        o[5] = s->center1.x; o[6] = s->center1.y; o[7] = s->center1.z; o[8] = s->moving ? 1.f : 0.f; o[9] = s->time0; o[10] = s->time1;
    } else if (auto* r = dynamic_cast<const AxisAlignedRect*>(h)) {
        float* o = out + (size_t)r->prim_id * 12;
        int id[3];
        axis_indexes(r->axis, id);
        o[0] = 1; o[1] = (float)(int)r->axis; o[2] = r->k; o[3] = r->p0[id[0]]; o[4] = r->p0[id[1]]; o[5] = r->p1[id[0]]; o[6] = r->p1[id[1]];
        o[7] = flip ? 1.f : 0.f;
    } else if (auto* m = dynamic_cast<const ConstantMedium*>(h)) {
        float* o = out + (size_t)m->prim_id * 12;
        o[0] = 2; o[1] = m->neg_inv_density;
        dump_walk(m->boundary.get(), flip, out);
    } else if (auto* l = dynamic_cast<const HittableList*>(h)) { for (auto& o : l->objects) dump_walk(o.get(), flip, out); }
    else if (auto* c = dynamic_cast<const Cuboid*>(h)) dump_walk(c->sides.get(), flip, out);
    else if (auto* t = dynamic_cast<const Translation*>(h)) dump_walk(t->obj.get(), flip, out);
    else if (auto* ro = dynamic_cast<const RotationY*>(h)) dump_walk(ro->obj.get(), flip, out);
    else if (auto* f = dynamic_cast<const FlipFace*>(h)) dump_walk(f->obj.get(), !flip, out);
    else if (auto* b = dynamic_cast<const BVHNode*>(h)) { for (auto& o : b->input_order) dump_walk(o.get(), flip, out); }
}
void orc_dump_prims(void* hv, float* out12) {
    Renderer& r = *static_cast<Handle*>(hv)->r;
    std::memset(out12, 0, (size_t)r.scene.num_leaf_ids * 12 * sizeof(float));
    dump_walk(r.scene.world.get(), false, out12);
}

}  // extern "C"
