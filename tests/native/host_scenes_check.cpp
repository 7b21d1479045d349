// Host-only sanitizer run of the scene mirror (ray_tracing_amd/csrc/host): builds every scene of scene.rs:19-44 (and
// the scaled random-spheres scene), flattens it and checks that the flat description is self-consistent — every
// index a table holds points inside the table it refers to, world entries cover disjoint primitive ranges, lights
// are un-instanced spheres / rects, media name a boundary range outside the world entries — then encodes a PNG.
// Built with -fsanitize=address,undefined by tests/test_hierarchy_build.py; prints "ok".
#include <cstdio>
#include <string>
#include <vector>
#include "../../ray_tracing_amd/csrc/host/png_writer.hpp"
#include "../../ray_tracing_amd/csrc/host/scenes.hpp"

// The scene builders draw their random numbers from the library's Philox block function; this host-only program
// is not linked with the library, so it carries the block function itself (Random123 Philox4x32-10; the library's
// and the oracle's are checked against the published known answers in tests/test_host.py and tests/test_oracle.py).
extern "C" void mirt_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (failures++ < 20) { std::printf("FAIL line %d: ", __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } } while (0)

static void check_scene(const char* name, int random_num) {
    rt::SceneOptions opt;
    opt.random_num = random_num;
    rt::Scene sc = rt::select_scene(name, opt);
    rt::FlatScene flat = sc.flatten();
    const mirt_scene_desc d = flat.desc();
    CHECK(d.n_prims > 0 && d.n_world > 0 && d.n_materials > 0, "%s: empty tables", name);
    std::vector<int> owner(d.n_prims, 0);
    for (uint32_t e = 0; e < d.n_world; ++e) {
        const mirt_entry& w = d.world[e];
        CHECK((uint64_t)w.first_prim + w.prim_count <= d.n_prims, "%s: entry %u range", name, e);
        CHECK(w.instance < std::max(d.n_instances, 1u), "%s: entry %u instance", name, e);
        if (w.kind == MIRT_ENTRY_BVH) CHECK(w.prims_per_object >= 1 && w.prim_count % w.prims_per_object == 0, "%s: entry %u objects", name, e);
        for (uint32_t k = 0; k < w.prim_count && w.first_prim + k < d.n_prims; ++k) ++owner[w.first_prim + k];
    }
    for (uint32_t i = 0; i < d.n_prims; ++i) {
        const mirt_prim& p = d.prims[i];
        CHECK(owner[i] <= 1, "%s: prim %u is in %d world entries", name, i, owner[i]);
        CHECK(p.material < d.n_materials, "%s: prim %u material %u", name, i, p.material);
        CHECK(p.instance < std::max(d.n_instances, 1u), "%s: prim %u instance %u", name, i, p.instance);
        if (p.kind == MIRT_PRIM_MEDIUM) {
            CHECK((uint64_t)p.boundary_first + p.boundary_count <= d.n_prims && p.boundary_count > 0, "%s: medium %u boundary range", name, i);
            for (uint32_t k = 0; k < p.boundary_count && p.boundary_first + k < d.n_prims; ++k)
                CHECK(owner[p.boundary_first + k] == 0, "%s: boundary prim %u is also a world member", name, p.boundary_first + k);
        }
    }
    for (uint32_t i = 0; i < d.n_materials; ++i) CHECK(d.materials[i].texture < std::max(d.n_textures, 1u), "%s: material %u texture", name, i);
    for (uint32_t i = 0; i < d.n_lights; ++i) {
        CHECK(d.lights[i] < d.n_prims, "%s: light %u", name, i);
        if (d.lights[i] < d.n_prims) CHECK(d.prims[d.lights[i]].kind != MIRT_PRIM_MEDIUM && d.prims[d.lights[i]].instance == 0, "%s: light %u kind", name, i);
    }
    const rt::Camera cam = rt::Camera::select_camera(16.f / 9.f, name);
    CHECK(cam.c.horizontal[0] == cam.c.horizontal[0], "%s: camera", name);
    std::printf("%-12s prims %8u  world entries %3u  materials %5u  textures %5u  instances %5u  lights %u\n", name, d.n_prims, d.n_world,
                d.n_materials, d.n_textures, d.n_instances, d.n_lights);
}

int main(int argc, char** argv) {
    for (const char* name : {"random", "2sp", "2psp", "earth", "simplelight", "cornell", "smoke", "final", "no such scene"}) check_scene(name, 11);
    check_scene("random", 60);  // 14 404 spheres through the same generator
    // PNG writer: odd sizes, one pixel, a row longer than a deflate window
    for (auto wh : {std::pair<uint32_t, uint32_t>{1, 1}, {3, 5}, {40000, 2}}) {
        std::vector<uint8_t> px((size_t)wh.first * wh.second * 4);
        for (size_t i = 0; i < px.size(); ++i) px[i] = (uint8_t)(i * 131u);
        const std::vector<uint8_t> png = rt::png_encode_rgba8(px.data(), wh.first, wh.second);
        CHECK(png.size() > 57 && png[1] == 'P' && png[2] == 'N' && png[3] == 'G', "png %ux%u", wh.first, wh.second);
    }
    if (argc > 1) rt::png_save_rgba8(argv[1], std::vector<uint8_t>(4 * 6, 200).data(), 3, 2);
    if (failures) { std::printf("%d checks failed\n", failures); return 1; }
    std::printf("ok\n");
    return 0;
}
