/* A client of the C ABI written in C (the language a cgo / Rust-FFI / ctypes binding sees): the sequence of calls
 * that stands for Renderer::new + draw (renderer.rs:41-123) — create, scene, camera, frame, render, resolve — on
 * the Cornell box at 64x36, 8 samples per pixel, through include/mirt.h and include/mirt_host.h only.
 * Built by tests/test_c_client.py with gcc -std=c99 -pedantic (the headers must be valid C) and linked against
 * ray_tracing_amd/libmirt.so; with the argument "run" it renders and prints a digest of the frame, without it
 * it only checks that every entry point it names resolves. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mirt.h"
#include "mirt_host.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != MIRT_OK) { fprintf(stderr, "%s -> %d: %s / %s\n", #call, rc_, ctx ? mirt_last_error(ctx) : "", mirt_host_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const uint32_t w = 64, h = 36, spp = 8, depth = 50;
    mirt_ctx* ctx = NULL;
    mirt_host_scene* scene = NULL;
    mirt_camera cam;
    mirt_render_params p;
    mirt_stats st;
    uint8_t* frame;
    unsigned long long digest = 1469598103934665603ull;
    size_t i, lit = 0;
    if (argc < 2 || strcmp(argv[1], "run") != 0) {
        /* no device needed: the scene mirror and the camera presets work on the host */
        CHECK(mirt_host_select_scene("cornell", 1, 11, NULL, 0, 0, &scene));
        CHECK(mirt_host_select_camera((float)w / (float)h, "cornell", &cam));
        printf("host ok: %u primitives, camera at (%g, %g, %g)\n", mirt_host_scene_desc(scene)->n_prims, cam.origin[0], cam.origin[1], cam.origin[2]);
        mirt_host_scene_free(scene);
        return 0;
    }
    CHECK(mirt_create(0, &ctx));
    CHECK(mirt_host_select_scene("cornell", 1, 11, NULL, 0, 0, &scene));
    CHECK(mirt_scene_upload(ctx, mirt_host_scene_desc(scene)));
    CHECK(mirt_host_select_camera((float)w / (float)h, "cornell", &cam));
    CHECK(mirt_camera_set(ctx, &cam));
    CHECK(mirt_frame_begin(ctx, w, h, NULL));
    memset(&p, 0, sizeof p);
    p.width = w; p.height = h; p.spp_begin = 0; p.spp_count = spp; p.max_depth = depth; p.seed = 3; p.jitter = 1; p.kernel = MIRT_KERNEL_AUTO;
    CHECK(mirt_render(ctx, &p));
    frame = (uint8_t*)malloc((size_t)w * h * 4);
    if (!frame) return 1;
    CHECK(mirt_resolve_rgba8(ctx, frame, (size_t)w * h * 4));
    CHECK(mirt_get_stats(ctx, &st));
    for (i = 0; i < (size_t)w * h * 4; ++i) {
        digest = (digest ^ frame[i]) * 1099511628211ull;
        if ((i & 3) != 3 && frame[i]) ++lit;
    }
    printf("frame digest %016llx lit %zu samples %llu segments %llu faults %u alpha %u\n", digest, lit, (unsigned long long)st.samples,
           (unsigned long long)st.segments, st.device_faults, (unsigned)frame[3]);
    free(frame);
    mirt_host_scene_free(scene);
    mirt_destroy(ctx);
    return 0;
}
