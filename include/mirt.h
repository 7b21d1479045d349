/*
 * mirt.h — C ABI of libmirt.so, the MI355X (gfx950) path-tracing backend.
 *
 * The reference crate (KaminariOS/ray_tracing) has no FFI; its seam for the per-pixel sample loop
 * is Renderer::new (src/renderer.rs:41) / Renderer::draw (src/renderer.rs:56). The entry points
 * below are what a Rust `extern "C"` block bound to that seam would declare: the caller flattens
 * its Scene (src/scene.rs:277-295) into plain arrays once, hands over its Camera fields
 * (src/camera.rs:5-20), and replaces the body of draw() with render + resolve.
 *
 * Conventions: every function returns MIRT_OK (0) or a negative MIRT_ERR_* code and never throws or
 * aborts across the boundary; mirt_last_error() returns the message of the last failure on that
 * context (or of the last failed mirt_create when ctx is NULL). One host thread per context at a
 * time (the reference's `&mut self`); contexts on different devices are independent. All pointers
 * in descriptors are host pointers unless a field says "device"; the library copies what it needs
 * during the call and the caller keeps ownership.
 */
#ifndef MIRT_H
#define MIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRT_OK 0
#define MIRT_ERR_INVALID (-1)  /* bad argument or descriptor */
#define MIRT_ERR_HIP (-2)      /* a HIP runtime call failed */
#define MIRT_ERR_STATE (-3)    /* call order: no scene / camera / frame yet */
#define MIRT_ERR_NO_DEVICE (-4)
#define MIRT_ERR_DEVICE (-5)   /* the kernels reported an internal fault (mirt_stats.device_faults) */
#define MIRT_ERR_COMM (-6)     /* an RCCL call failed, or librccl could not be loaded */

typedef struct mirt_ctx mirt_ctx;

/* ---- scene description: the flattened form of src/scene.rs `Scene` ------------------------- */

/* Leaf primitives (src/geo.rs). A Cuboid (geo.rs:245) is six rects in the order XY(kmin,kmax),
 * XZ(kmin,kmax), YZ(kmin,kmax); FlipFace (geo.rs:442) is a flag; Translation(RotationY(x))
 * (geo.rs:291,332) is an instance index; ConstantMedium (geo.rs:393) refers to a range of
 * boundary primitives that are not themselves part of the world. */
enum { MIRT_PRIM_SPHERE = 0, MIRT_PRIM_RECT = 1, MIRT_PRIM_MEDIUM = 2 };
enum { MIRT_AXIS_XY = 0, MIRT_AXIS_XZ = 1, MIRT_AXIS_YZ = 2 }; /* AlignedAxis, geo.rs:152-167 */
enum { MIRT_PRIM_FLIP_FACE = 1u, MIRT_PRIM_MOVING = 2u };

typedef struct mirt_prim {
    uint32_t kind;     /* MIRT_PRIM_* */
    uint32_t material; /* index into materials; for a medium: its Isotropic phase material */
    uint32_t instance; /* index into instances; 0 = no transform */
    uint32_t flags;    /* MIRT_PRIM_FLIP_FACE | MIRT_PRIM_MOVING */
    /* sphere (geo.rs:14-23) */
    float center0[3];
    float radius;
    float center1[3];
    float time0, time1;
    /* rect (geo.rs:168-174): plane coordinate k on the axis' third index, extents on the first
     * two (c0 = lo, c1 = hi of AxisAlignedRect::new) */
    uint32_t axis;
    float k;
    float lo[2];
    float hi[2];
    /* medium (geo.rs:393-397) */
    uint32_t boundary_first; /* first boundary primitive */
    uint32_t boundary_count;
    float neg_inv_density;
} mirt_prim;

/* Translation(RotationY(obj, degrees), offset): sin/cos of degrees/180*pi (camera.rs:25-27). */
typedef struct mirt_instance {
    float sin_theta, cos_theta;
    float offset[3];
} mirt_instance;

enum { MIRT_MAT_LAMBERTIAN = 0, MIRT_MAT_METAL = 1, MIRT_MAT_DIELECTRIC = 2, MIRT_MAT_DIFFUSE_LIGHT = 3, MIRT_MAT_ISOTROPIC = 4 };
typedef struct mirt_material { /* src/material.rs */
    uint32_t kind;
    uint32_t texture; /* Lambertian / DiffuseLight / Isotropic */
    float albedo[3];  /* Metal */
    float param;      /* Metal: fuzz (already min(fuzz,1)); Dielectric: index of refraction */
} mirt_material;

enum { MIRT_TEX_SOLID = 0, MIRT_TEX_CHECKER = 1, MIRT_TEX_NOISE = 2, MIRT_TEX_IMAGE = 3 };
typedef struct mirt_texture { /* src/texture.rs */
    uint32_t kind;
    float color[3];  /* Solid colour; Checker `even` */
    float color2[3]; /* Checker `odd` */
    float scale;     /* Noise */
    uint32_t perlin; /* Noise: index into perlins */
    uint32_t image;  /* Image: index into images */
} mirt_texture;

typedef struct mirt_perlin { /* src/perlin.rs:8-11 */
    float grad[256][3];
    uint8_t perm[3][256];
} mirt_perlin;

/* Texels exactly as ImageTexture indexes them (texture.rs:93-97): row y = v*h of the image AFTER
 * the vertical flip of texture.rs:84. width == 0 means "file missing" and yields (0,1,1). */
typedef struct mirt_image {
    uint32_t width, height;
    const uint8_t* rgba;
} mirt_image;

/* One element of the world list (scene.rs:290), in order. The list semantics of ray.rs:123-133
 * (of two hits at the same t the later element wins) are kept for MIRT_ENTRY_LIST entries. A
 * MIRT_ENTRY_BVH entry is the flattened form of BVHNode (aabb.rs:63): the library builds a linear
 * node array over its objects at upload. Object j of an entry is the `prims_per_object` primitives
 * starting at first_prim + j*prims_per_object (6 for cuboids, 1 otherwise). `instance` applies to
 * the whole entry (Translation(RotationY(BVHNode))). */
enum { MIRT_ENTRY_LIST = 0, MIRT_ENTRY_BVH = 1 };
typedef struct mirt_entry {
    uint32_t kind;
    uint32_t first_prim;
    uint32_t prim_count;
    uint32_t prims_per_object;
    uint32_t instance;
} mirt_entry;

typedef struct mirt_scene_desc {
    const mirt_prim* prims;         uint32_t n_prims;
    const mirt_entry* world;        uint32_t n_world;
    const uint32_t* lights;         uint32_t n_lights; /* primitive indices of scene.lights (scene.rs:286) */
    const mirt_material* materials; uint32_t n_materials;
    const mirt_texture* textures;   uint32_t n_textures;
    const mirt_perlin* perlins;     uint32_t n_perlins;
    const mirt_image* images;       uint32_t n_images;
    const mirt_instance* instances; uint32_t n_instances; /* entry 0 is the identity and is never applied */
    float background[3];            /* scene.rs:280 */
} mirt_scene_desc;

/* src/camera.rs:5-20, the fields get_ray (camera.rs:86-94) reads. */
typedef struct mirt_camera {
    float origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3];
    float lens_radius, time0, time1;
} mirt_camera;

typedef struct mirt_ray { float origin[3], direction[3], time; } mirt_ray;

/* Which form of the sample loop renders: FUSED = one launch, one pixel per lane for the whole call; QUEUE = one
 * persistent launch whose lanes draw (pixel, chunk of samples) items from a shared counter, so that no lane waits
 * for a tile's or the frame's most expensive pixel, followed by a pass that adds the chunk sums of every pixel in
 * chunk order; WAVEFRONT = generate / extend / shade launches per ray segment with queues in memory (kept as the
 * comparison baseline). AUTO takes QUEUE. All forms render the same samples; QUEUE and FUSED differ only in the
 * grouping of the float additions of a pixel's samples (none when spp_count is 1). */
enum { MIRT_KERNEL_AUTO = 0, MIRT_KERNEL_FUSED = 1, MIRT_KERNEL_WAVEFRONT = 2, MIRT_KERNEL_QUEUE = 3 };
/* OR-ed into mirt_render_params.kernel: the persistent kernel and the wavefront walk kernel also tally what their
 * hierarchy walks cost (mirt_stats.bvh_*). Tallying slows the launch by a few percent: for measurement passes, not
 * for timed ones. */
#define MIRT_KERNEL_COUNT_WALKS 0x100u
/* OR-ed into MIRT_KERNEL_WAVEFRONT: after the intersection stages of a round the paths are filed into one queue per
 * class of what they do next — the ScatterType of material.rs:19-23 (Diffuse, Specular, ISO) and the paths that end
 * (miss, emitter) — and the shading stage runs once per class. Same samples, same sums. */
#define MIRT_KERNEL_WF_CLASSES 0x200u
/* OR-ed into MIRT_KERNEL_AUTO / MIRT_KERNEL_QUEUE: in scenes with hierarchies the persistent kernel pools the walks of
 * a workgroup's 256 lanes in LDS and its four waves advance them together — one step per ray and round, the rays that
 * need a node step in one list, those that need a leaf test in another, 64 of a kind per wave — instead of every lane
 * walking its own ray (aabb.rs:112-126 is the step either way). Same walks, same samples, same sums. Ignored by
 * scenes without a hierarchy and by launches that carry MIRT_KERNEL_COUNT_WALKS. */
#define MIRT_KERNEL_POOL 0x400u

typedef struct mirt_render_params {
    uint32_t width, height;
    uint32_t spp_begin;  /* global index of the first sample of this call */
    uint32_t spp_count;  /* samples per pixel rendered by this call */
    uint32_t max_depth;  /* Renderer::max_depth, renderer.rs:35 */
    uint32_t seed;
    uint32_t jitter;     /* 0 = no pixel jitter (the multisample == 1 case of renderer.rs:175) */
    uint32_t kernel;     /* MIRT_KERNEL_* */
} mirt_render_params;

typedef struct mirt_stats {
    uint64_t samples;      /* primary paths since the last mirt_frame_begin */
    uint64_t segments;     /* ray segments traced (ray_color levels that reached world.hit) */
    double kernel_ms;      /* GPU time of the last mirt_render, HIP events on the render stream */
    uint32_t kernel_launches; /* launches in the last mirt_render */
    uint32_t kernel_used;  /* MIRT_KERNEL_FUSED, MIRT_KERNEL_QUEUE or MIRT_KERNEL_WAVEFRONT */
    uint64_t lane_slots;   /* fused kernel: 64 x loop iterations summed over waves; segments / lane_slots =
                              share of lane-iterations that traced a segment (the rest idled in a wave whose
                              other lanes had longer paths) */
    /* Algorithmic work of the hierarchy walks since mirt_frame_begin, tallied by launches that carry
     * MIRT_KERNEL_COUNT_WALKS, per lane that took the step: four-wide interior nodes visited (bvh_box_tests below
     * holds their child boxes, two to four each), leaf spheres and leaf rects intersected (a cuboid is six). With
     * the list_* fields — what one scan of the world's list entries tests, i.e. per segment — this prices a
     * segment: 24 FLOP per box, 30 per sphere, 12 per rect, 15 per instance transform. */
    uint64_t bvh_node_steps, bvh_sphere_tests, bvh_rect_tests;
    uint32_t list_rects, list_spheres, list_transforms;
    uint32_t device_faults;    /* MIRT_FAULT_* bits the kernels raised since mirt_frame_begin; 0 in a correct run */
    uint32_t bvh_deepest_leaf; /* level of the deepest leaf of the scene's binary hierarchies before they are
                                  collapsed into four-wide nodes (root node = 1; the builder stops at level 33) */
    uint32_t bvh_stack_entries; /* entries of the per-lane traversal stack the kernels are launched with: the exact
                                  need of the scene's deepest four-wide hierarchy (at most 64, or the upload fails) */
    uint64_t bvh_box_tests;    /* child boxes of the nodes counted in bvh_node_steps */
    double reduce_ms;          /* GPU time of this context's part in the last mirt_comm_reduce / mirt_reduce since
                                  mirt_frame_begin, HIP events on the render stream (it includes the wait for the
                                  slowest rank's render); 0 when none ran */
} mirt_stats;
#define MIRT_FAULT_STACK_OVERFLOW 1u /* a hierarchy walk found its traversal stack full and dropped a subtree */
#define MIRT_FAULT_STALE_CHUNK 2u    /* a (pixel, sample chunk) work item of the persistent kernel was never stored */
#define MIRT_FAULT_POOL_STALL 4u     /* a workgroup of the pooled-walk form gave up after 2^20 iterations (a ray lost in its pool) */

/* ---- entry points --------------------------------------------------------------------------- */

/* Process start: bind a context to HIP device `device`. */
int mirt_create(int device, mirt_ctx** out);
void mirt_destroy(mirt_ctx* ctx);
const char* mirt_last_error(const mirt_ctx* ctx);

/* Renderer::new's `scene` argument (renderer.rs:41): flatten, build BVHs, upload once. */
int mirt_scene_upload(mirt_ctx* ctx, const mirt_scene_desc* desc);
/* Renderer::new's `camera` argument; also the camera swap of renderer.rs:193. */
int mirt_camera_set(mirt_ctx* ctx, const mirt_camera* cam);

/* Start a frame: size and zero the float4 accumulation buffer (sum r, g, b, sample count) of
 * width*height pixels, pixel index y*width + x with y = 0 at the image BOTTOM (renderer.rs:97).
 * accum_device may be a caller-owned DEVICE buffer of width*height*4 floats (e.g. a torch tensor
 * that a collective will reduce) or NULL for a library-owned one. */
int mirt_frame_begin(mirt_ctx* ctx, uint32_t width, uint32_t height, float* accum_device);

/* The spp loop of renderer.rs:100-110 for samples [spp_begin, spp_begin+spp_count): each sample
 * is sanitised (NaN -> 0, +-inf -> 1, renderer.rs:107) and added to the accumulation buffer.
 * Asynchronous on the context's stream. Sample ranges make progressive refinement and the
 * multi-GPU split the same call. */
int mirt_render(mirt_ctx* ctx, const mirt_render_params* p);

/* mirt_render for callers that cannot allocate device memory — code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc. It has no counterpart in the reference. It renders the same samples with the
 * persistent kernel (kernel = MIRT_KERNEL_AUTO or MIRT_KERNEL_QUEUE), but the kernel's item counters and chunk sums
 * live in a static device array of the library (64 MiB + 4 KiB, held on every GPU the process uses the library on
 * until the process ends, whether or not this entry point is ever called) instead of buffers grown on demand, and
 * the call allocates and frees nothing. Frames of more than MIRT_STATIC_MAX_PIXELS pixels or 65535 in one dimension
 * are rejected. A frame that fits sixteen chunk images in the array (up to 256 Ki pixels) gets the sums of
 * mirt_render bit for bit; a larger one gets fewer, longer chunks (1920x1080: two), and one whose two chunk images
 * do not fit is rendered with whole pixels as work items — other groupings of the same float additions, and a
 * longer tail of the launch. All calls on one GPU share that GPU's array: calls that overlap on one GPU (two
 * contexts, two streams) race; one at a time is the contract. A captured HIP graph replays the stamp of its
 * capture, so the report of never-stored work items (MIRT_FAULT_STALE_CHUNK) is blind across replays. The frame
 * (mirt_frame_begin, itself allocation-free with a caller-owned accum_device), scene and camera are set up before. */
#define MIRT_STATIC_MAX_PIXELS 8388608u
int mirt_render_static(mirt_ctx* ctx, const mirt_render_params* p);

/* renderer.rs:109-116 and the row flip of :97: mean, clamp [0,0.999], sqrt, *256, truncate; alpha
 * 0xff; top row first. frame is a HOST buffer of len == width*height*4 bytes (checked, mirrors the
 * assert of renderer.rs:65). Blocks until done. Returns MIRT_ERR_DEVICE (frame still written) when the
 * kernels raised a fault bit since mirt_frame_begin — the reference would have panicked. */
int mirt_resolve_rgba8(mirt_ctx* ctx, uint8_t* frame, size_t len);
/* The same into a DEVICE buffer, asynchronous on the context's stream. frame_device may be NULL: the frame then
 * stays in a device buffer the library keeps (a viewer that only presents it, a benchmark). */
int mirt_resolve_rgba8_device(mirt_ctx* ctx, uint8_t* frame_device, size_t len);

/* Copy the accumulation buffer to the host (n == width*height*4 floats). Blocks. */
int mirt_read_accum(mirt_ctx* ctx, float* rgba, size_t n);
/* Replace the accumulation buffer's contents from the host (n == width*height*4 floats): resume of an
 * interrupted progressive render from a saved mirt_read_accum() image; continue with spp_begin = the
 * sample count it holds. Blocks. No counterpart in the reference, which restarts every draw. */
int mirt_write_accum(mirt_ctx* ctx, const float* rgba, size_t n);
/* Device address of the accumulation buffer in use. */
int mirt_accum_device_ptr(mirt_ctx* ctx, float** out);

/* world.hit(ray, 0.001, inf) (renderer.rs:201) for n host rays: leaf primitive index (-1 = miss)
 * and t. Ray i draws its constant-medium free paths from (pixel = i, sample = 0, bounce = 0).
 * aov may be NULL or n*12 floats: point 3, normal 3, front_face, u, v, material colour 3. */
int mirt_first_hit(mirt_ctx* ctx, const mirt_ray* rays, size_t n, uint32_t seed, int32_t* prim_id, float* t, float* aov);

/* ray_color (renderer.rs:197) for n host rays, one un-sanitised sample each: ray i uses
 * (pixel = i, sample). out = n*3 floats. */
int mirt_ray_color(mirt_ctx* ctx, const mirt_ray* rays, size_t n, uint32_t sample, uint32_t max_depth, uint32_t seed, float* out);

/* ---- the exchange step of the multi-GPU split --------------------------------------------------
 * The reference keeps its parallelism inside draw() (rayon tasks over rows, renderer.rs:69-72). With GPUs as the
 * workers the split is by sample range: context g renders samples [g*S, (g+1)*S) of every pixel (spp_begin /
 * spp_count of mirt_render) into its own accumulation buffer, and ONE sum of those buffers onto the root follows —
 * an RCCL ncclReduce(sum, float, width*height*4) over xGMI, enqueued on each context's own stream behind its
 * render (asynchronous; resolve on the root afterwards). librccl is loaded on first use.
 *
 * One process, n contexts on n different devices, every frame begun with the same size: */
int mirt_reduce(mirt_ctx** ctxs, int n, int root);
/* One process per GPU: some rank makes an id, the caller carries its bytes to all ranks by whatever channel it
 * has (MPI, torch.distributed, a file), every rank joins with it, then every rank calls mirt_comm_reduce. */
#define MIRT_COMM_ID_BYTES 128
int mirt_comm_unique_id(uint8_t id[MIRT_COMM_ID_BYTES]);
int mirt_comm_init_rank(mirt_ctx* ctx, int n_ranks, int rank, const uint8_t id[MIRT_COMM_ID_BYTES]);
int mirt_comm_reduce(mirt_ctx* ctx, int root);
int mirt_comm_ranks(const mirt_ctx* ctx); /* ranks of the communicator the context belongs to, 0 = none */

int mirt_synchronize(mirt_ctx* ctx);
int mirt_get_stats(mirt_ctx* ctx, mirt_stats* out);

/* Philox4x32-10 block, the generator behind every get_rand() (rand_gen.rs:7). Host-side. */
void mirt_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_H */
