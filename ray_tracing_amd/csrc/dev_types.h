// Device-resident scene layout. Built once by the upload code from the mirt_scene_desc arrays and
// passed to every kernel by value (it lands in SGPRs / the kernarg segment). Records are multiples
// of 16 bytes: list scans fetch them with scalar loads (wave-uniform index, data stays in SGPRs),
// BVH leaves and shading fetch them with vector / LDS loads.
#pragma once
#include <stdint.h>

namespace mirt {

// meta word of DevPrim
enum : uint32_t {
    PK_SPHERE = 0, PK_RECT = 1, PK_MEDIUM = 2,
    PM_KIND_MASK = 3u,
    PM_AXIS_SHIFT = 2, PM_AXIS_MASK = 3u << 2,
    PM_FLIP = 1u << 4,
    PM_MOVING = 1u << 5,
};

// Shape of a medium's boundary list, found at upload: one static sphere, or the three rect-pair scan records (XY, XZ,
// YZ, one instance) of a cuboid, let ConstantMedium::hit take both crossings from one evaluation of the boundary.
enum : uint32_t { MS_LIST = 0, MS_SPHERE = 1, MS_CUBOID = 2 };

struct __attribute__((aligned(16))) DevPrim {
    // sphere: a = (center0.xyz, radius)        b = (center1.xyz, time0)     c0 = time1
    // rect:   a = (k, lo0, hi0, lo1)           b = (hi1, k_mid, 0, 0)
    // medium: a = (bits first boundary scan record, bits record count, neg_inv_density, bits medium_index)
    //         b[0] = bits MS_* (shape of the boundary list)
    float a[4];
    float b[4];
    float c0;
    uint32_t material;
    uint32_t instance;
    uint32_t meta;
};
static_assert(sizeof(DevPrim) == 48, "DevPrim is three 16-byte quads");

// One primitive of a hierarchy leaf (32 bytes, two 16-byte vector loads per lane).
//   rect:   f = (k, lo0, hi0, lo1, hi1, -)      sphere: f = (center.xyz, radius, -, -)
//   moving sphere: f = (center0.xyz, radius, -, -), the rest in DevLeafMoving (list scans read prims[prim])
enum : uint32_t {
    SC_SPHERE = 0, SC_RECT_XY = 1, SC_RECT_XZ = 2, SC_RECT_YZ = 3, SC_MEDIUM = 4, SC_SPHERE_MOVING = 5, SC_END = 6,
    SC_RECT2_XY = 7, SC_RECT2_XZ = 8, SC_RECT2_YZ = 9,  // two consecutive rects of one axis (scan records only)
    SC_CUBOID = 10,  // the six rects of a cuboid, f = (lo.xyz, hi.xyz), ids prim .. prim + 5 (leaf records only)
};
struct __attribute__((aligned(32))) DevLeafRec {
    uint32_t code_inst;  // code | instance << 8
    uint32_t prim;       // index into prims (the id reported for a hit)
    float f[6];
};
static_assert(sizeof(DevLeafRec) == 32, "DevLeafRec is two 16-byte loads");
// What a moving sphere's leaf record needs beside its first centre and radius (geo.rs:25-33), at the record's index in
// an array of its own (DevScene::bvh_leaf_moving; only scenes with a moving sphere in a hierarchy have the array): the
// record and this are fetched together, where reading the primitive through the record's index took three dependent
// fetches more.
struct __attribute__((aligned(32))) DevLeafMoving {
    float center1[3];
    float time0, time1;
    float pad[3];
};
static_assert(sizeof(DevLeafMoving) == 32, "DevLeafMoving is two 16-byte loads");

// One element of a list scan (64 bytes = one s_load_dwordx16; the scan index is wave-uniform, so the
// record stays in scalar registers). A list (world entry or the boundary of a medium) is a contiguous
// range of these in list order; the array ends with one SC_END record. A record holds one primitive, or
// two consecutive rects of the same axis and instance
// (the opposite faces of a cuboid, a pair of walls): their plane distances and hit points are computed
// with packed two-float arithmetic, so the values of the two rects are interleaved.
//   rect:       f = (k, lo0, hi0, lo1, hi1)       sphere: f = (center.xyz, radius)
//   rect pair:  f = (k_a, k_b, lo0_a, lo0_b, hi0_a, hi0_b, lo1_a, lo1_b, hi1_a, hi1_b); ids prim, prim2
//   moving sphere: f = (center0.xyz, radius, center1.xyz, time0, time1)
//   medium: f[0..3] = its boundary sphere (when the boundary is one static sphere), f[4] = -1 / density, and as bits
//           f[5] = medium index, f[6] = MS_* shape of the boundary, f[7] = the boundary sphere's instance,
//           f[8], f[9] = first record and number of records of the boundary list
struct __attribute__((aligned(64))) DevScanRec {
    uint32_t code_inst;  // code | instance << 8
    uint32_t prim;       // index into prims (the id reported for a hit)
    uint32_t prim2;      // second rect of a pair
    uint32_t pad;
    float f[12];
};
static_assert(sizeof(DevScanRec) == 64, "DevScanRec is one 64-byte scalar load");

// What shading needs to know about the material of a primitive, resolved at upload so that a hit
// costs one table read: material kind, its parameter, and — for Metal, or for a texture that is a
// solid colour — the colour itself. Other textures are evaluated through `texture`.
struct __attribute__((aligned(16))) DevShade {
    uint32_t mat_kind;
    uint32_t tex_kind;
    uint32_t texture;
    float param;
    float color[4];
};

struct __attribute__((aligned(16))) DevInstance {
    float sin_t, cos_t, ox, oy;
    float oz, pad0, pad1, pad2;
};

// What a walk needs to start hierarchy k of the scene (world order): its root node and the frame of its entry
// (`instance` = 0: none), in one 32-byte record — through bvh_entries, entries and instances it was three dependent
// fetches at the start of every hierarchy, and again whenever a walk continued in a later iteration.
struct __attribute__((aligned(32))) DevBvhStart {
    float sin_t, cos_t, ox, oy, oz;
    uint32_t root;
    uint32_t instance;
    uint32_t pad;
};
static_assert(sizeof(DevBvhStart) == 32, "DevBvhStart is two 16-byte loads");

struct __attribute__((aligned(16))) DevTexture {
    uint32_t kind, perlin, image, pad;
    float c0[4];  // solid colour / checker even
    float c1[4];  // checker odd; c1[3] = noise scale
};

struct DevImage {
    uint32_t width, height;
    const uint8_t* rgba;  // device pointer
};

struct __attribute__((aligned(16))) DevPerlin {
    float grad[256][4];
    uint8_t perm[3][256];
};

// Depth bound of the binary trees the builder makes (root node = level 1) before they are collapsed into four-wide
// nodes, and the largest per-lane traversal stack (LDS, [level][lane]) a kernel is launched with. A walk pushes up
// to three references per four-wide node on its way down; the upload code computes the exact bound of every tree
// (DevScene::stack_limit, which sizes the kernels' dynamic LDS) and refuses a scene that exceeds kBvhStackMax.
constexpr int kBvhStackDepth = 32;
constexpr int kBvhStackMax = 64;
// Levels of a launched stack beyond DevScene::stack_limit: the three pushes of a node step store before the limit is
// checked (walk_push in dev_core.h).
constexpr uint32_t kBvhStackSlack = 3;

enum : uint32_t { EK_LIST = 0, EK_BVH = 1 };
struct __attribute__((aligned(16))) DevEntry {
    uint32_t kind;
    uint32_t first_prim;
    uint32_t prim_count;
    uint32_t prims_per_object;
    uint32_t instance;
    uint32_t bvh_root;    // EK_BVH: index of the root node in bvh_nodes
    uint32_t scan_first;  // EK_LIST: the entry's scan records
    uint32_t scan_count;
};

// The builder's 64-byte binary node: the boxes of both children and their references (host side only; the device
// walks the four-wide nodes below, which are collapsed from these).
// ref_i is the reference a walk continues with when it enters child i: an interior node index, or
// bit 31 | (records - 1) << 27 | first record for a leaf of up to 16 primitive records in bvh_leaves.
// A missing second child (single-leaf tree) has an empty box (lo = +inf, hi = -inf).
struct __attribute__((aligned(16))) DevBvhNode {
    float lo0x, lo0y, hi0x, hi0y;
    float lo1x, lo1y, hi1x, hi1y;
    float lo0z, hi0z, lo1z, hi1z;
    uint32_t ref0, ref1, pad0, pad1;
};
static_assert(sizeof(DevBvhNode) == 64, "DevBvhNode is four 16-byte quads");

// 128-byte node (one cache line, seven 16-byte loads) holding the boxes of up to four children: a binary node with
// its interior children replaced by THEIR children, so a walk takes one step, one dependent fetch, where the binary
// tree took two or three. The coordinates are stored by plane: plane[2 a] = the low bounds of the four children on
// axis a, plane[2 a + 1] = their high bounds. A lane reads, per axis, the quad its ray enters through and the quad it
// leaves through — which is which depends only on the sign of the ray's direction on that axis, so it is an address
// offset (0 or 16 bytes) fixed for the whole walk — and has the near and far distances of all four children without
// a single min / max to tell them apart.
// A missing child is the point (+inf, +inf, +inf) with the reference 0xffffffff: no ray with a finite far bound meets it.
struct __attribute__((aligned(128))) DevBvhNode4 {
    float plane[6][4];
    uint32_t ref[4];
    uint32_t pad[4];
};
static_assert(sizeof(DevBvhNode4) == 128, "DevBvhNode4 is one 128-byte line");

enum : uint32_t { MK_LAMBERTIAN = 0, MK_METAL = 1, MK_DIELECTRIC = 2, MK_DIFFUSE_LIGHT = 3, MK_ISOTROPIC = 4 };
enum : uint32_t { TK_SOLID = 0, TK_CHECKER = 1, TK_NOISE = 2, TK_IMAGE = 3 };

struct DevCamera {
    float origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3];
    float lens_radius, time0, time1;
};

// feature bits: which code the scene needs (selects the kernel specialisation)
enum : uint32_t {
    F_SPHERE = 1u << 0, F_RECT = 1u << 1, F_MEDIUM = 1u << 2, F_INSTANCE = 1u << 3, F_MOVING = 1u << 4,
    F_BVH = 1u << 5, F_TEX_CHECKER = 1u << 6, F_TEX_NOISE = 1u << 7, F_TEX_IMAGE = 1u << 8,
    F_MAT_METAL = 1u << 9, F_MAT_DIELECTRIC = 1u << 10, F_MAT_ISOTROPIC = 1u << 11, F_MAT_LIGHT = 1u << 12,
    F_ALL = 0x1fffu,
};

struct DevScene {
    const DevPrim* prims;         // n_prims + 1 records
    const DevShade* shade;        // per primitive
    const DevScanRec* scan;       // list-scan records, SC_END terminated
    const DevEntry* entries;
    const DevPrim* light_prims;   // copies of the light primitives, in light order
    const DevTexture* textures;
    const DevImage* images;
    const DevPerlin* perlins;
    const DevInstance* instances;
    const DevBvhNode4* bvh_nodes;
    const DevLeafRec* bvh_leaves;  // leaf primitive records in tree order
    const DevLeafMoving* bvh_leaf_moving;  // beside them, or nullptr when no leaf record is a moving sphere
    const uint32_t* bvh_entries;   // indices into entries of the EK_BVH ones, in world order
    const DevBvhStart* bvh_starts; // one per hierarchy, same order
    // World-space boxes of the hierarchies that are small against the scene (two float4 per hierarchy, bvh_entries
    // order: low corner with a flag in .w — 1 = test this box before starting the hierarchy — and high corner), padded;
    // nullptr when no hierarchy is flagged. A ray that misses the box cannot hit a member, so its walk skips the
    // hierarchy without the change of frame and the root step (walk_skip_missed).
    const float* bvh_bounds;
    uint32_t n_bvh_entries;
    uint32_t n_prims, n_entries, n_lights, n_media;
    uint32_t n_instances, n_textures, n_perlins;
    uint32_t features;
    float background[3];
    // Device-side diagnostics. `fault` points at one word the kernels OR bits into when they meet a state the host
    // promised could not occur (FAULT_*); the host reads it back with the statistics and in the blocking resolve.
    // stack_limit = entries of the per-lane traversal stack the scene's deepest walk can need (the kernels' LDS stacks
    // are launched with that many levels; a test may lower it to see the overflow report).
    uint32_t* fault;
    uint32_t stack_limit;
    // Nodes [0, lds_nodes) of bvh_nodes are also held in the workgroup's LDS (set by the launch of a kernel that copies
    // them there; 0 everywhere else): the array is ordered breadth first from the roots, so these are the nodes
    // nearest the roots, which every walk visits.
    uint32_t lds_nodes;
};
// A node in LDS takes kLdsNodeStride bytes, 16 more than its 128: with a stride of 128 every node would start in the
// same bank and the lanes of a wave, each reading the same quad of a different node, would all meet in four banks.
constexpr uint32_t kLdsNodeStride = 144;
enum : uint32_t {
    FAULT_STACK_OVERFLOW = 1u,  // a walk pushed beyond its stack: the children that did not fit were dropped
    FAULT_STALE_CHUNK = 2u,     // a (pixel, chunk) slot did not carry the stamp of its launch: its item was never stored
    FAULT_POOL_STALL = 4u,      // a workgroup of the pooled-walk kernel gave up after 2^20 iterations: a ray was lost in its pool
};

}  // namespace mirt
