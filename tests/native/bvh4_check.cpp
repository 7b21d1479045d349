// Host-only check of the hierarchy the upload code builds (bvh_build.h): a binned-SAH binary tree collapsed into
// four-wide nodes. For object sets of several sizes and shapes it verifies
//   - every object sits in exactly one leaf, with the slot range the leaf reference names;
//   - every stored child box contains the boxes of all objects below it;
//   - children are packed from slot 0, missing ones carry no reference, pad[0] is their number;
//   - the stack need the builder reports equals the deepest stack a depth-first walk that enters every child
//     can reach (nearest first, the others pushed), and stays within the launch limit.
// Built and run by tests/test_hierarchy_build.py (g++, no GPU). Prints one line per case and "ok" at the end.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../ray_tracing_amd/csrc/bvh_build.h"

using namespace mirt;

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (failures++ < 20) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } } while (0)

struct Box { float lo[3], hi[3]; };

static bool leaf_ref(uint32_t r) { return (r & 0x80000000u) != 0; }

// Walks the subtree of four-wide node n; returns the union box of the objects below and the stack need of a walk
// that starts here with an empty stack.
static uint32_t check_node(const std::vector<DevBvhNode4>& nodes, uint32_t n, const std::vector<uint32_t>& slots, uint32_t ppo,
                           const std::vector<Box>& object_box_by_first_prim_div, std::vector<int>& seen, Box* out, int depth) {
    CHECK(n < nodes.size(), "node index %u out of range", n);
    CHECK(depth < 64, "tree too deep");
    const DevBvhNode4& nd = nodes[n];
    const uint32_t k = nd.pad[0];
    CHECK(k >= 1 && k <= 4, "child count %u", k);
    Box all{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
    uint32_t below = 0;
    for (uint32_t i = 0; i < 4; ++i) {
        Box stored{{nd.plane[0][i], nd.plane[2][i], nd.plane[4][i]}, {nd.plane[1][i], nd.plane[3][i], nd.plane[5][i]}};
        if (i >= k) {
            CHECK(nd.ref[i] == 0xffffffffu, "slot %u beyond the child count holds a reference", i);
            continue;
        }
        CHECK(nd.ref[i] != 0xffffffffu, "child %u of %u has no reference", i, k);
        Box sub{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
        if (leaf_ref(nd.ref[i])) {
            const uint32_t first = nd.ref[i] & 0x07ffffffu, count = ((nd.ref[i] >> 27) & 15u) + 1u;
            CHECK(count == ppo, "leaf holds %u records, objects have %u", count, ppo);
            CHECK(first + count <= slots.size(), "leaf slots out of range");
            const uint32_t obj = slots[first] / ppo;
            for (uint32_t j = 0; j < count; ++j) CHECK(slots[first + j] == slots[first] + j, "leaf records are not consecutive primitives");
            CHECK(obj < seen.size(), "object index out of range");
            if (obj < seen.size()) {
                ++seen[obj];
                sub = object_box_by_first_prim_div[obj];
            }
        } else {
            below = std::max(below, check_node(nodes, nd.ref[i], slots, ppo, object_box_by_first_prim_div, seen, &sub, depth + 1));
        }
        for (int a = 0; a < 3; ++a) {
            CHECK(stored.lo[a] <= sub.lo[a] && stored.hi[a] >= sub.hi[a], "child box does not contain its objects (node %u child %u axis %d)", n, i, a);
            all.lo[a] = std::min(all.lo[a], sub.lo[a]);
            all.hi[a] = std::max(all.hi[a], sub.hi[a]);
        }
    }
    const uint32_t need = below + (k - 1u);
    CHECK(nd.pad[1] == need, "node %u: stored need %u, walked %u", n, nd.pad[1], need);
    *out = all;
    return need;
}

static void run_case_with(const char* name, std::vector<BuildObject> objs, uint32_t ppo, bool by_area) {
    std::vector<Box> boxes(objs.size());
    for (size_t i = 0; i < objs.size(); ++i) {
        objs[i].first_prim = (uint32_t)i * ppo;
        for (int a = 0; a < 3; ++a) { boxes[i].lo[a] = objs[i].lo[a]; boxes[i].hi[a] = objs[i].hi[a]; }
    }
    std::vector<DevBvhNode> binary;
    std::vector<uint32_t> slots;
    BvhBuilder builder(binary, slots, kBvhStackDepth, ppo, 1e-4f);
    const uint32_t root2 = builder.build(objs);
    std::vector<DevBvhNode4> nodes(3);  // a tree appended behind other trees' nodes
    uint32_t need = 0;
    const uint32_t root4 = collapse_to_four(binary, root2, nodes, &need, by_area);
    CHECK(root4 == 3, "root index %u", root4);
    CHECK(slots.size() == boxes.size() * ppo, "slot count");
    std::vector<int> seen(boxes.size(), 0);
    Box all;
    const uint32_t walked = check_node(nodes, root4, slots, ppo, boxes, seen, &all, 0);
    CHECK(walked == need, "reported need %u, walked %u", need, walked);
    CHECK(need <= (uint32_t)kBvhStackMax, "need %u beyond the launch limit", need);
    CHECK(builder.deepest_leaf() - 1 <= kBvhStackDepth, "binary depth %d", builder.deepest_leaf());
    for (size_t i = 0; i < seen.size(); ++i) CHECK(seen[i] == 1, "object %zu is in %d leaves", i, seen[i]);
    // breadth-first renumbering together with a second, one-node tree (the three placeholder nodes in front become
    // it): the same checks hold from the new root, both roots lead the array, and a node's interior children never
    // come before it
    {
        nodes[0] = DevBvhNode4{};
        for (int c = 0; c < 4; ++c) nodes[0].ref[c] = 0xffffffffu;
        nodes[0].ref[0] = 0x80000000u;  // one leaf
        uint32_t other_root = 0, root = root4;
        std::vector<uint32_t*> roots{&other_root, &root};
        std::vector<DevBvhNode4> ordered(nodes.begin(), nodes.end());
        ordered[1] = ordered[2] = ordered[0];  // unreachable placeholders keep a place behind the others
        order_nodes_breadth_first(ordered, roots);
        CHECK(other_root == 0 && root == 1, "roots after renumbering: %u %u", other_root, root);
        for (uint32_t i = 0; i < ordered.size(); ++i)
            for (int c = 0; c < 4; ++c) {
                const uint32_t ref = ordered[i].ref[c];
                if (ref != 0xffffffffu && !(ref & 0x80000000u)) CHECK(ref > i && ref < ordered.size(), "node %u child %u out of order", i, ref);
            }
        std::vector<int> seen2(boxes.size(), 0);
        Box all2;
        const uint32_t walked2 = check_node(ordered, root, slots, ppo, boxes, seen2, &all2, 0);
        CHECK(walked2 == need, "need after renumbering %u", walked2);
        for (size_t i = 0; i < seen2.size(); ++i) CHECK(seen2[i] == 1, "after renumbering object %zu is in %d leaves", i, seen2[i]);
    }
    std::printf("%-28s %-8s objects %7zu  binary nodes %7zu  deepest leaf %2d  four-wide nodes %7zu  stack need %2u\n", name, by_area ? "by area" : "2 levels", boxes.size(),
                binary.size(), builder.deepest_leaf(), nodes.size() - 3, need);
}

// both rules of collapse_to_four: two binary levels per node, and the widest interior child first
static void run_case(const char* name, const std::vector<BuildObject>& objs, uint32_t ppo) {
    run_case_with(name, objs, ppo, false);
    run_case_with(name, objs, ppo, true);
}

static std::vector<BuildObject> random_objects(size_t n, uint32_t seed, int shape) {
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    std::vector<BuildObject> v(n);
    for (size_t i = 0; i < n; ++i) {
        float c[3], r;
        if (shape == 0) {          // uniform cloud of small spheres
            for (float& x : c) x = u(g) * 100.f;
            r = 0.1f + u(g);
        } else if (shape == 1) {   // a line of objects (worst case for splits on the other axes), sizes over four decades
            c[0] = (float)i; c[1] = 0.f; c[2] = 0.f;
            r = std::pow(10.f, u(g) * 4.f - 2.f);
        } else if (shape == 2) {   // all objects identical: no split plane separates them, the median fallback must
            c[0] = c[1] = c[2] = 1.f;
            r = 1.f;
        } else {                   // a grid with one huge object over everything (the ground sphere of scene.rs:46-52)
            c[0] = (float)(i % 100); c[1] = 0.2f; c[2] = (float)(i / 100);
            r = i == 0 ? 1000.f : 0.2f;
            if (i == 0) c[1] = -1000.f;
        }
        for (int a = 0; a < 3; ++a) { v[i].lo[a] = c[a] - r; v[i].hi[a] = c[a] + r; }
    }
    return v;
}

int main() {
    for (size_t n : {1, 2, 3, 4, 5, 7, 16, 17, 485, 4096}) {
        char name[64];
        std::snprintf(name, sizeof name, "cloud %zu", n);
        run_case(name, random_objects(n, (uint32_t)n, 0), 1);
    }
    run_case("cloud 200000 (threaded)", random_objects(200000, 9, 0), 1);
    run_case("line 3000", random_objects(3000, 1, 1), 1);
    run_case("identical 1000", random_objects(1000, 2, 2), 1);
    run_case("grid 10000 + ground", random_objects(10000, 3, 3), 1);
    run_case("cuboids as 6 records", random_objects(400, 4, 0), 6);
    if (failures) {
        std::printf("%d checks failed\n", failures);
        return 1;
    }
    std::printf("ok\n");
    return 0;
}
