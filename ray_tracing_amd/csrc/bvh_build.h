// Host-side flattening of the reference's pointer BVH (aabb.rs:63-135): a linear array of 64-byte two-child
// nodes first (BvhBuilder), then the 128-byte four-wide nodes the kernels walk (collapse_to_four, at the end of
// this file). The tree SHAPE is this library's own (binned surface-area heuristic,
// median fallback, bounded depth); the reference picks a random axis per node (aabb.rs:73), so no
// particular shape is part of its behaviour — any conservative hierarchy returns the same closest
// hit (ties between equal-t leaves aside, which the reference resolves by tree shape too).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>
#include "dev_types.h"

namespace mirt {

struct BuildObject {
    float lo[3], hi[3];
    uint32_t first_prim;
};

class BvhBuilder {
   public:
    // One object per leaf; an object is prims_per_object consecutive primitives. max_depth bounds the traversal
    // stack. `slots` receives the primitive indices of the leaves in tree order; a leaf reference is (first slot,
    // number of slots). A hierarchy over n > 1 objects has exactly n - 1 nodes, numbered in pre-order, so every
    // subtree knows its node and slot ranges before it is built and large subtrees are built by separate threads
    // (same result as a serial build; 1 M spheres: 0.95 s -> 0.21 s on 8 cores).
    // `pad` is added to every stored child box on all sides (see walk_node: it covers the rounding of the fused
    // plane-distance evaluation near the ray origin).
    BvhBuilder(std::vector<DevBvhNode>& nodes, std::vector<uint32_t>& slots, int max_depth, uint32_t prims_per_object, float pad = 0.f)
        : nodes_(nodes), slots_(slots), max_depth_(max_depth), ppo_(prims_per_object), pad_(pad) {}

    // Returns the index of the root node.
    uint32_t build(std::vector<BuildObject>& objs) {
        const size_t n = objs.size(), node0 = nodes_.size(), slot0 = slots_.size();
        slots_.resize(slot0 + n * ppo_);
        if (is_leaf(n)) {
            // the whole entry is one leaf: root with a single child
            Ref r = recurse(objs, 0, n, 1, node0, slot0, 0);
            DevBvhNode nd{};
            set_child(nd, 0, r);
            Ref none;
            none.c = 0x7ffffffu;  // with n = 16 the reference is 0xffffffff, "no reference"
            none.n = 16;
            for (int a = 0; a < 3; ++a) { none.lo[a] = INFINITY; none.hi[a] = -INFINITY; }
            set_child(nd, 1, none);
            nodes_.push_back(nd);
            return (uint32_t)node0;
        }
        nodes_.resize(node0 + n - 1);
        unsigned hw = std::thread::hardware_concurrency();
        int fork_levels = 0;
        while ((1u << fork_levels) < std::min(hw ? hw : 1u, 16u)) ++fork_levels;
        recurse(objs, 0, n, 1, node0, slot0, n >= kParallelObjects ? fork_levels : 0);
        return (uint32_t)node0;
    }

    // Level of the deepest leaf built so far, the root node being level 1 (its children are level 2).
    int deepest_leaf() const { return deepest_.load(); }

   private:
    struct Ref {
        uint32_t c, n;
        float lo[3], hi[3];
    };
    static constexpr size_t kParallelObjects = 16384;  // smaller (sub)trees are not worth a thread
    std::vector<DevBvhNode>& nodes_;
    std::vector<uint32_t>& slots_;
    int max_depth_;
    uint32_t ppo_;
    float pad_;
    std::atomic<int> deepest_{0};

    void set_child(DevBvhNode& nd, int which, const Ref& r) const {
        const uint32_t ref = r.n ? (0x80000000u | ((r.n - 1u) << 27) | r.c) : r.c;
        const float p = pad_;  // (an empty box stays empty: inf - pad)
        if (which == 0) {
            nd.lo0x = r.lo[0] - p; nd.lo0y = r.lo[1] - p; nd.lo0z = r.lo[2] - p;
            nd.hi0x = r.hi[0] + p; nd.hi0y = r.hi[1] + p; nd.hi0z = r.hi[2] + p;
            nd.ref0 = ref;
        } else {
            nd.lo1x = r.lo[0] - p; nd.lo1y = r.lo[1] - p; nd.lo1z = r.lo[2] - p;
            nd.hi1x = r.hi[0] + p; nd.hi1y = r.hi[1] + p; nd.hi1z = r.hi[2] + p;
            nd.ref1 = ref;
        }
    }
    // A range of this many objects becomes one leaf. One object per leaf unless MIRT_BVH_LEAF asks for more (measurement
    // runs; single-primitive objects only, at most 16 records). The node numbering leaves the nodes such a range would
    // have had unused.
    bool is_leaf(size_t n) const {
        static const size_t leaf_max = [] { const char* v = std::getenv("MIRT_BVH_LEAF"); long l = v ? std::atol(v) : 1; return (size_t)std::min(std::max(l, 1l), 16l); }();
        return n == 1 || (ppo_ == 1 && n <= leaf_max);
    }
    static float area(const float* lo, const float* hi) {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
    static int ceil_log2(size_t n) {
        int l = 0;
        while (((size_t)1 << l) < n) ++l;
        return l;
    }

    // Builds the subtree over objs[begin, end): its n - 1 nodes go to nodes_[node, node + n - 1) (root first, then
    // the left subtree, then the right one), its leaves to slots_[slot, ...). `forks` more levels may start a thread.
    Ref recurse(std::vector<BuildObject>& objs, size_t begin, size_t end, int depth, size_t node, size_t slot, int forks) {
        Ref r;
        for (int a = 0; a < 3; ++a) { r.lo[a] = INFINITY; r.hi[a] = -INFINITY; }
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = begin; i < end; ++i)
            for (int a = 0; a < 3; ++a) {
                r.lo[a] = std::min(r.lo[a], objs[i].lo[a]);
                r.hi[a] = std::max(r.hi[a], objs[i].hi[a]);
                float c = 0.5f * (objs[i].lo[a] + objs[i].hi[a]);
                clo[a] = std::min(clo[a], c);
                chi[a] = std::max(chi[a], c);
            }
        size_t n = end - begin;
        if (is_leaf(n)) {
            int seen = deepest_.load();
            while (seen < depth && !deepest_.compare_exchange_weak(seen, depth)) {}
            r.c = (uint32_t)slot;
            r.n = (uint32_t)n * ppo_;
            for (size_t o = 0; o < n; ++o)
                for (uint32_t q = 0; q < ppo_; ++q) slots_[slot + o * ppo_ + q] = objs[begin + o].first_prim + q;
            return r;
        }
        size_t mid = begin;
        bool use_sah = depth + ceil_log2(n) + 2 < max_depth_;
        if (use_sah) mid = sah_split(objs, begin, end, clo, chi);
        if (mid == begin || mid == end) {
            // median split on the widest centroid axis
            int ax = 0;
            for (int a = 1; a < 3; ++a)
                if (chi[a] - clo[a] > chi[ax] - clo[ax]) ax = a;
            mid = begin + n / 2;
            std::nth_element(objs.begin() + begin, objs.begin() + mid, objs.begin() + end,
                             [ax](const BuildObject& p, const BuildObject& q) { return p.lo[ax] + p.hi[ax] < q.lo[ax] + q.hi[ax]; });
        }
        const size_t nl = mid - begin;
        Ref l, rr;
        if (forks > 0 && n >= kParallelObjects) {
            std::thread left([&] { l = recurse(objs, begin, mid, depth + 1, node + 1, slot, forks - 1); });
            rr = recurse(objs, mid, end, depth + 1, node + nl, slot + nl * ppo_, forks - 1);
            left.join();
        } else {
            l = recurse(objs, begin, mid, depth + 1, node + 1, slot, 0);
            rr = recurse(objs, mid, end, depth + 1, node + nl, slot + nl * ppo_, 0);
        }
        DevBvhNode nd{};
        set_child(nd, 0, l);
        set_child(nd, 1, rr);
        nodes_[node] = nd;
        r.c = (uint32_t)node;
        r.n = 0;
        return r;
    }

    // Binned SAH over centroids; returns the partition point or `begin` if no split helps. 16 bins, and 128 for a range
    // of 4096 objects or more: near the root a bin of sixteen holds tens of thousands of objects and the planes fall on
    // the same few places at every level (1 M spheres: 16.0 -> 13.7 node steps per segment, +13 % samples/s; smaller
    // scenes measured the same or 1 % slower with more bins, and no better with every partition tried).
    // MIRT_BVH_BINS, MIRT_BVH_BINS_LARGE (0 = as small ranges) and MIRT_BVH_LARGE_FROM override the three numbers.
    size_t sah_split(std::vector<BuildObject>& objs, size_t begin, size_t end, const float* clo, const float* chi) {
        constexpr int BMAX = 256;
        static const int B_small = [] { const char* v = std::getenv("MIRT_BVH_BINS"); int b = v ? std::atoi(v) : 16; return std::min(std::max(b, 2), 256); }();
        static const int B_large = [] { const char* v = std::getenv("MIRT_BVH_BINS_LARGE"); int b = v ? std::atoi(v) : 128; return std::min(std::max(b, 0), 256); }();
        static const size_t large_from = [] { const char* v = std::getenv("MIRT_BVH_LARGE_FROM"); return (size_t)(v ? std::atol(v) : 4096); }();
        const int B = (B_large && end - begin >= large_from) ? B_large : B_small;
        float best_cost = INFINITY;
        int best_axis = -1, best_bin = -1;
        for (int ax = 0; ax < 3; ++ax) {
            float ext = chi[ax] - clo[ax];
            if (!(ext > 0.f)) continue;
            float scale = (float)B / ext;
            struct Bin { float lo[3], hi[3]; uint32_t n; } bins[BMAX];
            for (int q = 0; q < B; ++q) { Bin& b = bins[q]; for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; } b.n = 0; }
            for (size_t i = begin; i < end; ++i) {
                float c = 0.5f * (objs[i].lo[ax] + objs[i].hi[ax]);
                int bi = std::min(B - 1, std::max(0, (int)((c - clo[ax]) * scale)));
                Bin& b = bins[bi];
                ++b.n;
                for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], objs[i].lo[a]); b.hi[a] = std::max(b.hi[a], objs[i].hi[a]); }
            }
            float la[BMAX], ra[BMAX];
            uint32_t ln[BMAX], rn[BMAX];
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            uint32_t cnt = 0;
            for (int i = 0; i < B; ++i) {
                if (bins[i].n) for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], bins[i].lo[a]); hi[a] = std::max(hi[a], bins[i].hi[a]); }
                cnt += bins[i].n;
                ln[i] = cnt;
                la[i] = cnt ? area(lo, hi) : 0.f;
            }
            for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
            cnt = 0;
            for (int i = B - 1; i >= 0; --i) {
                if (bins[i].n) for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], bins[i].lo[a]); hi[a] = std::max(hi[a], bins[i].hi[a]); }
                cnt += bins[i].n;
                rn[i] = cnt;
                ra[i] = cnt ? area(lo, hi) : 0.f;
            }
            for (int i = 0; i + 1 < B; ++i) {
                if (ln[i] == 0 || rn[i + 1] == 0) continue;
                float cost = la[i] * (float)ln[i] + ra[i + 1] * (float)rn[i + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = i; }
            }
        }
        if (best_axis < 0) return begin;
        float scale = (float)B / (chi[best_axis] - clo[best_axis]);
        float c0 = clo[best_axis];
        int ax = best_axis, bb = best_bin;
        auto it = std::partition(objs.begin() + begin, objs.begin() + end, [=](const BuildObject& o) {
            float c = 0.5f * (o.lo[ax] + o.hi[ax]);
            int bi = std::min(B - 1, std::max(0, (int)((c - c0) * scale)));
            return bi <= bb;
        });
        return (size_t)(it - objs.begin());
    }
};

// Collapses the binary tree rooted at nodes2[root2] into four-wide nodes appended to nodes4 (2 to 4 children each, in
// the binary tree's order). `by_area`: a node starts from the two children of its binary node and, while it has
// fewer than four, replaces the interior child with the largest surface area by that child's two children — the boxes
// a ray is most likely to enter are opened in the node itself (1 M spheres +2.4 %, `random` +0.8 % against the other
// rule: every interior child replaced once, two binary levels per node). Returns the new
// root's index; *stack_need receives the most references a walk of the new tree can have pushed at once — a step
// pushes every hit child but the one it enters, so a node with k children adds k - 1 to the deepest need below it.
inline uint32_t collapse_to_four(const std::vector<DevBvhNode>& nodes2, uint32_t root2, std::vector<DevBvhNode4>& nodes4, uint32_t* stack_need,
                                 bool by_area) {
    struct Child {
        float lo[3], hi[3];
        uint32_t ref;
    };
    auto children_of = [&](uint32_t n, Child out[2]) {
        const DevBvhNode& b = nodes2[n];
        out[0] = Child{{b.lo0x, b.lo0y, b.lo0z}, {b.hi0x, b.hi0y, b.hi0z}, b.ref0};
        out[1] = Child{{b.lo1x, b.lo1y, b.lo1z}, {b.hi1x, b.hi1y, b.hi1z}, b.ref1};
    };
    // explicit work list: (binary node, index of its four-wide node); needs are filled on the way back
    struct Item { uint32_t n2, n4; };
    std::vector<Item> todo;
    const uint32_t root4 = (uint32_t)nodes4.size();
    nodes4.emplace_back();
    todo.push_back({root2, root4});
    std::vector<uint32_t> made;  // four-wide nodes of this tree in creation order (parents before children)
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        made.push_back(it.n4);
        Child two[2], kids[4];
        int k = 0;
        children_of(it.n2, two);
        for (int c = 0; c < 2; ++c)
            if (two[c].ref != 0xffffffffu) kids[k++] = two[c];  // (a single-leaf tree has no second child)
        if (by_area) {
            // the interior child with the largest box gives way to its two children, until there are four (of equal
            // boxes the one fewer binary levels down: identical objects give the balanced node of the other rule)
            int level[4] = {0, 0, 0, 0};
            while (k < 4) {
                int widest = -1;
                float widest_area = -1.f;
                for (int c = 0; c < k; ++c) {
                    if (kids[c].ref & 0x80000000u) continue;
                    const float dx = kids[c].hi[0] - kids[c].lo[0], dy = kids[c].hi[1] - kids[c].lo[1], dz = kids[c].hi[2] - kids[c].lo[2];
                    const float area = dx * dy + dy * dz + dz * dx;
                    if (area > widest_area || (area == widest_area && level[c] < level[widest])) { widest_area = area; widest = c; }
                }
                if (widest < 0) break;
                Child grand[2];
                children_of(kids[widest].ref, grand);
                for (int c = k; c > widest + 1; --c) { kids[c] = kids[c - 1]; level[c] = level[c - 1]; }
                kids[widest] = grand[0];
                kids[widest + 1] = grand[1];
                level[widest] = level[widest + 1] = level[widest] + 1;
                ++k;
            }
        } else {
            // two levels of the binary tree per node
            Child four[4];
            int n = 0;
            for (int c = 0; c < k; ++c) {
                if (kids[c].ref & 0x80000000u) { four[n++] = kids[c]; continue; }
                Child grand[2];
                children_of(kids[c].ref, grand);
                for (int g = 0; g < 2; ++g)
                    if (grand[g].ref != 0xffffffffu) four[n++] = grand[g];
            }
            k = n;
            for (int c = 0; c < k; ++c) kids[c] = four[c];
        }
        DevBvhNode4 nd{};
        for (int i = 0; i < 4; ++i) {
            const bool have = i < k;
            // a missing child is the point at +infinity, which no ray with a finite far bound reaches (walk_node)
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {INFINITY, INFINITY, INFINITY};
            uint32_t ref = 0xffffffffu;
            if (have) {
                for (int a = 0; a < 3; ++a) { lo[a] = kids[i].lo[a]; hi[a] = kids[i].hi[a]; }
                ref = kids[i].ref;
                if (!(ref & 0x80000000u)) {                     // interior: its own four-wide node
                    const uint32_t n4 = (uint32_t)nodes4.size();
                    nodes4.emplace_back();
                    todo.push_back({ref, n4});
                    ref = n4;
                }
            }
            for (int a = 0; a < 3; ++a) { nd.plane[2 * a][i] = lo[a]; nd.plane[2 * a + 1][i] = hi[a]; }
            nd.ref[i] = ref;
        }
        nd.pad[0] = (uint32_t)k;
        nodes4[it.n4] = nd;
    }
    // needs, children before parents (reverse creation order); pad[1] holds a node's need while this runs
    uint32_t need_root = 0;
    for (size_t i = made.size(); i-- > 0;) {
        DevBvhNode4& nd = nodes4[made[i]];
        uint32_t below = 0;
        for (int c = 0; c < 4; ++c)
            if (nd.ref[c] != 0xffffffffu && !(nd.ref[c] & 0x80000000u)) below = std::max(below, nodes4[nd.ref[c]].pad[1]);
        nd.pad[1] = below + (nd.pad[0] > 0 ? nd.pad[0] - 1u : 0u);
        need_root = nd.pad[1];
    }
    *stack_need = need_root;
    return root4;
}

// The surface-area estimate of what a walk of the four-wide tree under `root` costs a ray that meets the root's box and is
// stopped by nothing: node_cost = the sum over the interior nodes of area(node) / area(root) (the root counts 1), the
// expected number of node steps; leaf_cost = the same sum over the leaf children, the expected number of leaf steps.
inline void tree_cost(const std::vector<DevBvhNode4>& nodes, uint32_t root, double* node_cost, double* leaf_cost) {
    auto child_area = [&](const DevBvhNode4& nd, int c, float* lo, float* hi) {
        for (int a = 0; a < 3; ++a) { lo[a] = nd.plane[2 * a][c]; hi[a] = nd.plane[2 * a + 1][c]; }
        const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    struct Item { uint32_t n; double area; };
    std::vector<Item> todo;
    double root_area = 0.0;
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        const DevBvhNode4& nd = nodes[root];
        for (int c = 0; c < 4; ++c) {
            if (nd.ref[c] == 0xffffffffu) continue;
            float l[3], h[3];
            child_area(nd, c, l, h);
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
        }
        const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        root_area = dx * dy + dy * dz + dz * dx;
    }
    *node_cost = 0.0;
    *leaf_cost = 0.0;
    if (!(root_area > 0.0)) return;
    todo.push_back({root, root_area});
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        *node_cost += it.area / root_area;
        const DevBvhNode4& nd = nodes[it.n];
        for (int c = 0; c < 4; ++c) {
            if (nd.ref[c] == 0xffffffffu) continue;
            float l[3], h[3];
            const double a = child_area(nd, c, l, h);
            if (nd.ref[c] & 0x80000000u) *leaf_cost += a / root_area;
            else todo.push_back({nd.ref[c], a});
        }
    }
}

// Renumbers the four-wide nodes of all hierarchies breadth first from their roots together (all roots, then all their
// interior children, level by level), so that the nodes nearest the roots — the ones every walk visits — are the
// first of the array, whatever number of hierarchies the scene has: a kernel can hold nodes [0, k) in LDS and tell by
// the index alone where a node lives. References inside the nodes and the roots are rewritten; leaf references and
// everything else stay. `roots` are the hierarchies' root indices, in world order.
inline void order_nodes_breadth_first(std::vector<DevBvhNode4>& nodes, std::vector<uint32_t*>& roots) {
    const uint32_t none = 0xffffffffu;
    std::vector<uint32_t> order, new_index(nodes.size(), none);
    order.reserve(nodes.size());
    for (uint32_t* r : roots) {
        if (new_index[*r] != none) continue;
        new_index[*r] = (uint32_t)order.size();
        order.push_back(*r);
    }
    for (size_t head = 0; head < order.size(); ++head) {
        const DevBvhNode4& nd = nodes[order[head]];
        for (int c = 0; c < 4; ++c) {
            const uint32_t ref = nd.ref[c];
            if (ref == none || (ref & 0x80000000u) || new_index[ref] != none) continue;
            new_index[ref] = (uint32_t)order.size();
            order.push_back(ref);
        }
    }
    // (nodes no root reaches cannot exist; they would keep their place behind the others)
    for (uint32_t i = 0; i < nodes.size(); ++i)
        if (new_index[i] == none) { new_index[i] = (uint32_t)order.size(); order.push_back(i); }
    std::vector<DevBvhNode4> out(nodes.size());
    for (uint32_t i = 0; i < nodes.size(); ++i) {
        DevBvhNode4 nd = nodes[i];
        for (int c = 0; c < 4; ++c)
            if (nd.ref[c] != none && !(nd.ref[c] & 0x80000000u)) nd.ref[c] = new_index[nd.ref[c]];
        out[new_index[i]] = nd;
    }
    for (uint32_t* r : roots) *r = new_index[*r];
    nodes.swap(out);
}

}  // namespace mirt
