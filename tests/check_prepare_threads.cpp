// Test program (built and run by tests/test_host_scene.py under ThreadSanitizer, and under AddressSanitizer + UBSan): the CPU half
// of a scene upload on a tree large enough for the threaded leaf matching, once with one thread and once with many; the two
// results must be the same bytes.  The tree is a complete binary tree over leaves of four triangles, written directly as a
// scene description (no SAH build: the preparation only needs a well-formed tree).
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../rust_raytracer_amd/csrc/rrt_scene_prepare.hpp"

extern "C" const char* rrt_last_error(void);

static void build(uint32_t lo, uint32_t hi, std::vector<rrt_node>& nodes) {  // leaves [lo, hi) in pre-order
    const uint32_t me = (uint32_t)nodes.size();
    nodes.push_back(rrt_node{});
    rrt_node n{};
    for (int c = 0; c < 3; ++c) {
        n.bmin[c] = -1.0e3f;
        n.bmax[c] = 1.0e3f;
    }
    if (hi - lo == 1) {
        n.a = lo * 4u;
        n.b = RRT_LEAF_FLAG | 4u;
        nodes[me] = n;
        return;
    }
    const uint32_t mid = lo + (hi - lo) / 2;
    build(lo, mid, nodes);
    n.a = (uint32_t)nodes.size();
    n.b = (lo + hi) % 3u;
    build(mid, hi, nodes);
    nodes[me] = n;
}

int main() {
    const uint32_t n_leaves = 1u << 16, n_slots = n_leaves * 4u;
    std::vector<rrt_node> nodes;
    nodes.reserve(2 * n_leaves);
    build(0, n_leaves, nodes);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> pos(-900.0f, 900.0f), edge(-1.0f, 1.0f);
    std::vector<rrt_slot> slots(n_slots);
    std::vector<rrt_tri_shade> shade(n_slots);
    std::memset(slots.data(), 0, slots.size() * sizeof(rrt_slot));
    std::memset(shade.data(), 0, shade.size() * sizeof(rrt_tri_shade));
    for (uint32_t i = 0; i < n_slots; ++i) {
        for (int c = 0; c < 3; ++c) {
            slots[i].v0[c] = pos(rng);
            slots[i].e1[c] = edge(rng);
            slots[i].e2[c] = edge(rng);
        }
        slots[i].prim = i;
        slots[i].info = 0u;  // a triangle of material 0
    }
    rrt_material mat{};
    mat.type = RRT_MAT_LAMBERT;
    mat.albedo = 0;
    mat.bump = -1;
    rrt_texture tex{};
    tex.type = RRT_TEX_CONSTANT;
    rrt_scene_desc d{};
    d.nodes = nodes.data();
    d.n_nodes = (uint32_t)nodes.size();
    d.slots = slots.data();
    d.tri_shade = shade.data();
    d.n_slots = n_slots;
    d.materials = &mat;
    d.n_materials = 1;
    d.textures = &tex;
    d.n_textures = 1;
    d.bvh_depth = 17;
    d.aggregate_kind = RRT_AGGREGATE_BVH;
    d.resolution_x = d.resolution_y = 8;
    d.samples = 1;
    rrt::PreparedScene one, many;
    if (rrt::prepare_scene(&d, &one, 1) != 0 || rrt::prepare_scene(&d, &many, 16) != 0) {
        std::printf("prepare_scene failed: %s\n", rrt_last_error());
        return 1;
    }
    const bool same = one.pairs == many.pairs && one.nodes == many.nodes && one.plain == many.plain && one.desc_hash == many.desc_hash &&
                      one.stack_depth == many.stack_depth && one.features == many.features;
    std::printf("%zu node records, %zu pair records: %s\n", one.nodes.size() / 16, one.pairs.size() / 32, same ? "threads ok" : "DIFFERENT");
    return same ? 0 : 1;
}
