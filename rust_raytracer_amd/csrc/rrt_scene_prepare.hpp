// Everything rrt_scene_create works out on the CPU before it touches a device: the checks of the description, the pair records
// and their padded boxes, the node records (the reference's tree with primitive boxes, or the four-wide byte-grid nodes of a clean
// tree), the plain shading records, the grazing guard's table, which kernel variant the scene needs.  Plain C++, no device calls.
#pragma once
#include <vector>

#include "rrt_layout.hpp"

namespace rrt {

struct PreparedScene {
    std::vector<float> nodes;   // 16 floats per node record (SceneData::nodes)
    std::vector<float> pairs;   // 4 * kPairQuads floats per pair record (SceneData::pairs)
    std::vector<float> plain;   // 8 floats per slot, or empty (SceneData::plain)
    std::vector<float> guards;  // shallow trees: the grazing guard's table (SceneData::guards)
    std::vector<rrt_sphere> spheres;  // the description's, with pad[0] filled in (kSphereTranslationOnly)
    uint32_t root_link = 0;     // clean trees: SceneData::root_link
    uint32_t stack_depth = 1;   // traversal stack entries per ray, measured on the tree
    uint32_t features = 0;      // F_* bits the scene needs
    bool harvest = false;       // single leaf, 1..kHarvestLights lights, each one a primitive of the leaf
    bool clean = false, deep = false;
    uint64_t desc_hash = 0;     // of the description's arrays and camera
};

// Returns an RRT_* code (the message through rrt_set_error).  max_threads: 0 = as many as the machine offers, 16 at most (large
// trees only; the result does not depend on it).
int prepare_scene(const rrt_scene_desc* desc, PreparedScene* out, unsigned max_threads = 0);

}  // namespace rrt
