// GPU construction of the reference's SAH BVH (aggregate.rs:187-322); see rrt_bvh_build.hip.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/rrt.h"

// boxes6: n x (lo.xyz, hi.xyz), the shapes' bounding boxes in scene order (host memory).
// Produces the nodes in the reference's depth-first pre-order (leaf `a` = first position in `order`), the final shape
// order (leaves own consecutive ranges) and the depth of the deepest node.  Returns RRT_OK or an error code.
// clean = false: the reference's tree, bit for bit.  clean = true: the same sweep without the reference's two slips (bounds
// that always include the world origin; suffix bounds that leave out the last shape) - a different, much better tree.
int rrt_bvh_build_gpu(const float* boxes6, uint32_t n, bool clean, std::vector<rrt_node>* nodes, std::vector<uint32_t>* order, uint32_t* depth);
