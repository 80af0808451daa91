// Construction of the reference's SAH bounding-volume hierarchy (new_bvh / new_bvh_helper, aggregate.rs:187-322) on the GPU.
//
// The reference recurses top-down and, at every node, stable-sorts its shapes by box centre along the widest axis of the
// centre bounds and sweeps all split positions (O(n log^2 n) on one host thread: 32 s for 2 M triangles).  Here all nodes of
// one tree level are processed together over the whole shape array, with segmented device primitives:
//   1. a scan by segment yields every node's bounds and centre bounds (at the segment's last element);
//   2. one stable radix sort on (segment start, centre along the node's axis) sorts every node's shapes;
//   3. a forward and a backward scan by segment yield the prefix and suffix bounds of every split position;
//   4. a fourth scan finds every node's first minimum of the reference's cost function;
//   5. nodes split (or become leaves) and the next level's segments are formed.
// The result is the reference's tree, bit for bit: bounds are min/max folds whose tie rule (`a < b ? a : b`, which makes the
// LATER operand win between +0 and -0) is kept by running the same non-commutative operator in element order; the cost is
// the reference's expression in f32 (this file is compiled without contraction); sorting is stable with -0 ordered as +0,
// as `<` does; and all of the reference's quirks are restated: every fold starts from the point box at the origin
// (aggregate.rs:195, 423-428), and the suffix bounds never include the last shape (aggregate.rs:247-252).
// Nodes are produced in breadth-first order and renumbered into the reference's depth-first pre-order on the host.
//
// `clean` builds the tree the algorithm was meant to build: bounds are those of the shapes alone (no origin box) and the
// suffix bounds include the last shape.  That tree is NOT the reference's; it serves RRT_AGGREGATE_BVH_CLEAN, whose
// traversal returns what the reference's List aggregate returns (aggregate.rs:142-165), at a fraction of the node visits.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cstring>
#include <string>
#include <vector>

#include "rrt_bvh_build.hpp"
#include "rrt_error.hpp"

namespace {

struct Box6 {
    float lo[3], hi[3];
};
struct Bounds2 {  // shape bounds and centre bounds, scanned together
    Box6 box, cen;
};
struct Best {
    float cost;
    uint32_t idx;  // split position relative to the segment start
};
struct Seg {
    uint32_t lo, n, node;  // element range and breadth-first node number
    uint32_t axis;         // 0..2, or 3 = becomes a leaf
    uint32_t split;        // split position (left part = split + 1 elements); valid for inner nodes
    uint32_t child;        // index of the left child in the next level's segment list; 0xffffffff = leaf
};
struct BfsNode {
    float bmin[3], bmax[3];
    uint32_t lo, n;
    uint32_t axis;          // 3 = leaf
    uint32_t left, right;   // breadth-first numbers of the children
};

constexpr uint32_t kNone = 0xffffffffu;
constexpr int kThreads = 256;

// Vector3::min / max (base.rs:29-43): `a < b ? a : b` per component; the accumulated value is the left operand.
__device__ __forceinline__ Box6 merge_boxes(const Box6& a, const Box6& b) {
    Box6 r;
    for (int c = 0; c < 3; ++c) {
        r.lo[c] = a.lo[c] < b.lo[c] ? a.lo[c] : b.lo[c];
        r.hi[c] = a.hi[c] > b.hi[c] ? a.hi[c] : b.hi[c];
    }
    return r;
}
struct MergeOp {
    __device__ __forceinline__ Box6 operator()(const Box6& a, const Box6& b) const { return merge_boxes(a, b); }
};
struct MergeFlippedOp {  // for scans over the reversed array: the operand that comes later in element order is still "b"
    __device__ __forceinline__ Box6 operator()(const Box6& a, const Box6& b) const { return merge_boxes(b, a); }
};
struct Merge2Op {
    __device__ __forceinline__ Bounds2 operator()(const Bounds2& a, const Bounds2& b) const {
        Bounds2 r;
        r.box = merge_boxes(a.box, b.box);
        r.cen = merge_boxes(a.cen, b.cen);
        return r;
    }
};
struct BestOp {  // `if cost < best { best = cost; best_i = i }` (aggregate.rs:266-270): the first minimum wins
    __device__ __forceinline__ Best operator()(const Best& a, const Best& b) const { return b.cost < a.cost ? b : a; }
};
__device__ __forceinline__ Box6 origin_box() {
    Box6 b;
    for (int c = 0; c < 3; ++c) b.lo[c] = b.hi[c] = 0.0f;
    return b;
}
__device__ __forceinline__ float box_area(const Box6& b) {  // aggregate.rs:463-469
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}
__device__ __forceinline__ float centre_of(const Box6& b, int c) { return b.lo[c] * 0.5f + b.hi[c] * 0.5f; }  // aggregate.rs:444-450

// element i of the current order -> its bounds and its centre as a point box
__global__ void k_gather_bounds(const Box6* __restrict__ boxes, const uint32_t* __restrict__ order, uint32_t n, Bounds2* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Box6 b = boxes[order[i]];
    Bounds2 r;
    r.box = b;
    for (int c = 0; c < 3; ++c) r.cen.lo[c] = r.cen.hi[c] = centre_of(b, c);
    out[i] = r;
}

// per segment: bounds (written to the node), leaf rules that need no sorting, cut axis
__global__ void k_choose_axis(Seg* __restrict__ segs, uint32_t n_segs, const Bounds2* __restrict__ scanned, BfsNode* __restrict__ nodes, bool clean,
                              uint32_t leaf_max) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    Seg sg = segs[s];
    const Bounds2 last = scanned[sg.lo + sg.n - 1];
    const Box6 total = clean ? last.box : merge_boxes(origin_box(), last.box);  // the reference's fold starts from the point box at the origin
    const Box6 cen = clean ? last.cen : merge_boxes(origin_box(), last.cen);
    BfsNode nd;
    for (int c = 0; c < 3; ++c) {
        nd.bmin[c] = total.lo[c];
        nd.bmax[c] = total.hi[c];
    }
    nd.lo = sg.lo;
    nd.n = sg.n;
    nd.left = nd.right = kNone;
    uint32_t axis = 3;
    if (sg.n > leaf_max) {  // aggregate.rs:204
        const float dx = cen.hi[0] - cen.lo[0], dy = cen.hi[1] - cen.lo[1], dz = cen.hi[2] - cen.lo[2];
        axis = (dx > dy && dx > dz) ? 0u : (dy > dz ? 1u : 2u);  // aggregate.rs:452-461
        if (cen.hi[axis] == cen.lo[axis]) axis = 3;              // aggregate.rs:225
    }
    nd.axis = axis;
    nodes[sg.node] = nd;
    sg.axis = axis;
    sg.child = kNone;
    sg.split = 0;
    segs[s] = sg;
}

// sort keys: (segment start, centre along the segment's axis) for segments that may split, (segment start, rank) otherwise;
// finished elements are segments of their own (start = own position), so the sort leaves them where they are
__global__ void k_make_keys(const Box6* __restrict__ boxes, const uint32_t* __restrict__ order, const uint32_t* __restrict__ pos_lo,
                            const uint32_t* __restrict__ pos_seg, const Seg* __restrict__ segs, uint32_t n, unsigned long long* __restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t lo = pos_lo[i], s = pos_seg[i];
    uint32_t sub = i - lo;
    if (s != kNone && segs[s].axis < 3) {
        float k = centre_of(boxes[order[i]], (int)segs[s].axis);
        if (k == 0.0f) k = 0.0f;  // -0 sorts as +0, as `<` has it
        const uint32_t b = __float_as_uint(k);
        sub = b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
    }
    keys[i] = ((unsigned long long)lo << 32) | sub;
}

// bounds in sorted order, forwards; and backwards with every segment's last element left out (aggregate.rs:247-252)
__global__ void k_gather_sorted(const Box6* __restrict__ boxes, const uint32_t* __restrict__ order, const uint32_t* __restrict__ pos_lo,
                                const uint32_t* __restrict__ pos_seg, const Seg* __restrict__ segs, uint32_t n, Box6* __restrict__ fwd,
                                Box6* __restrict__ rev, uint32_t* __restrict__ rev_key, bool clean) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Box6 b = boxes[order[i]];
    fwd[i] = b;
    const uint32_t s = pos_seg[i];
    bool is_last = true;
    if (s != kNone) is_last = i == segs[s].lo + segs[s].n - 1;
    Box6 r = b;
    if (is_last && !(clean && s != kNone))  // neutral element: loses every comparison
        for (int c = 0; c < 3; ++c) {
            r.lo[c] = __builtin_inff();
            r.hi[c] = -__builtin_inff();
        }
    rev[n - 1 - i] = r;
    rev_key[n - 1 - i] = pos_lo[i];
}

// cost of splitting after element i (aggregate.rs:254-265)
__global__ void k_costs(const uint32_t* __restrict__ pos_seg, const Seg* __restrict__ segs, const BfsNode* __restrict__ nodes,
                        const Box6* __restrict__ fwd_scan, const Box6* __restrict__ rev_scan, uint32_t n, Best* __restrict__ out, bool clean) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Best r;
    r.cost = FLT_MAX;  // never below the initial `best`
    r.idx = 0;
    const uint32_t s = pos_seg[i];
    if (s != kNone) {
        const Seg sg = segs[s];
        const uint32_t k = i - sg.lo;
        if (sg.axis < 3 && k + 1 < sg.n) {
            const BfsNode nd = nodes[sg.node];
            Box6 total;
            for (int c = 0; c < 3; ++c) {
                total.lo[c] = nd.bmin[c];
                total.hi[c] = nd.bmax[c];
            }
            const float total_area = box_area(total);
            // reference: suffix[k+1] = shapes k+1 .. n-2 in order, then the origin box (which the loop starts from); clean: the
            // bounds of shapes 0..k and of shapes k+1..n-1, nothing else
            const Box6 prefix = clean ? fwd_scan[i] : merge_boxes(origin_box(), fwd_scan[i]);
            const Box6 suffix = clean ? rev_scan[n - 1 - (i + 1)] : merge_boxes(rev_scan[n - 1 - (i + 1)], origin_box());
            const float cost = 1.0f + ((box_area(prefix) / total_area) * (float)(k + 1)) + ((box_area(suffix) / total_area) * (float)(sg.n - (k + 1)));
            if (cost < FLT_MAX) {
                r.cost = cost;
                r.idx = k;
            }
        }
    }
    out[i] = r;
}

// per segment: split when the best cost beats the leaf cost (aggregate.rs:272), and open the children
__global__ void k_split(Seg* __restrict__ segs, uint32_t n_segs, const Best* __restrict__ best_scan, BfsNode* __restrict__ nodes, Seg* __restrict__ next,
                        uint32_t* __restrict__ counters /* [0] = nodes so far, [1] = segments of the next level */) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    Seg sg = segs[s];
    if (sg.axis >= 3) return;
    const Best b = best_scan[sg.lo + sg.n - 1];
    if (!(b.cost < (float)sg.n)) {
        nodes[sg.node].axis = 3;
        sg.axis = 3;
        segs[s] = sg;
        return;
    }
    const uint32_t node = atomicAdd(&counters[0], 2u);
    const uint32_t slot = atomicAdd(&counters[1], 2u);
    nodes[sg.node].left = node;
    nodes[sg.node].right = node + 1;
    Seg l, r;
    l.lo = sg.lo;
    l.n = b.idx + 1;
    l.node = node;
    r.lo = sg.lo + b.idx + 1;
    r.n = sg.n - (b.idx + 1);
    r.node = node + 1;
    l.axis = r.axis = 3;
    l.split = r.split = 0;
    l.child = r.child = kNone;
    next[slot] = l;
    next[slot + 1] = r;
    sg.split = b.idx;
    sg.child = slot;
    segs[s] = sg;
}

// per element: which segment of the next level it belongs to (or none: its node became a leaf)
__global__ void k_advance(uint32_t* __restrict__ pos_lo, uint32_t* __restrict__ pos_seg, const Seg* __restrict__ segs, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = pos_seg[i];
    if (s == kNone) return;
    const Seg sg = segs[s];
    if (sg.child == kNone) {
        pos_lo[i] = i;
        pos_seg[i] = kNone;
    } else if (i - sg.lo <= sg.split) {
        pos_seg[i] = sg.child;
    } else {
        pos_lo[i] = sg.lo + sg.split + 1;
        pos_seg[i] = sg.child + 1;
    }
}

__global__ void k_iota(uint32_t* __restrict__ a, uint32_t n, uint32_t value_or_index) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = value_or_index == kNone ? i : value_or_index;
}

struct DeviceBuffers {
    std::vector<void*> owned;
    ~DeviceBuffers() {
        for (void* p : owned) (void)hipFree(p);
    }
    template <typename T>
    hipError_t get(T** out, size_t count) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) owned.push_back(p);
        *out = (T*)p;
        return e;
    }
};

#define BVH_TRY(expr)                                                                                                    \
    do {                                                                                                                 \
        hipError_t e_ = (expr);                                                                                          \
        if (e_ != hipSuccess) return rrt_set_error(RRT_ERR_DEVICE, std::string("BVH build: " #expr ": ") + hipGetErrorString(e_)); \
    } while (0)

inline unsigned blocks_for(uint32_t n) { return (n + kThreads - 1) / kThreads; }

}  // namespace

int rrt_bvh_build_gpu(const float* boxes6, uint32_t n, bool clean, std::vector<rrt_node>* out_nodes, std::vector<uint32_t>* out_order, uint32_t* out_depth) {
    out_nodes->clear();
    out_order->clear();
    *out_depth = 0;
    if (n == 0) return RRT_OK;
    if (n > 0x7fffffffu) return rrt_set_error(RRT_ERR_LIMIT, "BVH build: too many shapes");
    DeviceBuffers mem;
    Box6 *d_boxes, *d_fwd, *d_rev, *d_fwd_scan, *d_rev_scan;
    Bounds2 *d_b2, *d_b2_scan;
    Best *d_best, *d_best_scan;
    uint32_t *d_order, *d_order_alt, *d_pos_lo, *d_pos_seg, *d_rev_key, *d_counters;
    unsigned long long *d_keys, *d_keys_alt;
    Seg *d_segs, *d_next;
    BfsNode* d_nodes;
    const size_t max_nodes = 2 * (size_t)n;  // a binary tree over at most n leaves
    BVH_TRY(mem.get(&d_boxes, n));
    BVH_TRY(mem.get(&d_fwd, n));
    BVH_TRY(mem.get(&d_rev, n));
    BVH_TRY(mem.get(&d_fwd_scan, n));
    BVH_TRY(mem.get(&d_rev_scan, n));
    BVH_TRY(mem.get(&d_b2, n));
    BVH_TRY(mem.get(&d_b2_scan, n));
    BVH_TRY(mem.get(&d_best, n));
    BVH_TRY(mem.get(&d_best_scan, n));
    BVH_TRY(mem.get(&d_order, n));
    BVH_TRY(mem.get(&d_order_alt, n));
    BVH_TRY(mem.get(&d_pos_lo, n));
    BVH_TRY(mem.get(&d_pos_seg, n));
    BVH_TRY(mem.get(&d_rev_key, n));
    BVH_TRY(mem.get(&d_counters, 2));
    BVH_TRY(mem.get(&d_keys, n));
    BVH_TRY(mem.get(&d_keys_alt, n));
    BVH_TRY(mem.get(&d_segs, n));
    BVH_TRY(mem.get(&d_next, n));
    BVH_TRY(mem.get(&d_nodes, max_nodes));
    BVH_TRY(hipMemcpy(d_boxes, boxes6, (size_t)n * sizeof(Box6), hipMemcpyHostToDevice));

    // temporary storage of the device primitives: the largest of the four uses
    size_t temp_bytes = 0, need = 0;
    int end_bit = 32;
    while ((1ull << (end_bit - 32)) < (unsigned long long)n && end_bit < 64) ++end_bit;
    BVH_TRY(hipcub::DeviceScan::InclusiveScanByKey(nullptr, need, d_pos_lo, d_b2, d_b2_scan, Merge2Op(), (int)n));
    temp_bytes = need > temp_bytes ? need : temp_bytes;
    BVH_TRY(hipcub::DeviceScan::InclusiveScanByKey(nullptr, need, d_pos_lo, d_fwd, d_fwd_scan, MergeOp(), (int)n));
    temp_bytes = need > temp_bytes ? need : temp_bytes;
    BVH_TRY(hipcub::DeviceScan::InclusiveScanByKey(nullptr, need, d_rev_key, d_rev, d_rev_scan, MergeFlippedOp(), (int)n));
    temp_bytes = need > temp_bytes ? need : temp_bytes;
    BVH_TRY(hipcub::DeviceScan::InclusiveScanByKey(nullptr, need, d_pos_lo, d_best, d_best_scan, BestOp(), (int)n));
    temp_bytes = need > temp_bytes ? need : temp_bytes;
    BVH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, d_keys, d_keys_alt, d_order, d_order_alt, (int)n, 0, end_bit));
    temp_bytes = need > temp_bytes ? need : temp_bytes;
    void* d_temp = nullptr;
    BVH_TRY(mem.get((unsigned char**)&d_temp, temp_bytes));

    // aggregate.rs:204.  Clean trees keep it: leaves of up to 4 / 8 / 16 shapes save almost no node visits (10 M triangles: 87 ->
    // 86 / 83 / 77 per ray) and multiply the pair tests (12 -> 31 / 72 / 129): 349 -> 389 / 496 / 658 ms at 3840x2160 @ 8.
    const uint32_t leaf_max = 2;
    // level 0: one segment holding every shape in scene order
    hipLaunchKernelGGL(k_iota, dim3(blocks_for(n)), dim3(kThreads), 0, nullptr, d_order, n, kNone);
    hipLaunchKernelGGL(k_iota, dim3(blocks_for(n)), dim3(kThreads), 0, nullptr, d_pos_lo, n, 0u);
    hipLaunchKernelGGL(k_iota, dim3(blocks_for(n)), dim3(kThreads), 0, nullptr, d_pos_seg, n, 0u);
    Seg root;
    root.lo = 0;
    root.n = n;
    root.node = 0;
    root.axis = 3;
    root.split = 0;
    root.child = kNone;
    BVH_TRY(hipMemcpy(d_segs, &root, sizeof(Seg), hipMemcpyHostToDevice));
    uint32_t counters[2] = {1u, 0u};  // nodes so far, segments of the next level
    uint32_t n_segs = 1, levels = 0;
    while (n_segs > 0) {
        ++levels;
        if (levels > 4096) return rrt_set_error(RRT_ERR_LIMIT, "BVH build: tree deeper than 4096 levels");
        counters[1] = 0;
        BVH_TRY(hipMemcpy(d_counters, counters, sizeof(counters), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_gather_bounds, dim3(blocks_for(n)), dim3(kThreads), 0, nullptr, d_boxes, d_order, n, d_b2);
        need = temp_bytes;
        BVH_TRY(hipcub::DeviceScan::InclusiveScanByKey(d_temp, need, d_pos_lo, d_b2, d_b2_scan, Merge2Op(), (int)n));
        hipLaunchKernelGGL(k_choose_axis, dim3(blocks_for(n_segs)), dim3(kThreads), 0, nullptr, d_segs, n_segs, d_b2_scan, d_nodes, clean, leaf_max);
        hipLaunchKernelGGL(k_make_keys, dim3(blocks_for(n)), dim3(kThreads), 0, nullptr, d_boxes, d_order, d_pos_lo, d_pos_seg, d_segs, n, d_keys);
        need = temp_bytes;
        BVH_TRY(hipcub::DeviceRadixSort::SortPairs(d_temp, need, d_keys, d_keys_alt, d_order, d_order_alt, (int)n, 0, end_bit));
        std::swap(d_order, d_order_alt);
        hipLaunchKernelGGL(k_gather_sorted, dim3(blocks_for(n)), dim3(kThreads), 0, nullptr, d_boxes, d_order, d_pos_lo, d_pos_seg, d_segs, n, d_fwd, d_rev,
                           d_rev_key, clean);
        need = temp_bytes;
        BVH_TRY(hipcub::DeviceScan::InclusiveScanByKey(d_temp, need, d_pos_lo, d_fwd, d_fwd_scan, MergeOp(), (int)n));
        need = temp_bytes;
        BVH_TRY(hipcub::DeviceScan::InclusiveScanByKey(d_temp, need, d_rev_key, d_rev, d_rev_scan, MergeFlippedOp(), (int)n));
        hipLaunchKernelGGL(k_costs, dim3(blocks_for(n)), dim3(kThreads), 0, nullptr, d_pos_seg, d_segs, d_nodes, d_fwd_scan, d_rev_scan, n, d_best, clean);
        need = temp_bytes;
        BVH_TRY(hipcub::DeviceScan::InclusiveScanByKey(d_temp, need, d_pos_lo, d_best, d_best_scan, BestOp(), (int)n));
        hipLaunchKernelGGL(k_split, dim3(blocks_for(n_segs)), dim3(kThreads), 0, nullptr, d_segs, n_segs, d_best_scan, d_nodes, d_next, d_counters);
        hipLaunchKernelGGL(k_advance, dim3(blocks_for(n)), dim3(kThreads), 0, nullptr, d_pos_lo, d_pos_seg, d_segs, n);
        BVH_TRY(hipGetLastError());
        BVH_TRY(hipMemcpy(counters, d_counters, sizeof(counters), hipMemcpyDeviceToHost));
        if (counters[0] > max_nodes) return rrt_set_error(RRT_ERR_DEVICE, "BVH build: node count out of range");
        n_segs = counters[1];
        std::swap(d_segs, d_next);
    }

    // breadth-first -> the reference's depth-first pre-order (left child = index + 1, right child stored)
    const uint32_t n_nodes = counters[0];
    std::vector<BfsNode> bfs(n_nodes);
    BVH_TRY(hipMemcpy(bfs.data(), d_nodes, (size_t)n_nodes * sizeof(BfsNode), hipMemcpyDeviceToHost));
    out_order->resize(n);
    BVH_TRY(hipMemcpy(out_order->data(), d_order, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    out_nodes->reserve(n_nodes);
    struct Item {
        uint32_t bfs, parent_out, depth;  // parent_out: output index of the inner node whose RIGHT child this is (kNone otherwise)
    };
    std::vector<Item> todo;
    todo.push_back({0u, kNone, 0u});
    uint32_t deepest = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        const BfsNode& b = bfs[it.bfs];
        const uint32_t idx = (uint32_t)out_nodes->size();
        if (it.parent_out != kNone) (*out_nodes)[it.parent_out].a = idx;
        if (it.depth > deepest) deepest = it.depth;
        rrt_node nd;
        std::memcpy(nd.bmin, b.bmin, 12);
        std::memcpy(nd.bmax, b.bmax, 12);
        if (b.axis >= 3 || b.left == kNone) {
            nd.a = b.lo;  // leaves own consecutive ranges of the final order, in depth-first order
            nd.b = RRT_LEAF_FLAG | b.n;
            out_nodes->push_back(nd);
        } else {
            nd.a = 0;
            nd.b = b.axis;
            out_nodes->push_back(nd);
            todo.push_back({b.right, idx, it.depth + 1});
            todo.push_back({b.left, kNone, it.depth + 1});
        }
    }
    *out_depth = deepest;
    return RRT_OK;
}
