// The layout of what the host half of the library writes and the kernels read: the scene's arrays in device memory, the bits and
// constants of their records, the kernel variants a scene may need, the traversal modes.  Plain C++ (no device code): included by
// the host-side sources (scene preparation, rendering) and by the kernels' headers alike.
#pragma once
#include <hip/hip_vector_types.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/rrt.h"

namespace rrt {

// ---- scene features a kernel variant is compiled for ------------------------------------------------------------------
// The host picks the smallest variant that covers the scene, so scenes without media / image textures / bump maps do not
// carry that code (registers, instruction cache) through the inner loop.
enum : uint32_t {
    F_SPHERE = 1u,  // sphere primitives (or sphere lights)
    F_MEDIUM = 2u,  // ConstantMedium primitives and the Isotropic phase function
    F_TEX = 4u,     // textures other than Constant (needs uv / world point in texture lookups)
    F_BUMP = 8u,    // bump maps (needs pu, pv)
    F_ALL = 15u,    // all of the above
    F_NESTED = 16u, // media whose boundary is a medium
    F_EVERY = 31u
};

// ---- the scene's arrays in device memory ------------------------------------------------------------------------------------
struct SceneData {
    const float4* nodes;      // 4 x float4 per node: the rrt_node (a leaf's `a` holds its first PAIR index), then the padded
                              // box its primitives occupy (lo, hi), which the TIGHT traversal tests as well: one 64-byte line.
                              // Clean trees: inner nodes only, four-wide, 4 x float4 (64 bytes) each: the origin and the power-of-two steps
                              // of the node's byte grid, the primitive boxes of up to four children as bytes on it, plane by plane (lo.x of
                              // the four, hi.x, lo.y, hi.y, lo.z, hi.z), and their links (dwords 10..13); a link is an inner node's index,
                              // kLinkLeaf | (pairs - 1) << 28 | first pair, or kLinkNone
    const float4* pairs;      // kPairQuads x float4 per two consecutive primitives of a leaf (see triangle_pair_eval)
    const float4* slots;      // 3 x float4 per rrt_slot
    const float4* plain;      // scenes whose textures are all Constant, without bump maps and media: 2 x float4 per slot holding all
                              // that shading a hit on it needs - (unit normal | -, kind / material type / sphere index), (the
                              // material's colour, its roughness or refractive index) - one fetch instead of slot -> material -> texture
    const float4* tri_shade;  // 4 x float4 per rrt_tri_shade
    const rrt_sphere* spheres;
    const rrt_medium* media;
    const float4* btris;
    const float4* btri_shade;
    const rrt_material* materials;
    const rrt_texture* textures;
    const uint8_t* texels;
    const rrt_light* lights;
    const float4* guards;     // shallow trees: the grazing guard's table - a header of two float4, then one scaled plane normal per
                              // distinct triangle plane of the scene that is not perpendicular to an axis (see grazes_a_plane)
    uint32_t n_nodes, n_slots, n_lights, is_list;
    uint32_t root_link;  // clean trees: where the walk starts (see walk_step)
    rrt_camera cam;

};

// ---- pair records and leaf nodes ----------------------------------------------------------------------------------------------
constexpr uint32_t kPairQuads = 8;  // float4 per pair record: six of primitive data, then the padded box of the two primitives (lo, hi)
#define RRT_KIND_NONE 3u  // second half of an odd leaf's last pair
constexpr uint32_t kLeafDirect = 0x40000000u;  // in a shallow tree's leaf node, beside the pair count: the leaf's FIRST record is tested directly
// A pair record's last two words are a lane mask for the harvesting query: all ones when the pair holds a light's primitive (lanes that
// only collect light distances stay), zero when it does not (they are struck) - two words, so that the scalar unit takes them as one
// 64-bit operand.
constexpr uint32_t kPairLitMask = 0xffffffffu;
// The half-width of a padding record's box (the host evens out a shallow leaf's number of sifted records with one): negative, so the
// near distance lies beyond the far one on every axis the ray moves along, and no ray meets it.
constexpr uint32_t kPadBoxBits = 0xbf800000u;  // -1.0f
constexpr uint32_t kSiftCap = 512;  // items the candidate list holds (the CAP template argument below; a caller short of LDS passes less)
constexpr uint32_t kHarvestLights = 3;
constexpr uint32_t sift_bytes(uint32_t cap) { return 64 * 8 + cap * 4; }
constexpr uint32_t kSiftBytesPerWave = sift_bytes(kSiftCap);
constexpr uint32_t kHarvestBytesPerWave = 64 * kHarvestLights * 4;
#define RRT_PAIR_LIGHT(info) (((info) >> 24) & 15u)  // device-only bits of a pair record's info words: light number + 1
// the cooperative leaf phase of the deep-tree kernels (coop_leaves), per wave
constexpr uint32_t kCoopBytesPerWave = 64 * 8 + 64 * 16 + 128 * 4;  // closest keys, hit payloads, item list

// DScene::plain, word 3 of a slot's first quad: bits 0..2 material type, 3..4 primitive kind, 5.. sphere index
#define RRT_PLAIN_TYPE(bits) ((bits) & 7u)
#define RRT_PLAIN_KIND(bits) (((bits) >> 3) & 3u)
#define RRT_PLAIN_AUX(bits) ((bits) >> 5)
#define RRT_PLAIN_BITS(type, kind, aux) ((type) | ((kind) << 3) | ((aux) << 5))

// rrt_sphere::pad[0] in the device copy: the sphere's world_to_local matrix only translates (see sphere_hit)
constexpr uint32_t kSphereTranslationOnly = 1u;

// ---- traversal modes beyond the caller's RRT_AABB_TIGHT and RRT_AABB_REFERENCE ---------------------------------------------------
// Third traversal mode, chosen by the library for trees built with RRT_AGGREGATE_BVH_CLEAN (never by the caller): nodes are
// tested with the ordinary interval test on their primitive boxes alone, and equal distances go to the shape that comes
// first in SCENE order - together exactly what the reference's List aggregate returns (aggregate.rs:142-165).
constexpr int kAabbClean = 2;
// Fourth traversal mode (RRT_AABB_EXHAUSTIVE): nodes by the reference's test alone, and EVERY primitive of a visited leaf gets the
// exact test - no padded box decides anything, so the result is the reference's by construction.  The yardstick the padded-box
// shortcuts (primitive boxes of nodes, pair boxes of the sifting query) are held against.
constexpr int kAabbExhaustive = 3;
constexpr uint32_t kLinkLeaf = 0x80000000u;  // clean trees: a child link that names a leaf's pair records directly
constexpr uint32_t kLinkNone = 0xffffffffu;  // clean trees: an unused child place of a four-wide node

}  // namespace rrt
