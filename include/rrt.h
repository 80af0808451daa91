/* rrt.h — C ABI of the MI355X path-tracing core (librrt_hip.so).
 *
 * The reference (trevday/rust-raytracer) is a single Rust binary with no FFI; the
 * seams this library replaces are Rust-internal.  Each entry point below names the
 * reference interface it stands in for (paths relative to the reference root).
 *
 * Conventions: every function returning int gives 0 on success and a non-zero
 * rrt_status on failure, with rrt_last_error() returning a thread-local,
 * NUL-terminated description.  Handles are opaque; one render at a time per
 * handle.  All pointers are caller-owned unless stated.  Functions whose name
 * ends in _device take HIP device pointers; `stream` is a hipStream_t passed as
 * void* (NULL = the default stream).  The library operates on the calling
 * thread's current HIP device.
 */
#ifndef RRT_H
#define RRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rrt_status {
    RRT_OK = 0,
    RRT_ERR_INVALID = 1,   /* bad argument / malformed scene description */
    RRT_ERR_SCENE = 2,     /* scene file could not be loaded (message mirrors scene.rs's DeserializeError text) */
    RRT_ERR_DEVICE = 3,    /* HIP runtime failure */
    RRT_ERR_LIMIT = 4,     /* scene exceeds a stated limit of this build (e.g. BVH deeper than the traversal stack) */
    RRT_ERR_IO = 5         /* file system failure (e.g. output file already exists, main.rs:98-106) */
} rrt_status;

/* ---- flattened scene records (HBM layout; all 16-byte aligned) ---------------------------- */

/* One BVH node, 32 bytes, 4 per 128-byte line.  Nodes are stored in the DFS pre-order that
 * new_bvh_helper (aggregate.rs:193-322) produces: the left child of inner node i is i+1.
 *   inner: a = absolute index of the right child (i + right_offset), b = cut axis (0,1,2)
 *   leaf : a = first slot, b = 0x80000000 | slot count */
typedef struct rrt_node {
    float bmin[3];
    uint32_t a;
    float bmax[3];
    uint32_t b;
} rrt_node;
#define RRT_LEAF_FLAG 0x80000000u

/* One primitive slot, 48 bytes; slots are stored in leaf order (the order List::hit,
 * aggregate.rs:144-165, walks a leaf).
 *   triangle: v0, e1 = v1 - v0, e2 = v2 - v0 (the f32 differences Triangle::hit forms, shape.rs:340-341)
 *   sphere / medium: v0,e1,e2 unused; aux = index into spheres[] / media[]
 * info: bits 0..23 material index, bit 28 backface culling, bits 30..31 kind; bits 24..27 and 29 are reserved (ignored). */
typedef struct rrt_slot {
    float v0[3];
    uint32_t prim; /* index of the shape in scene order (what Aggregate::hit's &SyncShape identifies) */
    float e1[3];
    uint32_t info;
    float e2[3];
    uint32_t aux;
} rrt_slot;
#define RRT_KIND_TRIANGLE 0u
#define RRT_KIND_SPHERE 1u
#define RRT_KIND_MEDIUM 2u
#define RRT_INFO_CULL (1u << 28)
#define RRT_INFO_KIND(info) ((info) >> 30)
#define RRT_INFO_MATERIAL(info) ((info)&0x00ffffffu)

/* Per-triangle shading record, 64 bytes, same index as the slot: the unit normal normalized(e1 x e2) as
 * Triangle::get_hit_properties forms it (shape.rs:390, before the back-face flip), texture coordinates
 * (shape.rs:190-210 defaults applied) and the cached dp/du, dp/dv of Triangle::new (shape.rs:278-314). */
typedef struct rrt_tri_shade {
    float nrm[3];
    uint32_t pad;
    float uv0[2], uv1[2], uv2[2];
    float pu[3];
    float pv[3];
} rrt_tri_shade;

/* Sphere (shape.rs:32-46): rows 0..2 of world_to_local and local_to_world (row-major 3x4), 128 bytes. */
typedef struct rrt_sphere {
    float w2l[12];
    float l2w[12];
    float radius;
    uint32_t material;
    uint32_t pad[6];
} rrt_sphere;

/* ConstantMedium (volume.rs:44-48).  boundary_kind is RRT_KIND_SPHERE (index into spheres[]), RRT_KIND_TRIANGLE (index into
 * boundary_tris[] / boundary_tri_shade[]) or RRT_KIND_MEDIUM (index into media[], of an EARLIER record: scene.rs:588 lets the
 * boundary be any single shape, a ConstantMedium included).  A chain of media may be RRT_MAX_MEDIUM_NESTING records long. */
#define RRT_MAX_MEDIUM_NESTING 3
typedef struct rrt_medium {
    uint32_t boundary_kind;
    uint32_t boundary_index;
    float density;
    uint32_t material; /* the phase function */
} rrt_medium;

typedef enum rrt_material_type {
    RRT_MAT_LAMBERT = 0,       /* material.rs:84-115 */
    RRT_MAT_METAL = 1,         /* material.rs:117-165 */
    RRT_MAT_DIELECTRIC = 2,    /* material.rs:167-212 */
    RRT_MAT_DIFFUSE_LIGHT = 3, /* material.rs:214-236 */
    RRT_MAT_ISOTROPIC = 4      /* volume.rs:17-40 */
} rrt_material_type;

typedef struct rrt_material {
    uint32_t type;
    int32_t albedo; /* texture index: albedo or emission; -1 for Dielectric */
    int32_t bump;   /* texture index or -1 */
    float param;    /* Metal roughness (already clamped to [0,1]) or Dielectric refractive index */
} rrt_material;

typedef enum rrt_texture_type {
    RRT_TEX_CONSTANT = 0,  /* a,b,c = r,g,b                         texture.rs:70-78  */
    RRT_TEX_TEST = 1,      /*                                       texture.rs:80-93  */
    RRT_TEX_CHECKER = 2,   /* a = repeat, u0 = odd, u1 = even       texture.rs:95-119 */
    RRT_TEX_IMAGE = 3,     /* u0 = width, u1 = height, u2 = channels (1 or 3), u3 = byte offset into texels   texture.rs:121-141 */
    RRT_TEX_NOISE = 4,     /* a = scale                             texture.rs:143-151 */
    RRT_TEX_TURBULENCE = 5 /* a = scale, b = omega, u0 = depth      texture.rs:153-179 */
} rrt_texture_type;

typedef struct rrt_texture {
    uint32_t type;
    float a, b, c;
    uint32_t u0, u1, u2, u3;
} rrt_texture;

/* One member of the light mixture (pdf.rs:85-115; members are PDF::Shape of every shape whose
 * material is_important(), scene.rs:131-140, in scene order), 96 bytes.
 *   triangle: v0,v1,v2 world-space vertices; e1,e2 as in rrt_slot; area = 0.5*|e1 x e2| (shape.rs:454)
 *   sphere  : index into spheres[] */
typedef struct rrt_light {
    uint32_t kind;
    uint32_t index;
    uint32_t cull;
    float area;
    float v0[3], v1[3], v2[3];
    float e1[3], e2[3];
    float nrm[3]; /* normalized(e1 x e2), shape.rs:390 */
    uint32_t pad[2];
} rrt_light;

/* Camera (camera.rs:11-19), as Camera::new (camera.rs:22-51) leaves it. */
typedef struct rrt_camera {
    float origin[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float u[3];
    float v[3];
    float lens_radius;
} rrt_camera;

/* Everything the device needs, as host arrays.  ≙ the fields of scene::Scene (scene.rs:21-26) after
 * scene::deserialize (scene.rs:64-159) and new_bvh (aggregate.rs:187-191). */
typedef struct rrt_scene_desc {
    uint32_t n_nodes;
    const rrt_node* nodes;
    uint32_t n_slots;
    const rrt_slot* slots;
    const rrt_tri_shade* tri_shade; /* n_slots entries */
    uint32_t n_spheres;
    const rrt_sphere* spheres;
    uint32_t n_media;
    const rrt_medium* media;
    uint32_t n_boundary_tris;
    const rrt_slot* boundary_tris;
    const rrt_tri_shade* boundary_tri_shade;
    uint32_t n_materials;
    const rrt_material* materials;
    uint32_t n_textures;
    const rrt_texture* textures;
    uint64_t texel_bytes;
    const uint8_t* texels;
    uint32_t n_lights;
    const rrt_light* lights;
    rrt_camera camera;
    uint32_t resolution_x, resolution_y, samples; /* Logistics, scene.rs:28-34 */
    uint32_t bvh_depth;                           /* maximum number of simultaneously pending stack entries + 1 */
    uint32_t aggregate_kind;                      /* RRT_AGGREGATE_LIST ("Aggregate": "List", scene.rs:610: one leaf holding every shape in scene
                                                     order, no box test), RRT_AGGREGATE_BVH (the reference's tree, wherever it was built) or
                                                     RRT_AGGREGATE_BVH_CLEAN (see rrt_host_finalize) */
} rrt_scene_desc;

typedef struct rrt_scene rrt_scene; /* device-resident scene */
typedef struct rrt_host_scene rrt_host_scene; /* host-side builder / loader */

enum { RRT_BG_BLACK = 0 /* main.rs:268-270 */, RRT_BG_SKY = 1 /* main.rs:259-267 */ };
/* Bounding-box test of the traversal: what is allowed to cull.
 *   TIGHT       the reference's test on its own boxes AND the padded boxes the primitives really occupy (nodes), padded pair boxes
 *               before the exact primitive tests (shallow trees): the fast one
 *   REFERENCE   nodes by aggregate.rs:471-503 verbatim alone (the padded pair boxes of shallow trees still apply)
 *   EXHAUSTIVE  nodes by the reference's test alone and every primitive of a visited leaf through the exact test: no padded box
 *               decides anything, so it is the reference's result by construction - the yardstick for the other two
 * What each promises:
 *   shallow trees (at most 16 nodes, Lists included - every scene file of the reference): all three return the same hits and
 *     frames, bit for bit.  A padded box can only drop a "hit" that the rounding of Triangle::hit (shape.rs:335-372) lets in from
 *     BESIDE a triangle, for a ray that runs almost inside the triangle's plane; such rays are recognised (a test of the ray's
 *     direction against every plane direction of the scene, with the worst-case rounding bound of the triangle test) and take no
 *     shortcut at all.  Spheres rely on the padding alone (their test's rounding stays 50 times below it in scenes like the
 *     reference's; no bound is proven for a sphere a thousand radii away from the ray's origin).
 *   deep trees (more than 16 nodes): REFERENCE and EXHAUSTIVE are the same thing there (leaves are always tested whole) and
 *     equal the reference; TIGHT has no such recognition and may drop exactly those rounding artefacts for rays closer than
 *     about 0.01 rad to a triangle's plane (a few rays in 10 000 of a batch aimed along triangle outlines, none in a frame checked).
 *   clean trees (RRT_AGGREGATE_BVH_CLEAN): the mode is ignored - their traversal is their own (byte-quantised padded boxes,
 *     the reference's List semantics, see rrt_host_finalize), with TIGHT's contract for grazing rays. */
enum { RRT_AABB_TIGHT = 0, RRT_AABB_REFERENCE = 1, RRT_AABB_EXHAUSTIVE = 2 };

typedef struct rrt_render_params {
    uint32_t width, height;   /* 0 = the scene's Logistics */
    uint32_t samples;         /* samples per pixel rendered by this call; 0 = the scene's Logistics */
    uint32_t first_sample;    /* index of the first sample in the per-pixel random stream (sample-range sharding) */
    uint32_t seed;
    uint32_t background;      /* RRT_BG_* */
    int32_t max_depth;        /* < 0 = 50 (aggregate.rs:14) */
    uint32_t aabb_mode;       /* RRT_AABB_* */
    uint32_t tile_count;      /* interleaved 32x32-pixel tile sharding: this call renders tiles with index % tile_count == tile_index; 0/1 = all */
    uint32_t tile_index;
    uint32_t flags;           /* RRT_RENDER_* */
} rrt_render_params;

/* rrt_render_params.flags.  ACCUMULATE: the output buffer already holds the sums of the samples before first_sample and
 * this call's samples are added after them, in sample order (main.rs:244-245), so rendering [0,a) then [a,b) with the flag
 * gives bit for bit what one call over [0,b) gives; without it the call's pixels start from zero. */
enum { RRT_RENDER_ACCUMULATE = 1 };

/* ≙ the one measurement the reference makes, `Success! Took {} seconds` (main.rs:64, 189-192), split by phase.  All times are
 * seconds of device time measured with HIP events, for the host-buffer entry points (rrt_render, rrt_render_multi). */
typedef struct rrt_stats {
    uint64_t rays;              /* Aggregate::hit calls made by trace (aggregate.rs:24) */
    uint64_t paths;             /* camera samples traced */
    double render_seconds;      /* trace + accumulate kernels; rrt_render_multi: the slowest device */
    double trace_seconds;       /* the path-tracing kernel alone (same device as render_seconds) */
    double accumulate_seconds;  /* the per-pixel sums in sample order (main.rs:244-245) */
    double reduce_seconds;      /* rrt_render_multi: gathering the devices' frames on the first device and adding them; else 0 */
    double copy_seconds;        /* the frame's way to the device (when it must keep the caller's values) and back to the host */
} rrt_stats;

/* ---- host side: scene construction -------------------------------------------------------- */

/* ≙ scene::deserialize(&str, &Path, &mut Resources) (scene.rs:64-68) followed by the flattening.
 * Logistics overrides: pass 0 / NaN to keep the file's value. */
int rrt_host_scene_load(const char* scene_json_path, uint32_t override_width, uint32_t override_height, uint32_t override_samples,
                        float override_aspect, rrt_host_scene** out);
/* The same, choosing what the file's "BVH" aggregate becomes: RRT_AGGREGATE_BVH (the reference's tree, built on the host),
 * RRT_AGGREGATE_BVH_GPU (the same tree, built on the device) or RRT_AGGREGATE_BVH_CLEAN (see rrt_host_finalize).  A "List"
 * stays a list.  No counterpart in the reference, whose files have no such choice. */
int rrt_host_scene_load_as(const char* scene_json_path, uint32_t override_width, uint32_t override_height, uint32_t override_samples,
                           float override_aspect_ratio, int bvh_kind, rrt_host_scene** out);
/* Programmatic construction, one call per scene.rs deserialize_* function. Indices returned through *id. */
int rrt_host_scene_new(rrt_host_scene** out);
int rrt_host_add_texture(rrt_host_scene*, const rrt_texture* t, const uint8_t* texels /* Image only: w*h*channels bytes */, int32_t* id);
int rrt_host_add_material(rrt_host_scene*, uint32_t type, int32_t albedo, int32_t bump, float param, int32_t* id); /* scene.rs:263-407 */
int rrt_host_add_sphere(rrt_host_scene*, const float translate[3], const float rotate[3], const float scale[3], float radius,
                        int32_t material); /* scene.rs:449-470 */
/* One .obj object (scene.rs:503-561): vertices in object space (f64 like wavefront_obj), 3 indices per triangle, tidx < 0 = none. */
int rrt_host_add_mesh(rrt_host_scene*, const float translate[3], const float rotate[3], const float scale[3], uint32_t n_vertices,
                      const double* vertices, uint32_t n_tex, const double* tex, uint32_t n_triangles, const int32_t* vidx,
                      const int32_t* tidx, int cull, int32_t material);
int rrt_host_add_medium_sphere(rrt_host_scene*, const float translate[3], const float rotate[3], const float scale[3], float radius,
                               int32_t boundary_material, float density, int32_t phase_material); /* scene.rs:573-602 */
/* scene.rs:573-602 when the boundary's own "type" is "ConstantMedium": the shape added last, which must be a medium, leaves the
 * scene's list and becomes the boundary of a new medium, which takes its place. */
int rrt_host_wrap_last_medium(rrt_host_scene*, float density, int32_t phase_material);
int rrt_host_set_camera(rrt_host_scene*, const float position[3], const float look_at[3], const float up[3], float fov, float aspect,
                        float aperture, float focus_distance); /* camera.rs:66-88 */
int rrt_host_set_logistics(rrt_host_scene*, uint32_t resolution_x, uint32_t resolution_y, uint32_t samples, int use_importance_sampling);
/* Builds the light list (scene.rs:130-140), the aggregate (scene.rs:605-619; aggregate.rs:187-322) and the flattened arrays.
 * RRT_AGGREGATE_BVH_GPU builds the reference's tree, bit for bit, on the current device (one sort and four scans per tree
 * level instead of the reference's recursive re-sorting).
 * RRT_AGGREGATE_BVH_CLEAN (no counterpart in the reference; for large meshes) builds, also on the device, the tree the
 * reference's sweep yields WITHOUT its two slips (bounds that always include the world origin, aggregate.rs:195; suffix bounds
 * that leave out the last shape, aggregate.rs:247-252) and makes queries return what the reference's LIST aggregate returns:
 * the nearest hit, equal distances going to the shape that is first in scene order.  That differs from the reference's BVH
 * only where its box test prunes on an exact coincidence or a tie is met in another order (a handful of rays in a million on
 * meshes), and costs a small fraction of the node visits.  The two _GPU/_CLEAN kinds need a GPU, the first two do not. */
enum {
    RRT_AGGREGATE_LIST = 0 /* aggregate.rs:142-165 */,
    RRT_AGGREGATE_BVH = 1 /* aggregate.rs:187-322 on the host */,
    RRT_AGGREGATE_BVH_GPU = 2,
    RRT_AGGREGATE_BVH_CLEAN = 3
};
int rrt_host_finalize(rrt_host_scene*, int aggregate);
const rrt_scene_desc* rrt_host_desc(const rrt_host_scene*); /* valid after finalize/load, owned by the handle */
/* Host wall-clock seconds of the two phases the reference's total (main.rs:64, 191) contains before tracing: reading and decoding
 * the scene file with its meshes and images (scene.rs:64-128; 0 for scenes built through rrt_host_add_*), and building the light
 * list, the aggregate and the flattened arrays (scene.rs:130-159, aggregate.rs:187-322). */
int rrt_host_scene_times(const rrt_host_scene*, double* load_seconds, double* build_seconds);
void rrt_host_scene_destroy(rrt_host_scene*);

/* ---- device side ---------------------------------------------------------------------------- */

/* Uploads the flattened scene to the current device's HBM.  ≙ moving Scene into Arc (main.rs:85-95). */
int rrt_scene_create(const rrt_scene_desc* desc, rrt_scene** out);
void rrt_scene_destroy(rrt_scene*);
/* The CPU half of rrt_scene_create alone: no device is touched and nothing is uploaded.  Runs the same checks of the description
 * (every index a kernel would follow, the links forming one tree, the limits) with the same error codes and messages, prepares the
 * same device records, and reports what would be uploaded.  layout (may be NULL) receives 8 values: node records (64 bytes each),
 * pair records (128 bytes each), plain shading records (32 bytes each; 0 unless every texture is Constant and the scene has neither
 * bump maps nor media), 16-byte entries of the grazing guard's table (0 for trees of more than 16 nodes and clean trees),
 * traversal stack entries per ray, the feature bits of the kernel variant the scene needs (1 spheres, 2 media, 4 textures other
 * than Constant, 8 bump maps, 16 media bounded by media), flags (1: single leaf whose lights are its own primitives, 2: the
 * deep-tree kernels, 4: clean tree), and a hash of the description (what rrt_render_multi compares between its handles).
 * No counterpart in the reference, whose Scene is valid by construction (scene.rs:130-159). */
int rrt_scene_check(const rrt_scene_desc* desc, uint64_t* layout);
/* Host wall-clock seconds rrt_scene_create took (pairing the leaves' triangles, primitive boxes, the copies to HBM). */
int rrt_scene_upload_seconds(const rrt_scene*, double* seconds);
/* The parts of that time, in seconds (any pointer may be NULL): the first call that needs the device (a process that has not used
 * the GPU before creates its context there), the first access to the library's code object (the runtime loads it onto the device at
 * that moment), the host-side preparation (validation, pairing, boxes, tables), and the device allocations and copies.  Only a
 * process's first handle on a device pays the first two.  ≙ nothing in the reference, whose `Took {} seconds` (main.rs:64, 191) is
 * one number. */
int rrt_scene_upload_breakdown(const rrt_scene*, double* first_device_call, double* code_object, double* host_preparation, double* copies);

/* ≙ Aggregate::hit (aggregate.rs:118-124) for n rays: closest hit in (t_min, t_max).
 * o_xyz, d_xyz: n*3 floats.  prim_id[i] = shape index in scene order or -1; t_hit[i] = distance (t_max on miss).
 * Stochastic primitives (ConstantMedium) key their draw by (seed, ray index). */
int rrt_intersect_batch(rrt_scene*, const float* o_xyz, const float* d_xyz, size_t n, float t_min, float t_max, uint32_t aabb_mode,
                        uint32_t seed, int32_t* prim_id, float* t_hit);
int rrt_intersect_batch_device(rrt_scene*, const float* d_o_xyz, const float* d_d_xyz, size_t n, float t_min, float t_max, uint32_t aabb_mode,
                               uint32_t seed, int32_t* d_prim_id, float* d_t_hit, void* stream);

/* ≙ queue fill + thread_work + join (main.rs:150-165, 200-257): accumulates params->samples samples of every
 * pixel (of this call's tiles) and writes the per-pixel radiance SUM, row-major [y][x][3] f32 (pixels of other
 * tiles are left untouched).  stats may be NULL. */
int rrt_render(rrt_scene*, const rrt_render_params* params, float* rgb_sum, rrt_stats* stats);
/* Same with a device output buffer; enqueues on `stream` and returns without synchronising.
 * d_counters: optional device buffer of 2 uint64 {rays, paths} that the kernel adds to. */
int rrt_render_device(rrt_scene*, const rrt_render_params* params, float* d_rgb_sum, uint64_t* d_counters, void* stream);

/* ---- several GPUs of one node in one process ------------------------------------------------------------------
 * ≙ the thread pool of main.rs:136-165: there N host threads drain one queue of samples into one colour buffer; here one host
 * thread per GPU traces a share of the frame on a scene replicated in that GPU's HBM, and the shares meet on the first GPU:
 * every other GPU's frame is copied there over its own xGMI link (the links are point to point, so the N-1 copies run side by
 * side) and added in device order.  No counterpart of the sharding choice in the reference (its queue is dynamic). */
enum {
    RRT_SHARD_SAMPLES = 0, /* device i traces a contiguous range of the samples of every pixel (the first devices take the remainder);
                              equal to one device's frame up to the f32 rounding of adding N partial sums */
    RRT_SHARD_TILES = 1    /* device i traces all samples of the 32x32-pixel tiles with index % N == i: bit-identical to one device's frame */
};
int rrt_device_count(int* count);
/* rrt_scene_create on device `device` (the calling thread's current device is left as it was). */
int rrt_scene_create_on(const rrt_scene_desc* desc, int device, rrt_scene** out);
/* scenes[0..n): the same description uploaded once per device (two handles on one device are allowed and simply share it).
 * params->tile_count / tile_index must be 0 (the call does its own sharding).  rgb_sum: host buffer, row-major [y][x][3]; without
 * RRT_RENDER_ACCUMULATE every pixel is overwritten, with it the samples are added to the sums already there.  stats may be NULL. */
int rrt_render_multi(rrt_scene* const* scenes, uint32_t n, uint32_t shard_mode, const rrt_render_params* params, float* rgb_sum, rrt_stats* stats);

/* rrt_render_device for callers that cannot allocate device memory while they render (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same results bit for bit;
 * the per-sample radiance staging lives in an area of RRT_NOALLOC_STAGING_BYTES that is part of the library's code object (one
 * copy per GPU, held until the process ends) instead of the scene handle's own buffer, and the call makes no hipMalloc, hipFree,
 * event creation or other allocating call: kernel launches, and a memset of the scene handle's work-item counters, on `stream` only.
 * A frame whose samples do not all fit is traced in several passes.  Limits: width * height * 12 bytes must not exceed RRT_NOALLOC_STAGING_BYTES (11 184 810 pixels), else
 * RRT_ERR_LIMIT; rrt_scene_set_timing does not cover these renders.  All calls on one GPU share that GPU's area: calls that
 * overlap in time on one GPU (different streams or scene handles) race; calls on one stream, or ordered by events, do not.
 * The first call on a GPU brings that GPU's copy of the area into being (the runtime loads the area's code object then, about
 * 45 ms, and allocates while it does): a caller that must not allocate makes one call during set-up.
 * Like every render entry point it also uses the handle's work-item counters (zeroed on `stream` before each launch): two renders
 * of ONE handle must be ordered on the device whichever entry point they use, or they skip and duplicate work items. */
#define RRT_NOALLOC_STAGING_BYTES (128u << 20)
int rrt_render_device_noalloc(rrt_scene*, const rrt_render_params* params, float* d_rgb_sum, uint64_t* d_counters, void* stream);

/* Memory budget of the scene handle's per-sample radiance staging buffer (12 bytes per pixel per sample of a pass;
 * default 16 GiB, 0 restores the default).  A frame whose samples do not fit is traced in several passes; the image
 * does not depend on the split.  No counterpart in the reference (its colours vector, main.rs:115-121, is per pixel). */
int rrt_scene_set_staging_limit(rrt_scene*, uint64_t bytes);

/* Deep trees (more than 16 nodes, clean trees): how many of a ray's pending-node stack entries the render kernel keeps in
 * LDS (1..64; 0 restores the default of 32, which with the leaf phase's work area lets a compute unit hold 16 waves); deeper entries live
 * in the lane's private memory.  Changes speed only, never results.  No counterpart in the reference (its stack is a
 * Vec on the host heap, aggregate.rs:352). */
int rrt_scene_set_lds_stack(rrt_scene*, uint32_t entries);

/* Shallow trees (the asset scenes) with constant textures are traced on a pooled schedule - a wave keeps 128 paths in LDS and
 * runs one kind of work at a time over 64 of them - unless this is switched off (enable = 0: one path per lane, all kinds of work
 * in every turn).  Changes speed only, never results.  No counterpart in the reference. */
int rrt_scene_set_pooling(rrt_scene*, int enable);

/* Per-kernel timing (≙ the reference's only instrumentation, the wall clock of main.rs:64, 191, split by kernel).
 * While enabled, every render pass records device events around its trace and accumulate kernels on the caller's
 * stream; rrt_scene_kernel_times waits for the recorded passes, returns the summed durations in milliseconds and the
 * number of passes, and forgets them. */
int rrt_scene_set_timing(rrt_scene*, int enable);
int rrt_scene_kernel_times(rrt_scene*, double* trace_ms, double* accumulate_ms, uint32_t* passes);

/* ≙ main.rs:173-186: out[i] = (u8)(sqrt(sum[i] / spp) * 255.99), saturating, NaN -> 0.  Host buffers. */
int rrt_resolve_rgb8(const float* rgb_sum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* out);
int rrt_resolve_rgb8_device(const float* d_rgb_sum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* d_out, void* stream);

/* ≙ PNGEncoder::encode(&data, res_x, res_y, ColorType::RGB(8)) on a file opened with create_new (main.rs:98-107, 188):
 * writes an 8-bit RGB PNG; with create_new != 0 an existing file is an RRT_ERR_IO error ("Failed to create new file"). */
int rrt_write_png(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height, int create_new);

/* Device-side unit probes used by the parity tests (each ≙ the named reference function, evaluated on the GPU):
 *   which = 0 camera rays for (px,py,sample) triples in `in` (3 u32 each) -> 6 floats            (main.rs:227-229, camera.rs:53-63)
 *           1 Shape::get_hit_properties for (o,d,prim,t) (8 x 4 bytes each) -> 14 floats         (shape.rs:87-123, 374-424)
 *           2 Texture::value for texture `arg` at (u,v,px,py,pz) -> 3 floats                     (texture.rs:61-179)
 *           3 deterministic sin,cos,asin,log,atan2(x,y) for (x,y) -> 5 floats
 *           4 first 8 floats of stream (seed=arg, pixel, sample) for (pixel,sample) -> 8 floats */
int rrt_probe(rrt_scene*, uint32_t which, uint32_t arg, const void* in, size_t n, float* out);

/* No counterpart in the reference: checks on the current device that the 3-instruction reciprocal the triangle test uses
 * (shape.rs:354 `1.0 / determinant`) equals IEEE division for every one of the 2^32 float bit patterns. */
int rrt_selftest_reciprocal(uint64_t* mismatches);

const char* rrt_last_error(void);
const char* rrt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RRT_H */
