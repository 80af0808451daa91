//! Raw bindings of `librrt_hip.so`, the MI355X path-tracing core (C ABI: `include/rrt.h`).
//!
//! UNCOMPILED HERE: this crate was written against the header without a Rust toolchain at hand.  Every `#[repr(C)]` struct is
//! spelled out field for field in the header's order; the `const _: () = assert!(..)` lines state the sizes and offsets the C
//! side asserts too (`rust_raytracer_amd/csrc/rrt_abi_layout.cpp`), so a slip on either side fails a build, not a render.
//!
//! Where each call replaces code of the reference's `main.rs` / `scene.rs` is noted per function: the host keeps its command line
//! (`main.rs:38-107`) and its PNG writer (`main.rs:188-197`) and replaces queue fill, thread pool and resolve (`main.rs:115-186`) by
//! `rrt_host_scene_load` (or the builder calls), `rrt_scene_create`, `rrt_render` / `rrt_render_multi` and `rrt_resolve_rgb8`.
#![allow(non_camel_case_types)]

use std::mem::size_of;
use std::os::raw::{c_char, c_int, c_void};

// ---- status codes (rrt_status) ---------------------------------------------------------------------------------------
pub const RRT_OK: c_int = 0;
pub const RRT_ERR_INVALID: c_int = 1;
pub const RRT_ERR_SCENE: c_int = 2;
pub const RRT_ERR_DEVICE: c_int = 3;
pub const RRT_ERR_LIMIT: c_int = 4;
pub const RRT_ERR_IO: c_int = 5;

// ---- enumerations ----------------------------------------------------------------------------------------------------------
pub const RRT_LEAF_FLAG: u32 = 0x8000_0000;
pub const RRT_KIND_TRIANGLE: u32 = 0;
pub const RRT_KIND_SPHERE: u32 = 1;
pub const RRT_KIND_MEDIUM: u32 = 2;
pub const RRT_INFO_CULL: u32 = 1 << 28;
pub const RRT_MAT_LAMBERT: u32 = 0;
pub const RRT_MAT_METAL: u32 = 1;
pub const RRT_MAT_DIELECTRIC: u32 = 2;
pub const RRT_MAT_DIFFUSE_LIGHT: u32 = 3;
pub const RRT_MAT_ISOTROPIC: u32 = 4;
pub const RRT_TEX_CONSTANT: u32 = 0;
pub const RRT_TEX_TEST: u32 = 1;
pub const RRT_TEX_CHECKER: u32 = 2;
pub const RRT_TEX_IMAGE: u32 = 3;
pub const RRT_TEX_NOISE: u32 = 4;
pub const RRT_TEX_TURBULENCE: u32 = 5;
pub const RRT_BG_BLACK: u32 = 0; // main.rs:268-270
pub const RRT_BG_SKY: u32 = 1; // main.rs:259-267 (commented out in the reference)
// Shallow trees (at most 16 nodes, Lists: every scene file of the reference): all three modes return the same hits, bit for bit.
// Deep trees: REFERENCE = EXHAUSTIVE = the reference; TIGHT may drop a rounding artefact of the triangle test for a ray within
// about 0.01 rad of a triangle's plane.  Clean trees ignore the mode (see rrt.h).
pub const RRT_AABB_TIGHT: u32 = 0; // the reference's node test AND padded primitive boxes
pub const RRT_AABB_REFERENCE: u32 = 1; // nodes by aggregate.rs:471-503 verbatim alone
pub const RRT_AABB_EXHAUSTIVE: u32 = 2; // no padded box culls anything: the reference by construction
pub const RRT_RENDER_ACCUMULATE: u32 = 1;
pub const RRT_AGGREGATE_LIST: c_int = 0; // "Aggregate": "List", scene.rs:610
pub const RRT_AGGREGATE_BVH: c_int = 1; // aggregate.rs:187-322 on the host
pub const RRT_AGGREGATE_BVH_GPU: c_int = 2; // the same tree, built on the device
pub const RRT_AGGREGATE_BVH_CLEAN: c_int = 3; // clean tree, List semantics
pub const RRT_SHARD_SAMPLES: u32 = 0;
pub const RRT_SHARD_TILES: u32 = 1;
pub const RRT_NOALLOC_STAGING_BYTES: usize = 128 << 20;

// ---- opaque handles ----------------------------------------------------------------------------------------------------
#[repr(C)]
pub struct rrt_scene {
    _private: [u8; 0],
}
#[repr(C)]
pub struct rrt_host_scene {
    _private: [u8; 0],
}

// ---- flattened scene records -------------------------------------------------------------------------------------------
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_node {
    pub bmin: [f32; 3], // offset 0
    pub a: u32,         // offset 12
    pub bmax: [f32; 3], // offset 16
    pub b: u32,         // offset 28
}
const _: () = assert!(size_of::<rrt_node>() == 32);

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_slot {
    pub v0: [f32; 3], // offset 0
    pub prim: u32,    // offset 12
    pub e1: [f32; 3], // offset 16
    pub info: u32,    // offset 28
    pub e2: [f32; 3], // offset 32
    pub aux: u32,     // offset 44
}
const _: () = assert!(size_of::<rrt_slot>() == 48);

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_tri_shade {
    pub nrm: [f32; 3], // offset 0
    pub pad: u32,      // offset 12
    pub uv0: [f32; 2], // offset 16
    pub uv1: [f32; 2], // offset 24
    pub uv2: [f32; 2], // offset 32
    pub pu: [f32; 3],  // offset 40
    pub pv: [f32; 3],  // offset 52
}
const _: () = assert!(size_of::<rrt_tri_shade>() == 64);

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_sphere {
    pub w2l: [f32; 12], // offset 0
    pub l2w: [f32; 12], // offset 48
    pub radius: f32,    // offset 96
    pub material: u32,  // offset 100
    pub pad: [u32; 6],  // offset 104
}
const _: () = assert!(size_of::<rrt_sphere>() == 128);

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_medium {
    pub boundary_kind: u32,  // offset 0
    pub boundary_index: u32, // offset 4
    pub density: f32,        // offset 8
    pub material: u32,       // offset 12
}
const _: () = assert!(size_of::<rrt_medium>() == 16);

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_material {
    pub type_: u32,  // offset 0   (`type` in C)
    pub albedo: i32, // offset 4
    pub bump: i32,   // offset 8
    pub param: f32,  // offset 12
}
const _: () = assert!(size_of::<rrt_material>() == 16);

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_texture {
    pub type_: u32, // offset 0   (`type` in C)
    pub a: f32,     // offset 4
    pub b: f32,     // offset 8
    pub c: f32,     // offset 12
    pub u0: u32,    // offset 16
    pub u1: u32,    // offset 20
    pub u2: u32,    // offset 24
    pub u3: u32,    // offset 28
}
const _: () = assert!(size_of::<rrt_texture>() == 32);

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_light {
    pub kind: u32,     // offset 0
    pub index: u32,    // offset 4
    pub cull: u32,     // offset 8
    pub area: f32,     // offset 12
    pub v0: [f32; 3],  // offset 16
    pub v1: [f32; 3],  // offset 28
    pub v2: [f32; 3],  // offset 40
    pub e1: [f32; 3],  // offset 52
    pub e2: [f32; 3],  // offset 64
    pub nrm: [f32; 3], // offset 76
    pub pad: [u32; 2], // offset 88
}
const _: () = assert!(size_of::<rrt_light>() == 96);

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_camera {
    pub origin: [f32; 3],            // offset 0
    pub lower_left_corner: [f32; 3], // offset 12
    pub horizontal: [f32; 3],        // offset 24
    pub vertical: [f32; 3],          // offset 36
    pub u: [f32; 3],                 // offset 48
    pub v: [f32; 3],                 // offset 60
    pub lens_radius: f32,            // offset 72
}
const _: () = assert!(size_of::<rrt_camera>() == 76);

/// Everything the device needs, as host arrays (≙ `scene::Scene`, scene.rs:21-26, after `new_bvh`).  On the 64-bit targets the
/// library is built for every pointer is 8 bytes and 8-aligned, which puts 4 bytes of padding after each count that precedes one.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_scene_desc {
    pub n_nodes: u32,                             // offset 0
    pub nodes: *const rrt_node,                   // offset 8
    pub n_slots: u32,                             // offset 16
    pub slots: *const rrt_slot,                   // offset 24
    pub tri_shade: *const rrt_tri_shade,          // offset 32
    pub n_spheres: u32,                           // offset 40
    pub spheres: *const rrt_sphere,               // offset 48
    pub n_media: u32,                             // offset 56
    pub media: *const rrt_medium,                 // offset 64
    pub n_boundary_tris: u32,                     // offset 72
    pub boundary_tris: *const rrt_slot,           // offset 80
    pub boundary_tri_shade: *const rrt_tri_shade, // offset 88
    pub n_materials: u32,                         // offset 96
    pub materials: *const rrt_material,           // offset 104
    pub n_textures: u32,                          // offset 112
    pub textures: *const rrt_texture,             // offset 120
    pub texel_bytes: u64,                         // offset 128
    pub texels: *const u8,                        // offset 136
    pub n_lights: u32,                            // offset 144
    pub lights: *const rrt_light,                 // offset 152
    pub camera: rrt_camera,                       // offset 160
    pub resolution_x: u32,                        // offset 236
    pub resolution_y: u32,                        // offset 240
    pub samples: u32,                             // offset 244
    pub bvh_depth: u32,                           // offset 248
    pub aggregate_kind: u32,                      // offset 252
}
const _: () = assert!(size_of::<rrt_scene_desc>() == 256);

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_render_params {
    pub width: u32,        // offset 0
    pub height: u32,       // offset 4
    pub samples: u32,      // offset 8
    pub first_sample: u32, // offset 12
    pub seed: u32,         // offset 16
    pub background: u32,   // offset 20
    pub max_depth: i32,    // offset 24
    pub aabb_mode: u32,    // offset 28
    pub tile_count: u32,   // offset 32
    pub tile_index: u32,   // offset 36
    pub flags: u32,        // offset 40
}
const _: () = assert!(size_of::<rrt_render_params>() == 44);

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rrt_stats {
    pub rays: u64,               // offset 0
    pub paths: u64,              // offset 8
    pub render_seconds: f64,     // offset 16
    pub trace_seconds: f64,      // offset 24
    pub accumulate_seconds: f64, // offset 32
    pub reduce_seconds: f64,     // offset 40
    pub copy_seconds: f64,       // offset 48
}
const _: () = assert!(size_of::<rrt_stats>() == 56);

extern "C" {
    // ---- host side: one call per scene.rs deserialize_* function ------------------------------------------------------
    /// ≙ `scene::deserialize` (scene.rs:64-68) + flattening.  0 / NaN keep the file's Logistics.
    pub fn rrt_host_scene_load(path: *const c_char, width: u32, height: u32, samples: u32, aspect: f32, out: *mut *mut rrt_host_scene) -> c_int;
    pub fn rrt_host_scene_load_as(path: *const c_char, width: u32, height: u32, samples: u32, aspect: f32, bvh_kind: c_int,
                                  out: *mut *mut rrt_host_scene) -> c_int;
    pub fn rrt_host_scene_new(out: *mut *mut rrt_host_scene) -> c_int;
    /// scene.rs:189-261
    pub fn rrt_host_add_texture(s: *mut rrt_host_scene, t: *const rrt_texture, texels: *const u8, id: *mut i32) -> c_int;
    /// scene.rs:263-407
    pub fn rrt_host_add_material(s: *mut rrt_host_scene, type_: u32, albedo: i32, bump: i32, param: f32, id: *mut i32) -> c_int;
    /// scene.rs:449-470; `translate`, `rotate`, `scale` point at 3 floats each
    pub fn rrt_host_add_sphere(s: *mut rrt_host_scene, translate: *const f32, rotate: *const f32, scale: *const f32, radius: f32, material: i32) -> c_int;
    /// scene.rs:503-561: one .obj object, vertices in object space (f64 like wavefront_obj), 3 indices per triangle
    pub fn rrt_host_add_mesh(s: *mut rrt_host_scene, translate: *const f32, rotate: *const f32, scale: *const f32, n_vertices: u32,
                             vertices: *const f64, n_tex: u32, tex: *const f64, n_triangles: u32, vidx: *const i32, tidx: *const i32, cull: c_int,
                             material: i32) -> c_int;
    /// scene.rs:573-602
    pub fn rrt_host_add_medium_sphere(s: *mut rrt_host_scene, translate: *const f32, rotate: *const f32, scale: *const f32, radius: f32,
                                      boundary_material: i32, density: f32, phase_material: i32) -> c_int;
    /// scene.rs:573-602 with a ConstantMedium for a boundary: the medium added last becomes the boundary of a new one
    pub fn rrt_host_wrap_last_medium(s: *mut rrt_host_scene, density: f32, phase_material: i32) -> c_int;
    /// camera.rs:66-88
    pub fn rrt_host_set_camera(s: *mut rrt_host_scene, position: *const f32, look_at: *const f32, up: *const f32, fov: f32, aspect: f32,
                               aperture: f32, focus_distance: f32) -> c_int;
    /// scene.rs:28-34
    pub fn rrt_host_set_logistics(s: *mut rrt_host_scene, resolution_x: u32, resolution_y: u32, samples: u32, use_importance_sampling: c_int) -> c_int;
    /// scene.rs:130-140, 605-619; aggregate.rs:187-322
    pub fn rrt_host_finalize(s: *mut rrt_host_scene, aggregate: c_int) -> c_int;
    pub fn rrt_host_desc(s: *const rrt_host_scene) -> *const rrt_scene_desc;
    pub fn rrt_host_scene_times(s: *const rrt_host_scene, load_seconds: *mut f64, build_seconds: *mut f64) -> c_int;
    pub fn rrt_host_scene_destroy(s: *mut rrt_host_scene);

    // ---- device side ---------------------------------------------------------------------------------------------------
    /// ≙ moving `Scene` into an `Arc` (main.rs:85-95): uploads to the current device's HBM
    pub fn rrt_scene_create(desc: *const rrt_scene_desc, out: *mut *mut rrt_scene) -> c_int;
    pub fn rrt_scene_create_on(desc: *const rrt_scene_desc, device: c_int, out: *mut *mut rrt_scene) -> c_int;
    pub fn rrt_scene_destroy(s: *mut rrt_scene);
    /// The CPU half of `rrt_scene_create` alone (checks, device records); `layout`: 8 values or null.
    pub fn rrt_scene_check(desc: *const rrt_scene_desc, layout: *mut u64) -> c_int;
    pub fn rrt_scene_upload_seconds(s: *const rrt_scene, seconds: *mut f64) -> c_int;
    pub fn rrt_scene_upload_breakdown(s: *const rrt_scene, first_device_call: *mut f64, code_object: *mut f64, host_preparation: *mut f64, copies: *mut f64) -> c_int;
    pub fn rrt_device_count(count: *mut c_int) -> c_int;
    /// ≙ `Aggregate::hit` (aggregate.rs:118-124) for n rays
    pub fn rrt_intersect_batch(s: *mut rrt_scene, o_xyz: *const f32, d_xyz: *const f32, n: usize, t_min: f32, t_max: f32, aabb_mode: u32, seed: u32,
                               prim_id: *mut i32, t_hit: *mut f32) -> c_int;
    pub fn rrt_intersect_batch_device(s: *mut rrt_scene, d_o_xyz: *const f32, d_d_xyz: *const f32, n: usize, t_min: f32, t_max: f32, aabb_mode: u32,
                                      seed: u32, d_prim_id: *mut i32, d_t_hit: *mut f32, stream: *mut c_void) -> c_int;
    /// ≙ queue fill + `thread_work` + join (main.rs:150-165, 200-257)
    pub fn rrt_render(s: *mut rrt_scene, params: *const rrt_render_params, rgb_sum: *mut f32, stats: *mut rrt_stats) -> c_int;
    pub fn rrt_render_device(s: *mut rrt_scene, params: *const rrt_render_params, d_rgb_sum: *mut f32, d_counters: *mut u64, stream: *mut c_void) -> c_int;
    pub fn rrt_render_device_noalloc(s: *mut rrt_scene, params: *const rrt_render_params, d_rgb_sum: *mut f32, d_counters: *mut u64,
                                     stream: *mut c_void) -> c_int;
    /// ≙ the thread pool of main.rs:136-165 over the GPUs of one node
    pub fn rrt_render_multi(scenes: *const *mut rrt_scene, n: u32, shard_mode: u32, params: *const rrt_render_params, rgb_sum: *mut f32,
                            stats: *mut rrt_stats) -> c_int;
    pub fn rrt_scene_set_staging_limit(s: *mut rrt_scene, bytes: u64) -> c_int;
    pub fn rrt_scene_set_lds_stack(s: *mut rrt_scene, entries: u32) -> c_int;
    pub fn rrt_scene_set_pooling(s: *mut rrt_scene, enable: c_int) -> c_int;
    pub fn rrt_scene_set_timing(s: *mut rrt_scene, enable: c_int) -> c_int;
    pub fn rrt_scene_kernel_times(s: *mut rrt_scene, trace_ms: *mut f64, accumulate_ms: *mut f64, passes: *mut u32) -> c_int;
    /// ≙ main.rs:173-186
    pub fn rrt_resolve_rgb8(rgb_sum: *const f32, width: u32, height: u32, spp: u32, out: *mut u8) -> c_int;
    pub fn rrt_resolve_rgb8_device(d_rgb_sum: *const f32, width: u32, height: u32, spp: u32, d_out: *mut u8, stream: *mut c_void) -> c_int;
    /// ≙ `PNGEncoder::encode` on a `create_new` file (main.rs:98-107, 188)
    pub fn rrt_write_png(path: *const c_char, rgb: *const u8, width: u32, height: u32, create_new: c_int) -> c_int;
    pub fn rrt_probe(s: *mut rrt_scene, which: u32, arg: u32, input: *const c_void, n: usize, out: *mut f32) -> c_int;
    pub fn rrt_selftest_reciprocal(mismatches: *mut u64) -> c_int;
    pub fn rrt_last_error() -> *const c_char;
    pub fn rrt_version() -> *const c_char;
}
