// Compile-time statement of the C ABI's struct layouts (include/rrt.h): the sizes and offsets below are the numbers the Rust
// bindings (rrt-sys/src/lib.rs) and the Python bindings (rust_raytracer_amd/__init__.py) are written against.  A change of a
// struct that moves a field fails this file's compilation until all three are brought in line.
#include <cstddef>

#include "../../include/rrt.h"

#define RRT_SIZE(T, n) static_assert(sizeof(T) == (n), "sizeof(" #T ") changed: update rrt-sys/src/lib.rs and __init__.py")
#define RRT_AT(T, f, n) static_assert(offsetof(T, f) == (n), "offsetof(" #T ", " #f ") changed: update rrt-sys/src/lib.rs and __init__.py")

RRT_SIZE(rrt_node, 32);
RRT_AT(rrt_node, bmin, 0); RRT_AT(rrt_node, a, 12); RRT_AT(rrt_node, bmax, 16); RRT_AT(rrt_node, b, 28);
RRT_SIZE(rrt_slot, 48);
RRT_AT(rrt_slot, v0, 0); RRT_AT(rrt_slot, prim, 12); RRT_AT(rrt_slot, e1, 16); RRT_AT(rrt_slot, info, 28); RRT_AT(rrt_slot, e2, 32); RRT_AT(rrt_slot, aux, 44);
RRT_SIZE(rrt_tri_shade, 64);
RRT_AT(rrt_tri_shade, nrm, 0); RRT_AT(rrt_tri_shade, pad, 12); RRT_AT(rrt_tri_shade, uv0, 16); RRT_AT(rrt_tri_shade, uv1, 24); RRT_AT(rrt_tri_shade, uv2, 32);
RRT_AT(rrt_tri_shade, pu, 40); RRT_AT(rrt_tri_shade, pv, 52);
RRT_SIZE(rrt_sphere, 128);
RRT_AT(rrt_sphere, w2l, 0); RRT_AT(rrt_sphere, l2w, 48); RRT_AT(rrt_sphere, radius, 96); RRT_AT(rrt_sphere, material, 100); RRT_AT(rrt_sphere, pad, 104);
RRT_SIZE(rrt_medium, 16);
RRT_AT(rrt_medium, boundary_kind, 0); RRT_AT(rrt_medium, boundary_index, 4); RRT_AT(rrt_medium, density, 8); RRT_AT(rrt_medium, material, 12);
RRT_SIZE(rrt_material, 16);
RRT_AT(rrt_material, type, 0); RRT_AT(rrt_material, albedo, 4); RRT_AT(rrt_material, bump, 8); RRT_AT(rrt_material, param, 12);
RRT_SIZE(rrt_texture, 32);
RRT_AT(rrt_texture, type, 0); RRT_AT(rrt_texture, a, 4); RRT_AT(rrt_texture, b, 8); RRT_AT(rrt_texture, c, 12); RRT_AT(rrt_texture, u0, 16);
RRT_AT(rrt_texture, u1, 20); RRT_AT(rrt_texture, u2, 24); RRT_AT(rrt_texture, u3, 28);
RRT_SIZE(rrt_light, 96);
RRT_AT(rrt_light, kind, 0); RRT_AT(rrt_light, index, 4); RRT_AT(rrt_light, cull, 8); RRT_AT(rrt_light, area, 12); RRT_AT(rrt_light, v0, 16);
RRT_AT(rrt_light, v1, 28); RRT_AT(rrt_light, v2, 40); RRT_AT(rrt_light, e1, 52); RRT_AT(rrt_light, e2, 64); RRT_AT(rrt_light, nrm, 76); RRT_AT(rrt_light, pad, 88);
RRT_SIZE(rrt_camera, 76);
RRT_AT(rrt_camera, origin, 0); RRT_AT(rrt_camera, lower_left_corner, 12); RRT_AT(rrt_camera, horizontal, 24); RRT_AT(rrt_camera, vertical, 36);
RRT_AT(rrt_camera, u, 48); RRT_AT(rrt_camera, v, 60); RRT_AT(rrt_camera, lens_radius, 72);
RRT_SIZE(rrt_scene_desc, 256);
RRT_AT(rrt_scene_desc, n_nodes, 0); RRT_AT(rrt_scene_desc, nodes, 8); RRT_AT(rrt_scene_desc, n_slots, 16); RRT_AT(rrt_scene_desc, slots, 24);
RRT_AT(rrt_scene_desc, tri_shade, 32); RRT_AT(rrt_scene_desc, n_spheres, 40); RRT_AT(rrt_scene_desc, spheres, 48); RRT_AT(rrt_scene_desc, n_media, 56);
RRT_AT(rrt_scene_desc, media, 64); RRT_AT(rrt_scene_desc, n_boundary_tris, 72); RRT_AT(rrt_scene_desc, boundary_tris, 80);
RRT_AT(rrt_scene_desc, boundary_tri_shade, 88); RRT_AT(rrt_scene_desc, n_materials, 96); RRT_AT(rrt_scene_desc, materials, 104);
RRT_AT(rrt_scene_desc, n_textures, 112); RRT_AT(rrt_scene_desc, textures, 120); RRT_AT(rrt_scene_desc, texel_bytes, 128); RRT_AT(rrt_scene_desc, texels, 136);
RRT_AT(rrt_scene_desc, n_lights, 144); RRT_AT(rrt_scene_desc, lights, 152); RRT_AT(rrt_scene_desc, camera, 160); RRT_AT(rrt_scene_desc, resolution_x, 236);
RRT_AT(rrt_scene_desc, resolution_y, 240); RRT_AT(rrt_scene_desc, samples, 244); RRT_AT(rrt_scene_desc, bvh_depth, 248); RRT_AT(rrt_scene_desc, aggregate_kind, 252);
RRT_SIZE(rrt_render_params, 44);
RRT_AT(rrt_render_params, width, 0); RRT_AT(rrt_render_params, height, 4); RRT_AT(rrt_render_params, samples, 8); RRT_AT(rrt_render_params, first_sample, 12);
RRT_AT(rrt_render_params, seed, 16); RRT_AT(rrt_render_params, background, 20); RRT_AT(rrt_render_params, max_depth, 24); RRT_AT(rrt_render_params, aabb_mode, 28);
RRT_AT(rrt_render_params, tile_count, 32); RRT_AT(rrt_render_params, tile_index, 36); RRT_AT(rrt_render_params, flags, 40);
RRT_SIZE(rrt_stats, 56);
RRT_AT(rrt_stats, rays, 0); RRT_AT(rrt_stats, paths, 8); RRT_AT(rrt_stats, render_seconds, 16); RRT_AT(rrt_stats, trace_seconds, 24);
RRT_AT(rrt_stats, accumulate_seconds, 32); RRT_AT(rrt_stats, reduce_seconds, 40); RRT_AT(rrt_stats, copy_seconds, 48);
