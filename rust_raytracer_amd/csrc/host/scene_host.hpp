// Host-side scene under construction: the objects scene::deserialize (scene.rs:64-159) creates,
// kept as plain arrays, plus the flattened device layout produced by finalize().
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/rrt.h"
#include "hmath.hpp"

namespace rrt_host {

struct HostShape {
    uint32_t kind;        // RRT_KIND_*
    // triangle
    Vec3 v0, v1, v2;      // world space (scene.rs:506-508)
    float uv[6];          // uv0, uv1, uv2 with shape.rs:190-210 defaults
    Vec3 pu, pv;          // shape.rs:278-314
    bool cull;
    // sphere / medium
    uint32_t index;       // spheres[] or media[]
    uint32_t material;
    Box box;              // Shape::get_bounding_box
};

struct SceneBuild {
    std::vector<rrt_texture> textures;
    std::vector<uint8_t> texels;
    std::vector<rrt_material> materials;
    std::vector<rrt_sphere> spheres;
    std::vector<rrt_medium> media;
    std::vector<HostShape> boundary_shapes;  // triangle boundaries of media
    std::vector<HostShape> shapes;           // scene order
    rrt_camera camera{};
    uint32_t res_x = 0, res_y = 0, samples = 0;
    bool use_importance = true;
    bool finalized = false;
    double load_seconds = 0.0, build_seconds = 0.0;  // host wall clock: reading + decoding the scene file; finalize()

    // flattened
    std::vector<rrt_node> nodes;
    std::vector<rrt_slot> slots;
    std::vector<rrt_tri_shade> tri_shade;
    std::vector<rrt_slot> boundary_tris;
    std::vector<rrt_tri_shade> boundary_tri_shade;
    std::vector<rrt_light> lights;
    rrt_scene_desc desc{};

    // Each returns "" or the error text.
    std::string add_texture(const rrt_texture& t, const uint8_t* px, int32_t* id);
    std::string add_material(uint32_t type, int32_t albedo, int32_t bump, float param, int32_t* id);
    std::string add_sphere(Vec3 tr, Vec3 rot, Vec3 scl, float radius, int32_t material);
    std::string add_mesh(Vec3 tr, Vec3 rot, Vec3 scl, uint32_t nv, const double* v, uint32_t nt, const double* t, uint32_t ntri,
                         const int32_t* vidx, const int32_t* tidx, bool cull, int32_t material, bool as_boundary);
    std::string add_medium_sphere(Vec3 tr, Vec3 rot, Vec3 scl, float radius, int32_t boundary_material, float density, int32_t phase);
    std::string add_medium_last_boundary_triangle(float density, int32_t phase);
    std::string wrap_last_medium(float density, int32_t phase);
    void set_camera(Vec3 pos, Vec3 look_at, Vec3 up, float fov, float aspect, float aperture, float focus);
    std::string finalize(int aggregate);  // RRT_AGGREGATE_*
};

// scene file -> SceneBuild (scene.rs:64-159); returns "" or the error text
// tree: what the file's "BVH" aggregate becomes (RRT_AGGREGATE_BVH, _BVH_GPU or _BVH_CLEAN)
std::string load_scene_file(const std::string& path, uint32_t ow, uint32_t oh, uint32_t os, float oaspect, int tree, SceneBuild* out);

// Decodes an image file to 8-bit texels with 1 (grey) or 3 (RGB) channels; returns "" or the error text.
std::string decode_image_file(const std::string& path, uint32_t* w, uint32_t* h, uint32_t* channels, std::vector<uint8_t>* px);

}  // namespace rrt_host
