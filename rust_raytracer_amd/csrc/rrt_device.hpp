// Device-side building blocks of the path tracer (gfx950), in five parts:
//   rrt_math.hpp        3-vectors, deterministic elementary functions, per-path random stream
//   rrt_scene_view.hpp  device view of the scene, samplers, Perlin noise, textures
//   rrt_primitives.hpp  exact reciprocal, triangle / sphere / medium tests, bounding-box tests
//   rrt_traversal.hpp   closest-hit queries (pair test, cooperative leaves, node walk, sifting)
//   rrt_shading.hpp     hit properties, light mixture, materials, one step of aggregate::trace
// (rrt_profile.hpp: the probes of the profiling build, empty otherwise)
#pragma once
#include "rrt_shading.hpp"
