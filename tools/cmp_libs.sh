#!/bin/bash
# GPU-box timing of two builds of the library side by side (the in-tree one and another, e.g. built by
# build.build(extra_flags=[...], out=".../librrt_var.so")):  bash tools/cmp_libs.sh [other.so]
other=${1:-rust_raytracer_amd/librrt_var.so}
for lib in rust_raytracer_amd/librrt_hip.so "$other"; do
  echo "== $lib"
  for a in "Cornell_Box_With_Sphere.json 1024 1024 128 0" "Cornell_Box.json 1024 1024 128 0" "Materials.json 1920 1080 64 1" "Globe.json 1920 1080 32 1" "Example_Spec.json 1920 1080 32 1"; do
    RRT_LIB=$lib timeout -k 10 100 python -u tools/perf_quick.py $a 2>&1 | grep -v amdgpu.ids || exit 1
  done
  RRT_LIB=$lib timeout -k 10 200 python -u tools/perf_mesh.py 500 1024 1024 16 tight clean 2>&1 | grep -v amdgpu.ids | tail -2 || exit 1
done
