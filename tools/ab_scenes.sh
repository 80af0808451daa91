#!/bin/bash
# GPU-box A/B of library builds over the asset scenes: bash tools/ab_scenes.sh name1 name2 ...  (rust_raytracer_amd/librrt_<name>.so)
for t in "$@"; do
  echo "== $t"
  for a in "Cornell_Box_With_Sphere.json 1024 1024 128 0" "Materials.json 1920 1080 64 1" "Example_Spec.json 1920 1080 32 1"; do
    RRT_LIB=$PWD/rust_raytracer_amd/librrt_$t.so timeout -k 10 100 python -u tools/perf_quick.py $a 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
