#!/bin/bash
# GPU-box A/B of library builds on the headline scene: bash tools/ab_libs.sh name1 name2 ...  (rust_raytracer_amd/librrt_<name>.so)
for t in "$@"; do
  echo "== $t"
  RRT_LIB=$PWD/rust_raytracer_amd/librrt_$t.so timeout -k 10 100 python -u tools/perf_pool.py Cornell_Box_With_Sphere.json 1024 1024 128 0 3 2>&1 | grep -v amdgpu.ids || exit 1
done
