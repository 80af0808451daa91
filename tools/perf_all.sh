#!/bin/bash
# GPU-box timing of the three small scenes with the library RRT_LIB points at (default: the in-tree build).
set -e
timeout -k 10 120 python tools/perf_quick.py Cornell_Box_With_Sphere.json 1024 1024 128 0 3
timeout -k 10 120 python tools/perf_quick.py Cornell_Box.json 1024 1024 128 0 3
timeout -k 10 120 python tools/perf_quick.py Materials.json 1920 1080 32 1 3
