for lib in rust_raytracer_amd/librrt_hip.so rust_raytracer_amd/librrt_var.so; do
  echo "== $lib"
  for a in "Cornell_Box.json 1024 1024 128 0" "Materials.json 1920 1080 64 1" "Globe.json 1920 1080 32 1" "Example_Spec.json 1920 1080 32 1"; do
    RRT_LIB=$lib timeout -k 10 100 python -u tools/perf_quick.py $a 2>&1 | grep -v amdgpu.ids || exit 1
  done
  RRT_LIB=$lib timeout -k 10 200 python -u tools/perf_mesh.py 500 1024 1024 16 tight clean 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
done
