#!/usr/bin/env python3
"""Throughput of rrt_intersect_batch_device (Aggregate::hit for a ray batch): python tools/perf_intersect.py [scene.json|mesh=N[,clean]] [n_rays] [lds=K]
(mesh=N: the synthetic displaced grid of 2*N*N triangles, random rays from above it)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C

import torch

import rust_raytracer_amd as rrt

name = sys.argv[1] if len(sys.argv) > 1 else "Cornell_Box_With_Sphere.json"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 24
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(1)
if name.startswith("mesh="):
    import synth
    spec = name[5:].split(",")
    sc = rrt.Scene(synth.host_scene(synth.terrain_desc(int(spec[0])), "BVH_CLEAN" if "clean" in spec else "BVH_GPU"))
    o = torch.rand((n, 3), device="cuda", generator=g) * 2.0 - 1.0
    o[:, 1] = o[:, 1] * 0.25 + 0.6  # a slab above the terrain
    d = torch.randn((n, 3), device="cuda", generator=g)
    d[:, 1] = -d[:, 1].abs() * 0.5  # looking down at all angles
else:
    sc = rrt.Scene.load(os.path.join(ROOT, "tests", "golden", "assets", name))
    o = torch.rand((n, 3), device="cuda", generator=g) * 500.0 + 20.0
    d = torch.randn((n, 3), device="cuda", generator=g)
for a in sys.argv[3:]:
    if a.startswith("lds="):
        sc.set_lds_stack(int(a[4:]))  # stack entries per ray kept in LDS (deep trees)
prim = torch.empty(n, dtype=torch.int32, device="cuda")
t = torch.empty(n, dtype=torch.float32, device="cuda")
L = rrt.lib()
L.rrt_intersect_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
st = torch.cuda.current_stream()
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    rc = L.rrt_intersect_batch_device(sc._h, o.data_ptr(), d.data_ptr(), n, 0.001, 3.4e38, 0, 0, prim.data_ptr(), t.data_ptr(), st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    assert rc == 0
    ms = e0.elapsed_time(e1)
print("%s: %d rays in %.2f ms = %.2f Grays/s (%.1f %% hit)" % (name, n, ms, n / ms / 1e6, 100.0 * float((prim >= 0).float().mean())))
