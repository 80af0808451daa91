#!/usr/bin/env python3
"""Finds the rays of a random batch that dominate rrt_intersect_batch_device's time on the synthetic mesh (clean tree): times the
batch in ever smaller parts.   python tools/find_slow_ray.py N n_rays"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C

import torch

import rust_raytracer_amd as rrt
import synth

N, n = int(sys.argv[1]), int(sys.argv[2])
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(1)
sc = rrt.Scene(synth.host_scene(synth.terrain_desc(N), "BVH_CLEAN"))
o = torch.rand((n, 3), device="cuda", generator=g) * 2.0 - 1.0
o[:, 1] = o[:, 1] * 0.25 + 0.6
d = torch.randn((n, 3), device="cuda", generator=g)
d[:, 1] = -d[:, 1].abs() * 0.5
prim = torch.empty(n, dtype=torch.int32, device="cuda")
t = torch.empty(n, dtype=torch.float32, device="cuda")
L = rrt.lib()
L.rrt_intersect_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
st = torch.cuda.current_stream()


def timed(lo, hi):
    best = 1e9
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        rc = L.rrt_intersect_batch_device(sc._h, o[lo:hi].data_ptr(), d[lo:hi].data_ptr(), hi - lo, 0.001, 3.4e38, 0, 0, prim.data_ptr(), t.data_ptr(), st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        assert rc == 0
        best = min(best, e0.elapsed_time(e1))
    return best


lo, hi = 0, n
print("all %d rays: %.2f ms" % (n, timed(lo, hi)), flush=True)
while hi - lo > 1:
    parts = 16 if hi - lo >= 16 else hi - lo
    step = (hi - lo + parts - 1) // parts
    times = [(timed(a, min(a + step, hi)), a) for a in range(lo, hi, step)]
    times.sort(reverse=True)
    print("  parts of %d: slowest %.3f ms at %d, median %.3f ms" % (step, times[0][0], times[0][1], times[len(times) // 2][0]), flush=True)
    lo, hi = times[0][1], min(times[0][1] + step, hi)
print("ray %d: o = %r d = %r -> prim %d" % (lo, o[lo].tolist(), d[lo].tolist(), int(prim[0])))
print("bits o", [hex(x) for x in o[lo].view(torch.int32).tolist()], "d", [hex(x & 0xffffffff) for x in d[lo].view(torch.int32).tolist()])
