#!/usr/bin/env python3
"""How long does a small frame of the 10 M-triangle mesh take with the reference's box test alone (RRT_AABB_REFERENCE)?
python tools/probe_c5_reference_render.py [W] [H]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import rust_raytracer_amd as rrt
import synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
t0 = time.time()
d = synth.terrain_desc(2236, seed=1, width=64, height=64, samples=1)
g = rrt.Scene(synth.host_scene(d, "BVH_GPU"))
print("scene ready %.1f s" % (time.time() - t0), flush=True)
for mode in (rrt.AABB_TIGHT, rrt.AABB_REFERENCE):
    for (w, h) in ((8, 8), (W, H)):
        t = time.time()
        img, st = g.render(width=w, height=h, samples=1, seed=3, aabb_mode=mode)
        print("mode %d %dx%d: %.2f s, %d rays, kernel %.2f s" % (mode, w, h, time.time() - t, st["rays"], st["trace_seconds"]), flush=True)
