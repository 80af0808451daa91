#!/usr/bin/env python3
"""Renders the headline frame (Cornell_Box_With_Sphere, 1024x1024) on the GPU and with the CPU oracle at the same seed and compares
the radiance sums bit for bit: python tools/verify_headline.py [spp]   (the oracle needs about 0.15 s per spp on 16 cores)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401

import orc  # noqa: E402
import rust_raytracer_amd as rrt  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W = H = 1024
path = orc.asset("Cornell_Box_With_Sphere.json")
t0 = time.perf_counter()
img, st = rrt.Scene.load(path, width=W, height=H, samples=spp).render(seed=0)
t1 = time.perf_counter()
threads = min(16, len(os.sched_getaffinity(0)))
print("GPU frame done in %.2f s; CPU oracle on %d threads ..." % (t1 - t0, threads), flush=True)
import threading  # noqa: E402

done = threading.Event()


def heartbeat():  # a long silent run looks hung to whoever watches the job
    while not done.wait(30.0):
        print("  oracle still rendering, %.0f s" % (time.perf_counter() - t1), flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
ref, st_ref = orc.OracleScene.from_file(path, {"resolution_x": W, "resolution_y": H, "samples": spp}).render(W, H, spp, seed=0, threads=threads)
t2 = time.perf_counter()
done.set()
same = (img.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(img) & np.isnan(ref))
print("1024x1024 @ %d spp: GPU %d rays, oracle %d rays; %d of %d radiance sums differ; %d NaN sums; GPU call %.2f s, oracle %.1f s on %d threads" % (
    spp, st["rays"], int(st_ref[0]), int((~same).sum()), same.size, int(np.isnan(ref).sum()), t1 - t0, t2 - t1, threads))
sys.exit(0 if same.all() and st["rays"] == int(st_ref[0]) else 1)
