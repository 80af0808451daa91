#!/usr/bin/env python3
"""Every asset scene rendered on the GPU in the default and the reference traversal mode and by the CPU oracle, at a size where the
guard's rare cases (a ray in a few hundred) occur by the hundred thousand; radiance sums and ray counts must be the same bits:
   python tools/soak_parity.py [W H SPP]     (default 640 360 64: about 25 M paths per scene)"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401

import orc  # noqa: E402
import rust_raytracer_amd as rrt  # noqa: E402

W, H, spp = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (640, 360, 64)
threads = min(16, len(os.sched_getaffinity(0)))
bad = 0
for name, bg in (("Cornell_Box.json", rrt.BG_BLACK), ("Cornell_Box_With_Sphere.json", rrt.BG_SKY), ("Materials.json", rrt.BG_SKY),
                 ("Example_Spec.json", rrt.BG_SKY), ("Globe.json", rrt.BG_SKY)):
    path = orc.asset(name)
    scene = rrt.Scene.load(path, width=W, height=H, samples=spp, aspect_ratio=W / H)
    frames = {}
    for mode, label in ((rrt.AABB_TIGHT, "tight"), (rrt.AABB_REFERENCE, "reference")):
        frames[label] = scene.render(seed=11, background=bg, aabb_mode=mode)
    t0 = time.perf_counter()
    done = threading.Event()

    def heartbeat():
        while not done.wait(30.0):
            print("  oracle still rendering %s, %.0f s" % (name, time.perf_counter() - t0), flush=True)

    threading.Thread(target=heartbeat, daemon=True).start()
    o = orc.OracleScene.from_file(path, {"resolution_x": W, "resolution_y": H, "samples": spp, "aspect_ratio": W / H})
    ref, st_ref = o.render(W, H, spp, seed=11, bg=bg, threads=threads)
    done.set()
    for label, (img, st) in frames.items():
        same = (img.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(img) & np.isnan(ref))
        ok = same.all() and st["rays"] == int(st_ref[0])
        bad += 0 if ok else 1
        print("%-30s %dx%d @ %d, %-9s: %d rays (oracle %d), %d of %d sums differ, %d NaN sums  %s" % (
            name, W, H, spp, label, st["rays"], int(st_ref[0]), int((~same).sum()), same.size, int(np.isnan(ref).sum()), "ok" if ok else "DIFFERENT"), flush=True)
    print("  (oracle: %.1f s on %d threads)" % (time.perf_counter() - t0, threads), flush=True)
sys.exit(1 if bad else 0)
