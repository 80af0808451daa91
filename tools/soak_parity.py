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
    # the ray-batch entry point: random rays from inside and around the scene, all three modes
    bb = o.shape_bboxes()
    extent = float(np.abs(bb[np.isfinite(bb).all(axis=1)]).max())
    rng = np.random.default_rng(5)
    n = 4_000_000
    ro = rng.uniform(-1.5 * extent, 1.5 * extent, (n, 3)).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd[: n // 8, rng.integers(0, 3)] = 0.0  # some inside a coordinate plane
    t1 = time.perf_counter()
    p_ref, t_ref = o.intersect(ro, rd)
    for mode, label in ((rrt.AABB_TIGHT, "tight"), (rrt.AABB_REFERENCE, "reference"), (rrt.AABB_EXHAUSTIVE, "exhaustive")):
        p, t = scene.hit(ro, rd, aabb_mode=mode)
        diff = int(((p != p_ref) | (t.view(np.uint32) != t_ref.view(np.uint32))).sum())
        bad += 1 if diff else 0
        print("%-30s %d random rays, %-10s: %d hits (oracle %d), %d answers differ  %s" % (name, n, label, int((p >= 0).sum()), int((p_ref >= 0).sum()), diff,
                                                                                       "ok" if diff == 0 else "DIFFERENT"), flush=True)
    print("  (oracle: %.1f s)" % (time.perf_counter() - t1), flush=True)
sys.exit(1 if bad else 0)
