#!/usr/bin/env python3
"""Times the SAH BVH construction on the host and on the GPU for displaced-grid meshes and checks that the two trees are equal:
   python tools/perf_build.py [grid sizes ...]   (a grid of n has 2*n*n triangles)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import synth  # noqa: E402

torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
for n in [int(a) for a in sys.argv[1:]] or [100, 500]:
    d = synth.terrain_desc(n, seed=2)
    times = {}
    scenes = {}
    for mode in ("BVH_GPU", "BVH_GPU", "BVH"):  # the first device build also pays for loading the kernels
        if mode == "BVH" and n > 1500 and "--host" not in sys.argv:
            continue
        t0 = time.perf_counter()
        hs = synth.host_scene(d, mode)
        times[mode] = time.perf_counter() - t0
        scenes[mode] = hs
    same = ""
    if "BVH" in scenes:
        a, b = scenes["BVH"], scenes["BVH_GPU"]
        same = "equal" if a.nodes().tobytes() == b.nodes().tobytes() and a.slots().tobytes() == b.slots().tobytes() else "DIFFERENT"
    print("grid %d: %d triangles, %d nodes, depth %d; scene set-up incl. tree: device %.2f s, host %s  %s" % (
        n, 2 * n * n + 2, len(scenes["BVH_GPU"].nodes()), scenes["BVH_GPU"].desc.bvh_depth, times["BVH_GPU"],
        ("%.2f s" % times["BVH"]) if "BVH" in times else "skipped", same), flush=True)
