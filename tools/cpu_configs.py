#!/usr/bin/env python3
"""The CPU oracle (the C++ restatement of the reference's path, oracle/) on the workload configurations, at 2 threads (the
reference's default, main.rs:66-72) and at min(16, cores) threads: one JSON record per configuration.  Frames are rendered at a
few samples per pixel (stated in the record); the rate does not depend on the sample count.
   python tools/cpu_configs.py [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

cores = min(16, len(os.sched_getaffinity(0)))
out = []
for name, scene, W, H, spp, bg in (("C1 Cornell_Box 512x512@64", "Cornell_Box.json", 512, 512, 8, 0),
                                   ("C2 Cornell_Box_With_Sphere 1024x1024@1024", "Cornell_Box_With_Sphere.json", 1024, 1024, 4, 0),
                                   ("C3 Materials 1920x1080@1024", "Materials.json", 1920, 1080, 2, 1),
                                   ("C3b Example_Spec 1920x1080@256", "Example_Spec.json", 1920, 1080, 2, 1),
                                   ("C4 Globe 3840x2160@2048", "Globe.json", 3840, 2160, 1, 1)):
    o = orc.OracleScene.from_file(orc.asset(scene), {"resolution_x": W, "resolution_y": H, "aspect_ratio": W / H})
    rec = {"config": name, "width": W, "height": H, "spp_rendered": spp}
    for threads in (2, cores):
        t0 = time.perf_counter()
        _, st = o.render(W, H, spp, seed=0, bg=bg, threads=threads)
        dt = time.perf_counter() - t0
        rec["mrays_per_s_%d_threads" % threads] = int(st[0]) / dt / 1e6
        rec["rays"] = int(st[0])
    out.append(rec)
    print(json.dumps(rec), flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1)
