#!/usr/bin/env python3
"""Times the render kernel at several max_depth values (0 = primary rays only) to split traversal from shading cost."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rust_raytracer_amd as rrt  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "Cornell_Box_With_Sphere.json"
W = H = 1024
spp = 128
torch.cuda.set_device(0)
sc = rrt.Scene.load(os.path.join(ROOT, "tests", "golden", "assets", name), width=W, height=H, samples=spp)
img = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream()
for md in (0, 1, 2, 3, 5, 50):
    best = 1e9
    for rep in range(3):
        cnt.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        sc.render_device(img.data_ptr(), cnt.data_ptr(), st.cuda_stream, max_depth=md)
        e1.record(st)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    rays = int(cnt[0].item())
    print("max_depth=%2d  %.2f ms  rays=%d  %.2f Grays/s  %.4f ns/ray" % (md, best, rays, rays / best / 1e6, best * 1e6 / rays))
