#!/usr/bin/env python3
"""Quick kernel timing on the GPU box: renders a scene a few times and prints ms / Grays/s.
   python tools/perf_quick.py [scene.json] [W] [H] [spp] [bg] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rust_raytracer_amd as rrt  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "Cornell_Box_With_Sphere.json"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 128
bg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
torch.cuda.set_device(0)
sc = rrt.Scene.load(os.path.join(ROOT, "tests", "golden", "assets", name), width=W, height=H, samples=spp)
if os.environ.get("RRT_POOLING") == "0":
    sc.set_pooling(False)  # one path per lane (k_trace) instead of the pooled schedule
img = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream()
for mode in [int(m) for m in os.environ.get("RRT_MODES", "0").split(",")]:  # 0 tight, 1 reference, 2 exhaustive
    sc.render_device(img.data_ptr(), cnt.data_ptr(), st.cuda_stream, background=bg, aabb_mode=mode)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        cnt.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        sc.render_device(img.data_ptr(), cnt.data_ptr(), st.cuda_stream, background=bg, aabb_mode=mode)
        e1.record(st)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    rays = int(cnt[0].item())
    best = min(ts)
    print("%s %dx%d@%d mode=%d: %.2f ms (min of %s)  rays=%d  %.2f Grays/s  rays/path=%.3f" % (
        name, W, H, spp, mode, best, ["%.1f" % t for t in ts], rays, rays / best / 1e6, rays / (W * H * spp)))
