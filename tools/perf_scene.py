#!/usr/bin/env python3
"""GPU-box timing of one asset scene: python tools/perf_scene.py SCENE W H SPP BG [reps]  (best of reps, device time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rust_raytracer_amd as rrt
name, W, H, spp, bg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
sc = rrt.Scene.load(os.path.join(ROOT, "tests", "golden", "assets", name + ".json"), width=W, height=H, samples=spp)
img = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream()
best = None
for _ in range(reps + 1):
    cnt.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    sc.render_device(img.data_ptr(), cnt.data_ptr(), st.cuda_stream, background=bg, out_floats=img.numel())
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    best = ms if best is None else min(best, ms)
print("%s %dx%d@%d: %.3f ms  rays=%d  %.2f Grays/s" % (name, W, H, spp, best, int(cnt[0]), int(cnt[0]) / best / 1e6), flush=True)
