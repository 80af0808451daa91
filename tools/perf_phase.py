#!/usr/bin/env python3
"""Per-phase cycle split of k_trace from a library built with -DRRT_PHASE_PROFILE (RRT_LIB points at it):
python tools/perf_phase.py SCENE W H SPP BG"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rust_raytracer_amd as rrt
name, W, H, spp, bg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
sc = rrt.Scene.load(os.path.join(ROOT, "tests", "golden", "assets", name + ".json"), width=W, height=H, samples=spp)
img = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
cnt = torch.zeros(16, dtype=torch.int64, device="cuda")
sc.render_device(img.data_ptr(), cnt.data_ptr(), 0, samples=spp, background=bg)
torch.cuda.synchronize()
c = cnt.cpu().numpy()
tot = float(c[2] + c[3] + c[4] + c[5])
print("rays %d paths %d wave-iterations %d" % (c[0], c[1], c[6]))
for i, n in enumerate(["regen", "traverse", "shade", "store+pull"]):
    print("%-10s %6.2f %%  %8.0f cycles per wave-iteration" % (n, 100.0 * c[2 + i] / tot, c[2 + i] / c[6]))
print("fresh lanes per iteration %.2f, valid lanes %.2f, done lanes %.2f" % (c[7] / c[6], c[8] / c[6], c[9] / c[6]))
L = rrt.lib()
if hasattr(L, "rrt_profile_read"):
    import ctypes
    import numpy as np
    out = np.zeros(8, dtype=np.uint64)
    L.rrt_profile_read(ctypes.c_void_p(out.ctypes.data))
    if out[7]:
        print("sift: %.2f filter steps, %.1f items, %.2f dense groups per phase; %.2f phases per wave-iteration" % (
            out[4] / out[7], out[5] / out[7], out[6] / out[7], out[7] / c[6]))
