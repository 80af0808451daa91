#!/usr/bin/env python3
"""Per-kind split of k_trace_pool from a library built with -DRRT_PHASE_PROFILE (RRT_LIB points at it):
python tools/perf_pool_phase.py SCENE W H SPP BG"""
import ctypes, os, sys
import numpy as np
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
K = 5
out = np.zeros(3 * K, dtype=np.uint64)
rrt.lib().rrt_profile_read_pool(ctypes.c_void_p(out.ctypes.data))
steps, lanes, cyc = out[:K].astype(float), out[K:2 * K].astype(float), out[2 * K:3 * K].astype(float)
c = cnt.cpu().numpy()
print("rays %d paths %d" % (c[0], c[1]))
for i, nm in enumerate(["FREE", "TRACE", "GEN", "LTRI", "LCS"]):
    if steps[i]:
        print("%-6s steps %10d  paths/step %5.1f  cycles %5.1f %%  cycles/step %8.0f" % (nm, steps[i], lanes[i] / steps[i], 100 * cyc[i] / cyc.sum(), cyc[i] / steps[i]))
print("mean paths per step %.1f; steps per ray %.3f" % (lanes.sum() / steps.sum(), steps.sum() / c[0]))
h = np.zeros(20, dtype=np.uint64)
rrt.lib().rrt_profile_read_sift(ctypes.c_void_p(h.ctypes.data))
h = h.astype(float)
if h[:16].sum():
    print("candidate lists by size (items, share of queries): " + "  ".join("%d-%d: %.1f%%" % (16 * i, 16 * i + 15, 100 * h[i] / h[:16].sum()) for i in range(16) if h[i]))
    print("query cycles: sifting %.1f %%, dense groups %.1f %%; rays per query %.1f; cycles per query: sifting %.0f, dense %.0f" % (
        100 * h[16] / (h[16] + h[17]), 100 * h[17] / (h[16] + h[17]), h[18] / h[:16].sum(), h[16] / h[:16].sum(), h[17] / h[:16].sum()))
