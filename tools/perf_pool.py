#!/usr/bin/env python3
"""Pooled schedule (k_trace_pool) against one path per lane (k_trace): same frame bit for bit, and the time of each.
python tools/perf_pool.py [scene.json] [W] [H] [spp] [bg] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rust_raytracer_amd as rrt
name = sys.argv[1] if len(sys.argv) > 1 else "Cornell_Box_With_Sphere.json"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 128
bg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
torch.cuda.set_device(0)
sc = rrt.Scene.load(os.path.join(ROOT, "tests", "golden", "assets", name), width=W, height=H, samples=spp)
st = torch.cuda.current_stream()
frames = {}
for pooled in (False, True):
    sc.set_pooling(pooled)
    img = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    sc.render_device(img.data_ptr(), cnt.data_ptr(), st.cuda_stream, background=bg)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        cnt.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        sc.render_device(img.data_ptr(), cnt.data_ptr(), st.cuda_stream, background=bg)
        e1.record(st)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    rays = int(cnt[0].item())
    print("%s %dx%d@%d %s: %.2f ms (%s)  rays=%d paths=%d  %.2f Grays/s" % (name, W, H, spp, "pooled  " if pooled else "lockstep", min(ts),
          " ".join("%.1f" % t for t in ts), rays, int(cnt[1].item()), rays / min(ts) / 1e6), flush=True)
    frames[pooled] = (img.cpu().numpy().view("uint32"), rays)
same = (frames[0][0] == frames[1][0]).all() and frames[0][1] == frames[1][1]
print("bit-identical frames and ray counts:", bool(same))
sys.exit(0 if same else 1)
