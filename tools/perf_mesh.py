#!/usr/bin/env python3
"""GPU-box timing of the synthetic displaced-grid mesh: python tools/perf_mesh.py N W H SPP [tight] [host|clean] [lds=K]  (2*N*N triangles;
"host" builds the tree on the host instead of the GPU)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rust_raytracer_amd as rrt
import synth
n, W, H, spp = (int(a) for a in sys.argv[1:5])
t0 = time.time()
d = synth.terrain_desc(n, width=W, height=H, samples=spp)
torch.zeros(1, device="cuda")
t0 = time.time()
hs = synth.host_scene(d, "BVH" if "host" in sys.argv else ("BVH_CLEAN" if "clean" in sys.argv else "BVH_GPU"))
t1 = time.time()
sc = rrt.Scene(hs)
for a in sys.argv:
    if a.startswith("lds="):
        sc.set_lds_stack(int(a[4:]))  # stack entries per ray kept in LDS
t2 = time.time()
print("triangles %d nodes %d depth %d: scene + tree build %.2f s, upload %.2f s" % (2 * n * n, hs.desc.n_nodes, hs.desc.bvh_depth, t1 - t0, t2 - t1), flush=True)
img = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream()
for mode, name in ((rrt.AABB_TIGHT, "tight"), (rrt.AABB_REFERENCE, "reference")):
    if mode == rrt.AABB_REFERENCE and len(sys.argv) > 5 and sys.argv[5] == "tight":
        continue
    cnt.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    sc.render_device(img.data_ptr(), cnt.data_ptr(), st.cuda_stream, aabb_mode=mode)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    L = rrt.lib()
    if hasattr(L, "rrt_profile_read"):
        import ctypes, numpy as np
        out = np.zeros(8, dtype=np.uint64)
        L.rrt_profile_read(ctypes.c_void_p(out.ctypes.data))
        rays = max(int(cnt[0]), 1)
        print("  per ray: %.1f node visits, %.1f pair tests; per wave-query: node steps %.1f, pair steps %.1f (wave-queries ~ rays/64 = %d)" % (
            out[0] / rays, out[1] / rays, out[2] / (rays / 64), out[3] / (rays / 64), rays // 64))
        cyc = float(out[4] + out[5] + out[6])
        if cyc:
            print("  wave cycles: shading %.1f %% (%.0f per phase), node steps %.1f %% (%.0f per step), leaf phases %.1f %% (%.0f per phase)" % (
                100 * out[4] / cyc, out[4] / max(int(out[7]), 1), 100 * out[5] / cyc, out[5] / max(int(out[2]), 1), 100 * out[6] / cyc, out[6] / max(int(out[3]), 1)))
    print("%-9s %dx%d@%d: %.2f ms  rays=%d  %.3f Grays/s  mean=%.5f" % (name, W, H, spp, ms, int(cnt[0]), int(cnt[0]) / ms / 1e6, float(img.mean()) / spp), flush=True)
