#!/usr/bin/env python3
"""Renders the five workload configurations on one GPU and prints one JSON record per configuration
(time per frame, rays, Grays/s; frames too long for one run are rendered at fewer samples and say so).
   python tools/run_configs.py [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import rust_raytracer_amd as rrt  # noqa: E402
import synth  # noqa: E402

torch.cuda.set_device(0)
ASSETS = os.path.join(ROOT, "tests", "golden", "assets")


def timed(scene, W, H, spp, bg, reps=2):
    img = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream()
    best = None
    for _ in range(reps):
        cnt.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        scene.render_device(img.data_ptr(), cnt.data_ptr(), st.cuda_stream, width=W, height=H, samples=spp, background=bg, out_floats=img.numel())
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best, int(cnt[0].item()), int(cnt[1].item())


out = []


def record(name, W, H, spp_full, spp_run, bg, scene, note=""):
    ms, rays, paths = timed(scene, W, H, spp_run, bg)
    rec = {"config": name, "width": W, "height": H, "spp_config": spp_full, "spp_rendered": spp_run, "background": "sky" if bg else "black",
           "ms": ms, "rays": rays, "grays_per_s": rays / ms / 1e6, "rays_per_path": rays / max(paths, 1),
           "ms_full_config_extrapolated": ms * spp_full / spp_run, "note": note}
    out.append(rec)
    print(json.dumps(rec), flush=True)


def asset(name, W, H, spp, aspect=float("nan"), tree="BVH"):
    return rrt.Scene.load(os.path.join(ASSETS, name), width=W, height=H, samples=spp, aspect_ratio=aspect, tree=tree)


record("C1 Cornell_Box 512x512@64", 512, 512, 64, 64, 0, asset("Cornell_Box.json", 512, 512, 64))
record("C2 Cornell_Box_With_Sphere 1024x1024@1024", 1024, 1024, 1024, 1024, 0, asset("Cornell_Box_With_Sphere.json", 1024, 1024, 1024))
record("C3 Materials 1920x1080@1024", 1920, 1080, 1024, 1024, 1, asset("Materials.json", 1920, 1080, 1024, 16.0 / 9.0))
record("C3b Example_Spec (noise, turbulence, image, bump, medium) 1920x1080@256", 1920, 1080, 256, 256, 1, asset("Example_Spec.json", 1920, 1080, 256, 16.0 / 9.0))
g = asset("Globe.json", 3840, 2160, 2048, 16.0 / 9.0)
g.set_staging_limit(48 << 30)
record("C4 Globe 3840x2160@2048 (one GPU of the eight)", 3840, 2160, 2048, 512, 1, g, "rendered at 512 spp per pass budget; the 8-GPU job gives each rank 256 spp")
del g
t0 = time.perf_counter()
d = synth.terrain_desc(2236, seed=1, width=3840, height=2160, samples=4096)
hs = synth.host_scene(d, "BVH_CLEAN")
t1 = time.perf_counter()
sc = rrt.Scene(hs)
t2 = time.perf_counter()
record("C5 synthetic 10M-triangle mesh 3840x2160@4096, clean tree", 3840, 2160, 4096, 16, 0, sc,
       "9 999 392 triangles; scene + tree build on the device %.2f s, upload %.2f s; 16 of 4096 spp rendered" % (t1 - t0, t2 - t1))
del sc, hs
hs = synth.host_scene(d, "BVH_GPU")
sc = rrt.Scene(hs)
record("C5 synthetic 10M-triangle mesh, the reference's own tree (built on the device)", 1920, 1080, 4096, 1, 0, sc,
       "1 of 4096 spp at a quarter of the frame: the reference's tree costs 1 784 node visits per ray")
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1)
