import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
torch.cuda.set_device(0); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
import rust_raytracer_amd as rrt
p = "tests/golden/assets/Cornell_Box_With_Sphere.json"
for i in range(3):
    t0 = time.perf_counter(); hs = rrt.HostScene.load(p, 1024, 1024, 1024); t1 = time.perf_counter()
    sc = rrt.Scene(hs); t2 = time.perf_counter()
    print("round %d: host load %.2f ms, scene create %.2f ms (library says %.2f ms)" % (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, sc.upload_seconds() * 1e3), flush=True)
    img = torch.zeros((1024, 1024, 3), dtype=torch.float32, device="cuda")
    t3 = time.perf_counter(); sc.render_device(img.data_ptr(), 0, 0, samples=1); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("   first render of 1 spp on this handle: %.2f ms" % ((t4 - t3) * 1e3), flush=True)
    t3 = time.perf_counter(); sc.render_device(img.data_ptr(), 0, 0, samples=1024); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("   render of 1024 spp (allocates the 12.9 GB staging buffer): %.2f ms" % ((t4 - t3) * 1e3), flush=True)
    t3 = time.perf_counter(); sc.render_device(img.data_ptr(), 0, 0, samples=1024); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("   again: %.2f ms" % ((t4 - t3) * 1e3), flush=True)
    sc.close()
