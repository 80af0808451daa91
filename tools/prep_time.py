import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, synth
torch.zeros(1, device="cuda")
d = synth.terrain_desc(2236, seed=1, width=64, height=64, samples=1)
hs = synth.host_scene(d, "BVH_CLEAN")
t0 = time.time(); c = hs.check(); print("check %.2f s" % (time.time() - t0), c)
