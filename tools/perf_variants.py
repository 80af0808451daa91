#!/usr/bin/env python3
"""Times alternative builds of the library (librrt_hip_*.so next to the main one) on the headline scene."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for so in sorted(glob.glob(os.path.join(ROOT, "rust_raytracer_amd", "librrt_hip*.so"))):
    env = dict(os.environ, RRT_LIB=so)
    print(os.path.basename(so), flush=True)
    subprocess.call([sys.executable, os.path.join(ROOT, "tools", "perf_quick.py")] + sys.argv[1:], env=env)
