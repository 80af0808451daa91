"""Builds librrt_hip.so (HIP kernels + C ABI + host-side scene code) in-tree for gfx950."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librrt_hip.so")
CLI = os.path.join(HERE, "rust-raytracer")

SOURCES = ["rrt_kernels_shallow.hip", "rrt_scene_prepare.cpp", "rrt_scene.cpp", "rrt_render.cpp", "rrt_multi.cpp", "rrt_stream.hip", "rrt_bvh_build.hip", "rrt_noalloc_area.hip", "rrt_error.cpp", "rrt_abi_layout.cpp", "host/scene_host.cpp", "host/scene_json.cpp", "host/image_io.cpp", "host/image_codecs.cpp"]
HEADERS = ["cli/main.cpp", "rrt_device.hpp", "rrt_layout.hpp", "rrt_launch.hpp", "rrt_scene_handle.hpp", "rrt_scene_prepare.hpp", "rrt_math.hpp", "rrt_scene_view.hpp", "rrt_primitives.hpp", "rrt_profile.hpp", "rrt_traversal.hpp", "rrt_shading.hpp", "rrt_error.hpp", "rrt_bvh_build.hpp", "host/scene_host.hpp", "host/hmath.hpp", "../../include/rrt.h"]

# -ffp-contract=off: the reference never fuses a*b+c; bit-exact hit decisions need unfused f32 on host and device.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# The SLP vectoriser packs scalar f32 arithmetic into two-lane instructions; on gfx950 a packed instruction issues no faster than
# the two scalar ones and the packing costs register moves: the shallow-tree kernels run 1 - 9 % faster without it, the deep-tree
# kernels (rrt_stream.hip) 4 % faster with it.  Results are the same bits either way (same IEEE operations, no contraction).
PER_SOURCE_FLAGS = {"rrt_kernels_shallow.hip": ["-fno-slp-vectorize"]}


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(CLI):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, extra_flags=(), out=None):
    lib = out or LIB
    if not out and not force and not needs_build():
        return lib
    objs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(bdir, (os.path.basename(out) + "." if out else "") + src.replace("/", "_") + ".o")
        objs.append(obj)
        cmd = [hipcc()] + FLAGS + PER_SOURCE_FLAGS.get(src, []) + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-o", lib] + objs + ["-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if not out:
        build_cli(verbose)
    return lib


def build_cli(verbose=False):
    """The command line front end: plain C++ on top of the C ABI, finds librrt_hip.so next to itself."""
    cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(HERE, "..", "include"),
           os.path.join(CSRC, "cli", "main.cpp"), "-o", CLI, "-L", HERE, "-l:librrt_hip.so", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
