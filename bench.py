#!/usr/bin/env python3
"""Headline benchmark: Mrays/s on the Cornell box with the glass sphere, 1024x1024 @ 1024 spp per GPU.

    python bench.py --gpus N --steps K --warmup W

A step is one full render of the frame (1024*1024*1024 camera samples, each traced to termination) with
the scene already resident in HBM.  With N > 1 (launched by torch.distributed.run, one rank per GPU) every
rank renders its own 1024-sample range of the per-pixel random streams (first_sample = rank * 1024) and the
radiance sums are combined with ONE RCCL reduce per step, so N GPUs deliver an N*1024-spp frame: weak scaling.
`--shard tiles` instead splits one 1024-spp frame into interleaved 32x32 tiles (strong scaling).

Rank 0 prints one JSON line.  `roofline` is for the dominant kernel, k_trace: its launch durations are measured
live with HIP events that the library records around every launch on the render stream, and `achieved` prices the
rays of one launch with the per-ray byte figure of a queue-based wavefront tracer (192 B per path segment).
k_trace keeps path state in registers, so its measured HBM traffic (`traffic`, from the committed PMC passes) is the
12 bytes per path it writes and the kernel is VALU-bound, not HBM-bound.
`cpu_baseline` times the CPU oracle (a port of the reference's algorithm) on a bounded sample of the same
frame on this host's cores; it is never part of the product path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENE = os.path.join(ROOT, "tests", "golden", "assets", "Cornell_Box_With_Sphere.json")
WIDTH = HEIGHT = 1024
SPP = 1024
BYTES_PER_RAY = 192  # wavefront queue traffic per path segment (extend: ray in 32 + hit out 16; shade: 80 in, 64 out)
HBM_PEAK_GBS = 8000.0


def cpu_baseline(spp):
    """Times the CPU oracle on the full 1024x1024 frame at `spp` samples per pixel (same scene, same RNG, same ray definition)."""
    import orc

    threads = min(len(os.sched_getaffinity(0)), 16)
    sc = orc.OracleScene.from_file(SCENE, {"resolution_x": WIDTH, "resolution_y": HEIGHT, "samples": spp})
    t0 = time.perf_counter()
    _, st = sc.render(WIDTH, HEIGHT, spp, seed=0, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": float(st[0]) / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "full %dx%d frame at %d spp (%d rays), CPU oracle, loop-form trace, reference AABB test" % (WIDTH, HEIGHT, spp, int(st[0])),
            "seconds": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP, help="samples per pixel per GPU per step (the headline config is 1024)")
    ap.add_argument("--shard", choices=["samples", "tiles"], default="samples")
    ap.add_argument("--cpu-spp", type=int, default=96, help="samples per pixel of the CPU-oracle sample (0 = skip); 96 spp is about 15 s on 16 cores")
    ap.add_argument("--aabb", choices=["tight", "reference"], default="tight")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import rust_raytracer_amd as rrt
    from rust_raytracer_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    scene = rrt.Scene.load(SCENE, width=WIDTH, height=HEIGHT, samples=args.spp)  # upload happens here, outside the timed region
    img = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.float32, device=dev)
    rgb8 = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.uint8, device=dev)
    counters = torch.zeros(2, dtype=torch.int64, device=dev)
    aabb = rrt.AABB_TIGHT if args.aabb == "tight" else rrt.AABB_REFERENCE
    if args.shard == "samples":
        first, count = shard.sample_range(rank, args.spp)
        kw = dict(samples=count, first_sample=first, seed=0, aabb_mode=aabb)
    else:
        kw = dict(samples=args.spp, first_sample=0, seed=0, aabb_mode=aabb, tile_count=world, tile_index=rank)

    scene.set_timing(True)

    def step(ev0=None, ev1=None):
        stream = torch.cuda.current_stream()
        if args.shard == "tiles":
            img.zero_()
        if ev0 is not None:
            ev0.record(stream)
        scene.render_device(img.data_ptr(), counters.data_ptr(), stream.cuda_stream, out_floats=img.numel(), **kw)
        if ev1 is not None:
            ev1.record(stream)
        shard.reduce_frame(img, dst=0)  # the one collective of the job (no-op on one GPU)
        if rank == 0:  # time-to-image ends with the 8-bit frame (main.rs:173-186), still in HBM
            total_spp = args.spp * world if args.shard == "samples" else args.spp
            rrt.resolve_rgb8_device(img.data_ptr(), WIDTH, HEIGHT, total_spp, rgb8.data_ptr(), torch.cuda.current_stream().cuda_stream)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    counters.zero_()
    scene.kernel_times()  # forget the warm-up passes
    torch.cuda.synchronize()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for e0, e1 in events:
        step(e0, e1)
    fence()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    c = counters.clone()
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    rays_total, paths_total = int(c[0].item()), int(c[1].item())
    render_ms = [e0.elapsed_time(e1) for e0, e1 in events]
    trace_ms, accumulate_ms, passes = scene.kernel_times()

    if rank == 0:
        launches = max(passes, 1)  # one k_trace launch per pass of the staging buffer
        rays_per_launch = int(counters[0].item()) / launches
        k_s = trace_ms / launches * 1e-3
        achieved = rays_per_launch * BYTES_PER_RAY / k_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                rec = json.load(f)
            if rec.get("spp") == args.spp and rec.get("width") == WIDTH:
                traffic = rec.get("hbm_bytes_per_launch")  # PMC passes of the same command, corrected for gfx950
        out = {
            "metric": "Mrays/sec (whole node) + time-to-image, Cornell box 1024^2 @1024 spp",
            "value": rays_total / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if args.shard == "samples" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "assets/Cornell_Box_With_Sphere.json %dx%d @ %d spp per GPU per step" % (WIDTH, HEIGHT, args.spp),
                       "shard": "sample ranges, one RCCL reduce per step" if args.shard == "samples" else "interleaved 32x32 tiles, one RCCL reduce per step",
                       "background": "black", "seed": 0, "aabb_mode": args.aabb, "max_depth": 50,
                       "samples_per_second": paths_total / elapsed, "rays_per_path": rays_total / max(paths_total, 1),
                       "time_to_image_ms": elapsed / max(args.steps, 1) * 1e3},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_trace", "kernel_ms": k_s * 1e3, "launches_per_step": launches / max(args.steps, 1),
                         "bytes_per_ray": BYTES_PER_RAY, "rays_per_launch": rays_per_launch,
                         "accumulate_ms_per_step": accumulate_ms / max(args.steps, 1), "render_ms_per_step": float(np.mean(render_ms)),
                         "limiter": "vector ALU issue: SQ_ACTIVE_INST_VALU covers the whole launch (profiles/r01_d_summary.md); "
                                    "the 192 B/ray figure is what a queue-based tracer would move, this kernel moves 3.4 B/ray"},
        }
        if world == 1 and args.cpu_spp > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_spp)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
