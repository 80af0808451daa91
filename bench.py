#!/usr/bin/env python3
"""Headline benchmark: Mrays/s on the Cornell box with the glass sphere, 1024x1024 @ 1024 spp per GPU.

    python bench.py --gpus N --steps K --warmup W [--config headline|c4|c5] [--shard samples|tiles]

A step is one full render of the frame (1024*1024*1024 camera samples, each traced to termination) with the scene already
resident in HBM, plus the one collective and the 8-bit resolve.  With N > 1 (launched by torch.distributed.run, one rank per GPU)
every rank renders its own 1024-sample range of the per-pixel random streams (first_sample = rank * 1024) and the radiance sums
are combined with ONE RCCL reduce per step, so N GPUs deliver an N*1024-spp frame: weak scaling, and that is `value`.  For N > 1
the line also carries `strong`: the SAME frame divided over the N GPUs (sample ranges, and interleaved 32x32 tiles), timed for
the same number of steps after the weak ones, each with `efficiency_vs_1gpu` = (rank 0 rendering the whole frame alone) /
(N x the divided frame's time), both measured in this run.
--config c4: assets/Globe.json at 3840x2160 @ 2048 spp (sky background); --config c5: a 10 M-triangle mesh at 3840x2160 @ 16 spp.

Rank 0 prints one JSON line.  `roofline` describes the dominant kernel (the path tracer): `achieved` = the workload model's bytes
per ray (192 B per path segment for a scene that stays in cache: what a tracer with ray queues in HBM moves) x the rays of a launch
/ the launch's duration, measured live with HIP events that the library records around every launch on the render stream;
`peak` 8 TB/s.  The headline kernel keeps all path state on chip, so it moves a fraction of those bytes (`traffic`, from PMC
passes) and what really bounds it is vector-ALU issue: `roofline.valu` holds that accounting (instructions per ray, lane
utilisation, busy fraction, from the committed PMC passes of this same command, profiles/pmc_counters.json).  Those counters
belong to one build of the kernels: the file records a hash of the kernel sources, and when it does not match the sources this
run loaded, `traffic` is null, `valu` is left out and `counters_stale` says why.
`cpu_baseline` times the CPU oracle (a port of the reference's algorithm) on a bounded sample of the same frame on this host's
cores - all of them (at most 16) and the reference's default of two threads (main.rs:66-72); it is never part of the product path.
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
# Vector-ALU issue: 256 CUs x 4 SIMDs, one wave64 instruction per 2 cycles per SIMD with two or more waves resident, at the
# 2.4 GHz peak engine clock.
SIMDS = 1024
PEAK_CLOCK_GHZ = 2.4
VALU_PEAK_GINSTR = SIMDS * PEAK_CLOCK_GHZ / 2.0
# The 10 M-triangle mesh (--config c5): mean node fetches and pair records tested per ray on the clean four-wide tree, counted by
# the profiling build (tools/perf_mesh.py); bytes per ray by the formula 192 + 32 * nodes + 48 * triangles.
C5_GRID = 2236
C5_NODES_PER_RAY = 87.0
C5_TRIANGLES_PER_RAY = 24.0


def cpu_baseline(spp, threads):
    """Times the CPU oracle on the full 1024x1024 frame at `spp` samples per pixel (same scene, same RNG, same ray definition)."""
    import orc

    sc = orc.OracleScene.from_file(SCENE, {"resolution_x": WIDTH, "resolution_y": HEIGHT, "samples": spp})
    t0 = time.perf_counter()
    _, st = sc.render(WIDTH, HEIGHT, spp, seed=0, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": float(st[0]) / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "full %dx%d frame at %d spp (%d rays), CPU oracle, loop-form trace, reference AABB test" % (WIDTH, HEIGHT, spp, int(st[0])),
            "seconds": dt}


def kernel_source_hash():
    """Hash of every source the kernels are built from: the PMC counters in profiles/pmc_counters.json describe one build."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "rust_raytracer_amd", "csrc")
    names = sorted(n for n in os.listdir(csrc) if n.endswith((".hip", ".hpp")))
    for n in names + [os.path.join("..", "build.py"), os.path.join("..", "..", "include", "rrt.h")]:
        with open(os.path.join(csrc, n), "rb") as f:
            h.update(n.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def pmc_record(name):
    """(record, stale reason or None): the counters of the named workload, and whether they describe the kernels of this tree."""
    path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    if not os.path.exists(path):
        return None, "profiles/pmc_counters.json is missing"
    with open(path) as f:
        rec = json.load(f).get(name)
    if rec is None:
        return None, "no counters recorded for this workload"
    now = kernel_source_hash()
    if rec.get("kernel_source_hash") != now:
        return rec, "counters were collected for kernel sources %s, this tree's are %s: collect them again (tools/profile_round.sh, tools/pmc_to_json.py)" % (
            rec.get("kernel_source_hash"), now)
    return rec, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["headline", "c4", "c5"], default="headline",
                    help="headline: Cornell box with the glass sphere 1024x1024 @ 1024 spp; c4: Globe.json 3840x2160 @ 2048 spp, sky background; "
                         "c5: the 10 M-triangle mesh at 3840x2160 @ 16 spp per step")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per GPU per step (default: 1024 for the headline config, 2048 for c4, 16 for c5)")
    ap.add_argument("--shard", choices=["samples", "tiles"], default="samples")
    ap.add_argument("--cpu-spp", type=int, default=96, help="samples per pixel of the CPU-oracle sample (0 = skip); 96 spp is about 20 s on 16 cores")
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
    n_devices = torch.cuda.device_count()
    rehearsal = world > n_devices  # more ranks than GPUs: the ranks share GPUs and talk over gloo - exercises the N > 1 code path, measures nothing
    device_index = local_rank % n_devices
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def note(text):
        if rehearsal:
            print("rehearsal: rank %d %s" % (rank, text), file=sys.stderr, flush=True)

    note("joined the group")
    c5 = args.config == "c5"
    background = rrt.BG_BLACK
    data = "the reference's own scene file (assets/Cornell_Box_With_Sphere.json) at the benchmark's resolution; no dataset involved"
    if args.config == "c4":
        W, H = 3840, 2160
        spp = args.spp or 2048
        background = rrt.BG_SKY  # the scene has no emitter: black would give a black frame (main.rs:259-270)
        globe = os.path.join(ROOT, "tests", "golden", "assets", "Globe.json")
        host = rrt.HostScene.load(globe, width=W, height=H, samples=spp, aspect_ratio=16.0 / 9.0)
        build_s = sum(host.times())
        scene = rrt.Scene(host)
        workload = "assets/Globe.json %dx%d @ %d spp per GPU per step, aspect ratio 16:9, sky background" % (W, H, spp)
        bytes_per_ray = float(BYTES_PER_RAY)
        data = "the reference's own scene file (assets/Globe.json with its two earth textures) at the benchmark's resolution"
    elif c5:
        data = "synthetic (seeded displaced-grid mesh, generated at run time)"
        import synth

        W, H = 3840, 2160
        spp = args.spp or 16
        desc = synth.terrain_desc(C5_GRID, seed=1, width=W, height=H, samples=spp)
        t0 = time.perf_counter()
        host = synth.host_scene(desc, "BVH_CLEAN")
        build_s = time.perf_counter() - t0
        scene = rrt.Scene(host)
        workload = ("synthetic displaced grid, %d triangles, %dx%d @ %d spp per GPU per step; CLEAN tree (built on the GPU; queries return "
                    "what the reference's List aggregate returns - not the reference's own BVH, which this mesh makes 250x slower)"
                    % (2 * C5_GRID * C5_GRID + 2, W, H, spp))
        bytes_per_ray = BYTES_PER_RAY + 32.0 * C5_NODES_PER_RAY + 48.0 * C5_TRIANGLES_PER_RAY
    else:
        W, H = WIDTH, HEIGHT
        spp = args.spp or SPP
        host = rrt.HostScene.load(SCENE, width=W, height=H, samples=spp)
        build_s = sum(host.times())
        scene = rrt.Scene(host)  # upload happens here, outside the timed region
        workload = "assets/Cornell_Box_With_Sphere.json %dx%d @ %d spp per GPU per step" % (W, H, spp)
        bytes_per_ray = float(BYTES_PER_RAY)
    upload_ms = scene.upload_seconds() * 1e3  # the process's first handle: includes loading the library's code object onto the device
    upload_warm_ms = upload_ms
    if not c5:  # (the mesh's upload is 2 GB of copies either way)
        second = rrt.Scene(host)  # what a scene costs once the process is warm
        upload_warm_ms = second.upload_seconds() * 1e3
        second.close()
    note("scene uploaded")
    img = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    host_rgb8 = torch.empty((H, W, 3), dtype=torch.uint8).pin_memory()
    counters = torch.zeros(2, dtype=torch.int64, device=dev)
    aabb = rrt.AABB_TIGHT if args.aabb == "tight" else rrt.AABB_REFERENCE
    render_stream = torch.cuda.Stream(device=dev)
    rendered = torch.cuda.Event()

    def plan(kind, total_spp):
        """Render arguments of this rank: weak = `total_spp` samples of its own range; strong = its share of ONE `total_spp` frame."""
        if kind == "weak-samples":
            first, count = shard.sample_range(rank, total_spp)
            return dict(samples=count, first_sample=first, seed=0, aabb_mode=aabb, background=background), total_spp * world
        if kind == "strong-samples":
            first, count = shard.split_samples(rank, world, total_spp)
            return dict(samples=count, first_sample=first, seed=0, aabb_mode=aabb, background=background), total_spp
        if kind == "solo":  # rank 0 renders the whole frame, the others add nothing: the one-GPU time the strong figures are held against
            return dict(samples=total_spp if rank == 0 else 0, first_sample=0, seed=0, aabb_mode=aabb, background=background), total_spp
        return dict(samples=total_spp, first_sample=0, seed=0, aabb_mode=aabb, background=background, tile_count=world, tile_index=rank), total_spp

    def step(kw, frame_spp, tiles, ev0=None, ev1=None):
        with torch.cuda.stream(render_stream):
            if ev0 is not None:
                ev0.record(render_stream)
            if kw["samples"] > 0:
                scene.render_device(img.data_ptr(), counters.data_ptr(), render_stream.cuda_stream, out_floats=img.numel(), **kw)
            else:
                img.zero_()  # more ranks than samples: this rank adds nothing
            if ev1 is not None:
                ev1.record(render_stream)
            rendered.record(render_stream)
        cur = torch.cuda.current_stream()
        cur.wait_event(rendered)  # the collective below is enqueued behind the render explicitly, not by stream coincidence
        # the one collective of the job (no-op on one GPU): sample ranges are summed, tiles - each rank's own pixels, packed: an
        # N-th of the frame per rank - are gathered and put in place
        if tiles:
            shard.gather_tiles(img, dst=0)
        else:
            shard.reduce_frame(img, dst=0)
        if rank == 0:  # time-to-image ends with the 8-bit frame (main.rs:173-186), still in HBM
            rrt.resolve_rgb8_device(img.data_ptr(), W, H, frame_spp, rgb8.data_ptr(), cur.cuda_stream)
        render_stream.wait_stream(cur)  # the next step's render must not overwrite the frame under the reduce / resolve

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(kind, total_spp, steps, warmup, record_events):
        kw, frame_spp = plan(kind, total_spp)
        tiles = kind == "tiles"
        for _ in range(warmup):
            step(kw, frame_spp, tiles)
            note("warm-up step of %s enqueued" % kind)
        fence()
        note("warm-up of %s done" % kind)
        counters.zero_()
        scene.kernel_times()  # forget the warm-up passes
        torch.cuda.synchronize()
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] if record_events else [(None, None)] * steps
        fence()
        t0 = time.perf_counter()
        for e0, e1 in events:
            step(kw, frame_spp, tiles, e0, e1)
        fence()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        c = counters.cpu() if rehearsal else counters.clone()
        if rehearsal:
            print("rehearsal: rank %d timed %d steps of %s in %.3f s" % (rank, steps, kind, elapsed), file=sys.stderr, flush=True)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
        render_ms = [e0.elapsed_time(e1) for e0, e1 in events] if record_events else []
        return float(t.item()), int(c[0].item()), int(c[1].item()), render_ms, int(counters[0].item())

    scene.set_timing(True)
    main_kind = "weak-samples" if args.shard == "samples" else "tiles"
    elapsed, rays_total, paths_total, render_ms, rays_rank0 = timed(main_kind, spp, args.steps, args.warmup, True)
    trace_ms, accumulate_ms, passes = scene.kernel_times()

    strong = None
    if world > 1:
        strong = {}
        solo_s, _, _, _, _ = timed("solo", spp, args.steps, 1, False)
        solo_ms = solo_s / max(args.steps, 1) * 1e3
        for kind in ("strong-samples", "tiles"):
            e, r, p, _, _ = timed(kind, spp, args.steps, 1, False)
            ms = e / max(args.steps, 1) * 1e3
            strong[kind.replace("strong-", "")] = {"ms_per_frame": ms, "value": r / e / 1e6, "unit": "Mrays/s",
                                                   "one_gpu_ms_per_frame": solo_ms, "efficiency_vs_1gpu": solo_ms / (world * ms),
                                                   "frame": "%dx%d @ %d spp in total, divided over %d GPUs" % (W, H, spp, world)}
        scene.kernel_times()

    # the 8-bit frame's way to the host: part of time-to-image, not of `value`.  The process's first copy into a pinned buffer also
    # pays for mapping that buffer to the device (milliseconds); later copies of the same 3 MB take a tenth of a millisecond.
    d2h = []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host_rgb8.copy_(rgb8, non_blocking=True)
        torch.cuda.synchronize()
        d2h.append((time.perf_counter() - t0) * 1e3)
    d2h_ms, d2h_warm_ms = d2h

    if rank == 0:
        launches = max(passes, 1)  # one path-tracer launch per pass of the staging buffer
        rays_per_launch = rays_rank0 / launches
        k_s = trace_ms / launches * 1e-3
        nominal = rays_per_launch * bytes_per_ray / k_s / 1e9
        step_ms = elapsed / max(args.steps, 1) * 1e3
        pmc, stale = pmc_record(args.config)
        if pmc is not None and not (pmc.get("spp") == spp and pmc.get("width") == W):
            stale = "counters were collected at %s x ... @ %s spp, this run is %d @ %d" % (pmc.get("width"), pmc.get("spp"), W, spp)
        fresh = pmc is not None and stale is None
        # The contract's figure: the workload model's bytes per ray x rays / kernel time against the HBM peak.
        roofline = {"kernel": "k_trace_stream" if c5 else "k_trace_pool", "bound": "hbm", "achieved": nominal, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": nominal / HBM_PEAK_GBS, "traffic": pmc.get("hbm_bytes_per_launch") if fresh else None,
                    "kernel_ms": k_s * 1e3, "launches_per_step": launches / max(args.steps, 1), "rays_per_launch": rays_per_launch,
                    "bytes_per_ray": bytes_per_ray, "accumulate_ms_per_step": accumulate_ms / max(args.steps, 1),
                    "render_ms_per_step": float(np.mean(render_ms)) if render_ms else None,
                    "kernel_source_hash": kernel_source_hash()}
        if fresh:
            roofline["counters_from"] = pmc.get("source")
            roofline["hbm_traffic_frac"] = pmc.get("hbm_bytes_per_launch", 0.0) / k_s / 1e9 / HBM_PEAK_GBS
        else:
            roofline["counters_stale"] = stale
        if not c5:
            roofline["achieved_is"] = ("algorithmic bytes: 192 B per ray is what a tracer with its ray queues in HBM would move; this kernel keeps path state "
                                       "on chip and moves `traffic` instead, so its real limiter is vector-ALU issue: see `valu`")
            if fresh and pmc.get("valu_wave_instr_per_ray"):
                per_ray = pmc["valu_wave_instr_per_ray"]
                issued = rays_per_launch * per_ray / k_s / 1e9
                roofline["valu"] = {"unit": "Gwave-instr/s", "peak": VALU_PEAK_GINSTR, "achieved": issued, "issue_frac": issued / VALU_PEAK_GINSTR,
                                    "wave_instr_per_ray": per_ray, "lane_utilisation": pmc.get("lane_utilisation"),
                                    "salu_instr_per_ray": pmc.get("salu_instr_per_ray"), "busy_frac": pmc.get("valu_busy_frac"),
                                    "cycles_per_instruction": pmc.get("valu_cycles_per_instruction"),
                                    "note": "`peak` is one wave instruction per 2 cycles per SIMD (plain f32 multiply-add); this kernel's mix (packed f32, "
                                            "scalar operands, compares, selects, min/max, reciprocal, square root) costs cycles_per_instruction on average, "
                                            "and with that mix the ALU is busy busy_frac of the launch"}
        out = {
            "metric": "Mrays/sec (whole node) + time-to-image, Cornell box 1024^2 @1024 spp",
            "value": rays_total / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_ms,
            "higher_is_better": True,
            "scaling": "weak" if args.shard == "samples" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": data,
            "config": {"workload": workload,
                       "shard": "sample ranges, one RCCL reduce per step" if args.shard == "samples" else "interleaved 32x32 tiles, one RCCL gather of the ranks' own pixels per step",
                       "value_is": "weak scaling: every GPU traces %d spp of its own, the frame has %d spp" % (spp, spp * world) if args.shard == "samples"
                                   else "strong scaling: one %d-spp frame divided into interleaved tiles" % spp,
                       "background": "sky" if background == rrt.BG_SKY else "black", "seed": 0, "aabb_mode": args.aabb, "max_depth": 50,
                       "samples_per_second": paths_total / elapsed, "rays_per_path": rays_total / max(paths_total, 1),
                       "time_to_image_ms": upload_ms + step_ms + d2h_ms,
                       "time_to_image_parts_ms": {"scene_upload": upload_ms, "render_reduce_resolve": step_ms, "rgb8_to_host": d2h_ms,
                                                  "host_load_and_build_not_included": build_s * 1e3,
                                                  "scene_upload_in_a_warm_process": upload_warm_ms, "rgb8_to_host_in_a_warm_process": d2h_warm_ms,
                                                  "scene_upload_parts": scene.upload_parts()},
                       "time_to_image_warm_ms": upload_warm_ms + step_ms + d2h_warm_ms},
            "roofline": roofline,
        }
        if strong is not None:
            out["strong"] = strong
        if rehearsal:
            out["config"]["rehearsal"] = ("%d ranks on %d GPU(s) over gloo: the ranks share a GPU, so `value` and the strong-scaling figures only show "
                                          "that the N > 1 path runs (render stream -> event -> collective -> resolve), not how it scales" % (world, n_devices))
        if world == 1 and args.cpu_spp > 0 and args.config == "headline":
            threads = min(len(os.sched_getaffinity(0)), 16)
            out["cpu_baseline"] = cpu_baseline(args.cpu_spp, threads)
            two = cpu_baseline(max(args.cpu_spp // 8, 1), 2)  # the reference's default thread count (main.rs:66-72)
            out["cpu_baseline"]["reference_default_threads"] = {k: two[k] for k in ("value", "unit", "cores", "sample", "seconds")}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
