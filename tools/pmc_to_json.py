#!/usr/bin/env python3
"""Turns one run of tools/profile_round.sh into the record bench.py reads (profiles/pmc_counters.json), stamped with the hash
of the kernel sources it was collected for, so that counters of another build cannot be mistaken for this one's:

    python tools/pmc_to_json.py gpurun_out/<dir> headline profiles/r03_x      (workload name, prefix of the committed copies)

Reads <dir>/pmc_summary.txt (the k_trace sections), <dir>/kernel_stats.csv and <dir>/bench_trace.json."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash, the peak figures)

src, name, prefix = sys.argv[1], sys.argv[2], sys.argv[3]


def mesh_record():
    """The c5 record, from tools/profile_mesh.sh (+ profile_mesh_mem.sh into the same directory): one launch of k_trace_stream."""
    c = {}
    for line in open(os.path.join(src, "pmc_summary.txt")):
        parts = line.split()
        if len(parts) == 2 and not line.startswith("=="):
            try:
                c[parts[0]] = float(parts[1])
            except ValueError:
                pass
    run = open(os.path.join(src, "run_trace.txt")).read()
    m = re.search(r"(\d+)x(\d+)@(\d+): ([0-9.]+) ms\s+rays=(\d+)", run)
    W, spp, rays = int(m.group(1)), int(m.group(3)), int(m.group(5))
    kernel = None
    for row in csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))):
        if "k_trace_stream" in row["Name"]:
            kernel = row
            break
    calls, total_s = int(kernel["Calls"]), float(kernel["TotalDurationNs"]) * 1e-9
    hbm = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
    rec = {
        "spp": spp, "width": W,
        "kernel": re.sub(r"\(rrt::DScene.*", "", kernel["Name"]).replace("void (anonymous namespace)::", ""),
        "kernel_source_hash": bench.kernel_source_hash(),
        "launches_profiled": calls,
        "kernel_ms_in_the_trace_run": float(kernel["AverageNs"]) * 1e-6,
        "rays_profiled": rays * calls,
        "SQ_INSTS_VALU": c["SQ_INSTS_VALU"], "SQ_INSTS_SALU": c.get("SQ_INSTS_SALU"), "SQ_INSTS_LDS": c.get("SQ_INSTS_LDS"),
        "SQ_ACTIVE_INST_VALU": c["SQ_ACTIVE_INST_VALU"], "SQ_THREAD_CYCLES_VALU": c["SQ_THREAD_CYCLES_VALU"],
        "valu_wave_instr_per_ray": c["SQ_INSTS_VALU"] / (rays * calls),
        "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
        "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4.0 / (bench.SIMDS * total_s * bench.PEAK_CLOCK_GHZ * 1e9),
        "fetch_size_kb_total": c.get("FETCH_SIZE"), "write_size_kb_total": c.get("WRITE_SIZE"),
        "TCC_HIT_sum": c.get("TCC_HIT_sum"), "TCC_MISS_sum": c.get("TCC_MISS_sum"),
        "TCP_TOTAL_CACHE_ACCESSES_sum": c.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
        # (TA_BUSY_avr is a mean over the address units; GRBM_GUI_ACTIVE comes summed over the eight XCDs)
        "TA_BUSY_avr_frac": (8.0 * c["TA_BUSY_avr"] / c["GRBM_GUI_ACTIVE"]) if c.get("TA_BUSY_avr") and c.get("GRBM_GUI_ACTIVE") else None,
        "hbm_bytes_per_launch": hbm / calls,
        "hbm_bytes_per_ray": hbm / (rays * calls),
        "correction": "FETCH_SIZE x2 (gfx950 counts 128-B read requests as 64 B); WRITE_SIZE taken as reported",
        "source": "%s_pmc_summary.txt (rocprofv3 --pmc, separate passes of `python3 tools/perf_mesh.py 2236 3840 2160 16 tight clean`)" % prefix,
    }
    return rec


if name == "c5":
    rec = mesh_record()
    path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    allrec = json.load(open(path)) if os.path.exists(path) else {}
    allrec[name] = rec
    with open(path, "w") as f:
        json.dump(allrec, f, indent=1)
        f.write("\n")
    print(json.dumps(rec, indent=1))
    sys.exit(0)
counters = {}
section = None
for line in open(os.path.join(src, "pmc_summary.txt")):
    m = re.match(r"== (\S+) (\S+)", line)
    if m:
        section = m.group(2)
        continue
    parts = line.split()
    if section == "k_trace" and len(parts) == 2:
        counters[parts[0]] = float(parts[1])
line = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
kernel = None
for row in csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))):
    if "k_trace" in row["Name"]:
        kernel = row
        break
calls = int(kernel["Calls"])
total_s = float(kernel["TotalDurationNs"]) * 1e-9
rays = line["roofline"]["rays_per_launch"] * calls
active = counters["SQ_ACTIVE_INST_VALU"]
fetch_kb = counters.get("FETCH_SIZE", 0.0)
write_kb = counters.get("WRITE_SIZE", 0.0)
rec = {
    "spp": int(re.search(r"@ (\d+) spp", line["config"]["workload"]).group(1)),
    "width": int(re.search(r"(\d+)x\d+ @", line["config"]["workload"]).group(1)),
    "kernel": re.sub(r"\(rrt::DScene.*", "", kernel["Name"]).replace("void (anonymous namespace)::", ""),
    "kernel_source_hash": bench.kernel_source_hash(),
    "launches_profiled": calls,
    "kernel_ms_in_the_trace_run": float(kernel["AverageNs"]) * 1e-6,
    "rays_profiled": rays,
    "SQ_INSTS_VALU": counters["SQ_INSTS_VALU"],
    "SQ_INSTS_SALU": counters.get("SQ_INSTS_SALU"),
    "SQ_INSTS_LDS": counters.get("SQ_INSTS_LDS"),
    "SQ_ACTIVE_INST_VALU": active,
    "SQ_THREAD_CYCLES_VALU": counters["SQ_THREAD_CYCLES_VALU"],
    "SQ_WAIT_INST_ANY_over_SQ_WAVE_CYCLES": counters.get("SQ_WAIT_INST_ANY", 0.0) / counters["SQ_WAVE_CYCLES"] if counters.get("SQ_WAVE_CYCLES") else None,
    "valu_wave_instr_per_ray": counters["SQ_INSTS_VALU"] / rays,
    "salu_instr_per_ray": counters.get("SQ_INSTS_SALU", 0.0) / rays,
    "lane_utilisation": counters["SQ_THREAD_CYCLES_VALU"] / (64.0 * active),
    "valu_busy_frac": active * 4.0 / (bench.SIMDS * total_s * bench.PEAK_CLOCK_GHZ * 1e9),
    "valu_busy_note": "SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / 1024 SIMDs over the launches' duration at the 2.4 GHz peak clock",
    "valu_cycles_per_instruction": 4.0 * active / counters["SQ_INSTS_VALU"],
    "valu_cycles_note": "4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU: cycles the vector ALU spends per wave instruction",
    "write_size_kb_total": write_kb,
    "fetch_size_kb_total": fetch_kb,
    "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0 / calls,
    "correction": "FETCH_SIZE x2 (gfx950 counts 128-B read requests as 64 B); WRITE_SIZE taken as reported",
    "source": "%s_pmc_summary.txt (rocprofv3 --pmc, separate passes of `python3 bench.py --steps 2 --warmup 1 --cpu-spp 0`)" % prefix,
}
path = os.path.join(ROOT, "profiles", "pmc_counters.json")
allrec = json.load(open(path)) if os.path.exists(path) else {}
allrec[name] = rec
with open(path, "w") as f:
    json.dump(allrec, f, indent=1)
    f.write("\n")
print(json.dumps(rec, indent=1))
