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
