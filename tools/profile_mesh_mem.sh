#!/bin/bash
# Texture-address / vector-cache counters of the streaming kernel on the synthetic 10 M-triangle mesh with a clean tree (GPU box):
#   [MESH_ARGS="N W H SPP tight clean"] bash tools/profile_mesh_mem.sh <out-dir-under-gpurun_out>
# Each rocprofv3 pass is its own run; counters are never combined with API traces.
set -e
out=gpurun_out/${1:-prof_mesh_mem}
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 tools/perf_mesh.py ${MESH_ARGS:-2236 3840 2160 16 tight clean}"
timeout -s KILL 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_ta" -- $B > "$out/run_ta.txt"
timeout -s KILL 300 rocprofv3 --pmc TA_BUSY_avr TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_tcp" -- $B > "$out/run_tcp.txt"
for d in pmc_ta pmc_tcp; do
  echo "== $d k_trace_stream" >> "$out/pmc_summary.txt"
  python3 tools/pmc_summary.py "$out/$d" k_trace_stream >> "$out/pmc_summary.txt"
done
find "$out" -name '*counter_collection.csv' -delete
cat "$out/pmc_summary.txt"
cat "$out/run_ta.txt"
