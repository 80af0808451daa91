#!/bin/bash
# Kernel trace and HBM-traffic counters for the streaming kernel on the synthetic 10 M-triangle mesh with a clean tree (GPU box):
#   [MESH_ARGS="N W H SPP tight clean"] bash tools/profile_mesh.sh <out-dir-under-gpurun_out>
# Each rocprofv3 pass is its own run; counters are never combined with API traces.
set -e
out=gpurun_out/${1:-prof_mesh}
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 tools/perf_mesh.py ${MESH_ARGS:-2236 3840 2160 16 tight clean}"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $B > "$out/run_trace.txt"
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- $B > "$out/run_fetch.txt"
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_write" -- $B > "$out/run_write.txt"
timeout -s KILL 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out/pmc_sq" -- $B > "$out/run_sq.txt"
for d in pmc_fetch pmc_write pmc_sq; do
  echo "== $d k_trace_stream" >> "$out/pmc_summary.txt"
  python3 tools/pmc_summary.py "$out/$d" k_trace_stream >> "$out/pmc_summary.txt"
done
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
find "$out" -name '*counter_collection.csv' -delete
find "$out" -name '*kernel_trace.csv' -delete
cat "$out/pmc_summary.txt"
head -8 "$out/kernel_stats.csv"
cat "$out/run_trace.txt"
