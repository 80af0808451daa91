#!/bin/bash
# Kernel trace, instruction counters and instruction-cache / LDS counters of the trace kernel on one asset scene (GPU box):
#   bash tools/profile_scene.sh <out-dir-under-gpurun_out> SCENE W H SPP BG
# Each rocprofv3 pass is its own run; counters are never combined with API traces.
set -e
out=gpurun_out/${1:-prof_scene}
shift
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 tools/perf_scene.py $* 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $B > "$out/run_trace.txt"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out/pmc_sq" -- $B > "$out/run_sq.txt"
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d "$out/pmc_sq3" -- $B > "$out/run_sq3.txt"
for d in pmc_sq pmc_sq3; do
  echo "== $d k_trace" >> "$out/pmc_summary.txt"
  python3 tools/pmc_summary.py "$out/$d" k_trace >> "$out/pmc_summary.txt"
done
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
find "$out" -name '*counter_collection.csv' -delete
find "$out" -name '*kernel_trace.csv' -delete
cat "$out/pmc_summary.txt"
head -5 "$out/kernel_stats.csv"
cat "$out/run_trace.txt"
