#!/bin/bash
# SQ counters of the headline scene's trace kernel, pooled schedule against one path per lane (run on the GPU box):
#   bash tools/pmc_compare.sh <out-dir-under-gpurun_out> [spp]
set -e
out=gpurun_out/${1:-pmc_compare}
spp=${2:-128}
mkdir -p "$out"
export TMPDIR=/tmp
for pooling in 1 0; do
  export RRT_POOLING=$pooling
  B="python3 tools/perf_quick.py Cornell_Box_With_Sphere.json 1024 1024 $spp 0 1"
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out/sq_$pooling" -- $B > "$out/run_sq_$pooling.txt"
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d "$out/sq2_$pooling" -- $B > "$out/run_sq2_$pooling.txt"
  for d in sq sq2; do
    echo "== pooling=$pooling $d" >> "$out/summary.txt"
    python3 tools/pmc_summary.py "$out/${d}_$pooling" k_trace >> "$out/summary.txt"
  done
  cat "$out/run_sq_$pooling.txt" >> "$out/summary.txt"
done
find "$out" -name '*counter_collection.csv' -delete
cat "$out/summary.txt"
