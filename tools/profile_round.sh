#!/bin/bash
# Collects the kernel-trace statistics and the five PMC passes for the headline bench (run on the GPU box):
#   bash tools/profile_round.sh <out-dir-under-gpurun_out>
# Each rocprofv3 pass is its own run; counters are never combined with API traces.
set -e
out=gpurun_out/${1:-prof}
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 bench.py --steps 2 --warmup 1 --cpu-spp 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $B > "$out/bench_trace.json"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out/pmc_sq" -- $B > "$out/bench_pmc_sq.json"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d "$out/pmc_sq2" -- $B > "$out/bench_pmc_sq2.json"
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d "$out/pmc_sq3" -- $B > "$out/bench_pmc_sq3.json"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- $B > "$out/bench_pmc_fetch.json"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_write" -- $B > "$out/bench_pmc_write.json"
for d in pmc_sq pmc_sq2 pmc_sq3 pmc_fetch pmc_write; do
  for k in k_trace k_accumulate; do
    echo "== $d $k" >> "$out/pmc_summary.txt"
    python3 tools/pmc_summary.py "$out/$d" $k >> "$out/pmc_summary.txt"
  done
done
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
find "$out" -name '*counter_collection.csv' -delete
find "$out" -name '*kernel_trace.csv' -delete
cat "$out/pmc_summary.txt"
cat "$out/kernel_stats.csv"
cat "$out/bench_trace.json"
