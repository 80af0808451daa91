set -e
out=gpurun_out/r2e/mesh2
mkdir -p $out
export TMPDIR=/tmp
B="python3 tools/perf_mesh.py 2236 3840 2160 16 tight clean"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --output-format csv -d "$out/sq2" -- $B > "$out/run_sq2.txt"
timeout -s KILL 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_BUSY_CYCLES --output-format csv -d "$out/sq3" -- $B > "$out/run_sq3.txt"
for d in sq2 sq3; do echo "== $d" >> $out/summary.txt; python3 tools/pmc_summary.py $out/$d k_trace_stream >> $out/summary.txt; done
find "$out" -name '*counter_collection.csv' -delete
cat $out/summary.txt
