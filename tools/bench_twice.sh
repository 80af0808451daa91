#!/bin/bash
# Two headline bench runs back to back on the GPU box (run-to-run spread): prints value, ms per step, trace kernel ms, accumulate ms
for i in 1 2; do
  timeout -k 10 200 python bench.py --cpu-spp 0 --steps 5 > gpurun_out/b$i.log 2>&1 || exit 1
  python - "gpurun_out/b$i.log" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"], 2), round(d["roofline"]["kernel_ms"], 2), round(d["roofline"]["accumulate_ms_per_step"], 2))
PY
done
