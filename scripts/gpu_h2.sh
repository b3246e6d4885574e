#!/bin/bash
# mode-2 (k_geo_rows_h2) check on the GPU box: soak + bench A/B.  Usage: gpurun -- bash scripts/gpu_h2.sh [tag]
tag=${1:-h2}
mkdir -p gpurun_out
python scripts/soak_mode2.py --repeats 20 > gpurun_out/${tag}_soak.json 2> gpurun_out/${tag}_soak.err
cat gpurun_out/${tag}_soak.json
for m in 0 2; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --geo-rows-mode $m > gpurun_out/${tag}_bench_mode$m.json 2> gpurun_out/${tag}_bench_mode$m.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/${tag}_bench_mode$m.json") if l.startswith("{")][-1])
print("mode $m:", d["ms_per_step"], "ms/frame", d["roofline"])
PY
done
