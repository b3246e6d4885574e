#!/bin/bash
# Round 6: the whole GPU suite, then the driver-style bench line (usage: gpu_r6_suite_and_bench.sh TAG [bench flags])
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r06}; shift
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15) | tee gpurun_out/pytest_gpu_$TAG.log
(timeout 1500 python bench.py --steps 20 --warmup 5 "$@") > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_$TAG.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "clock", r.get("effective_clock_ghz"))
print("sigma_zero_fraction", d.get("sigma_zero_fraction"), d.get("density_first"))
print("parity", json.dumps({k: v for k, v in d.get("parity", {}).items() if k != "what"})[:2500])
s = d.get("secondary", {})
for k, v in s.items():
    print(k, json.dumps({kk: vv for kk, vv in v.items() if kk != "roofline"})[:700])
PY
