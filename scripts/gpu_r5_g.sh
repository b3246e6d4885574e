#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_g.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_g.log; tail -n 3 gpurun_out/pytest_g.log
(timeout 900 python scripts/fuzz_parity.py 200) > gpurun_out/fuzz_g.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/fuzz_g.log | cut -c1-600; cp gpurun_out/fuzz_parity.json gpurun_out/fuzz_parity_200_scenes_default_g.json
(timeout 1200 python bench.py --steps 10 --warmup 3) > gpurun_out/bench_g.json 2>gpurun_out/bench_g.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_g.json").read().strip().splitlines()[-1])
print("ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "parity", {k: v for k, v in d["parity"].items() if k not in ("what", "sample")})
print("configs4", d["secondary"]["configs4_full"]["ms_per_step"], d["secondary"]["configs4_full"]["parity"])
PY
