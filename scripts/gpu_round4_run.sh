#!/bin/bash
# round-4 evidence run on the GPU box: smoke, full GPU suite, the operand-range report, the driver-style bench line (with the in-run
# comparison against the oracle) and the rocprofv3 kernel table of the same command.   usage: gpu_round4_run.sh TAG [quick]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r4a}
(timeout 600 python __graft_entry__.py --smoke) > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke_$TAG.log
(timeout 1800 python -m pytest tests -m gpu -q -x) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu_$TAG.log
(timeout 900 python scripts/range_gate.py) > gpurun_out/range_gate_$TAG.txt 2>&1; echo "range rc=$?"; tail -70 gpurun_out/range_gate_$TAG.txt | cut -c1-200
if [ "$2" != "quick" ]; then
(timeout 1500 python bench.py) > gpurun_out/bench_$TAG.json 2>gpurun_out/bench_$TAG.err; echo "bench rc=$?"; tail -1 gpurun_out/bench_$TAG.json | cut -c1-600
else
(timeout 900 python bench.py --no-configs4 --steps 5) > gpurun_out/bench_$TAG.json 2>gpurun_out/bench_$TAG.err; echo "bench rc=$?"; tail -1 gpurun_out/bench_$TAG.json | cut -c1-600
fi
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_$TAG.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "avg_launch_ms", d["roofline"]["avg_launch_ms"], "surplus", d["roofline"]["surplus_launches"])
print("parity", d.get("parity"))
print("guard", d["config"].get("range_guard_batches_redone"))
s = d.get("secondary", {})
for k, v in s.items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_frame", "ms_per_step", "forward_ms", "backward_ms", "parity")})
PY
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary) > $R/gpurun_out/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $R; DB=$(find gpurun_out/prof -name "${TAG}_bench*.db" | head -1); python scripts/rocprof_summary.py $DB gpurun_out/${TAG}_kernel_stats.md > /dev/null 2>&1; head -14 gpurun_out/${TAG}_kernel_stats.md; rm -rf gpurun_out/prof
