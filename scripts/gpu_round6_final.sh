#!/bin/bash
# Evidence run of round 6 on the GPU box: smoke, the whole GPU suite, the driver-style bench line, the rocprofv3 kernel table of the
# same command, PMC passes (separate, kernel-trace only), the training iteration's kernel table, soak of both field kernels, the
# 200-scene sweep.   usage: gpu_round6_final.sh TAG [quick]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r06_z}; QUICK=$2; export HSA_ENABLE_IPC_MODE_LEGACY=0
(timeout 600 python __graft_entry__.py --smoke) > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke_$TAG.log
(timeout 1800 python -m pytest tests -m gpu -q) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu_$TAG.log | cut -c1-300
(timeout 1500 python bench.py --steps 20 --warmup 5) > gpurun_out/bench_$TAG.json 2>gpurun_out/bench_$TAG.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_$TAG.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "clock", r.get("effective_clock_ghz"))
print("workloads", json.dumps(r.get("workloads"), indent=0)[:1500])
print("parity", {k: v for k, v in d.get("parity", {}).items() if k != "what"})
s = d.get("secondary", {})
for k, v in s.items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_frame", "ms_per_step", "forward_ms", "backward_ms", "parity")})
PY
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary) > $R/gpurun_out/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $R; DB=$(find gpurun_out/prof -name "${TAG}_bench*.db" | head -1); python scripts/rocprof_summary.py $DB gpurun_out/${TAG}_bench_kernel_stats.md > /dev/null 2>&1; head -9 gpurun_out/${TAG}_bench_kernel_stats.md; rm -rf gpurun_out/prof
cd /tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o ${TAG}_train -- python $R/scripts/bench_train.py) > $R/gpurun_out/rocprof_train_$TAG.log 2>&1; echo "rocprof train rc=$?"
cd $R; DB=$(find gpurun_out/prof -name "${TAG}_train*.db" | head -1); python scripts/rocprof_summary.py $DB gpurun_out/${TAG}_train_kernel_stats.md > /dev/null 2>&1; head -12 gpurun_out/${TAG}_train_kernel_stats.md; rm -rf gpurun_out/prof
grep -v amdgpu gpurun_out/rocprof_train_$TAG.log | tail -2
[ -n "$QUICK" ] && exit 0
bash scripts/gpu_pmc.sh $TAG 2>&1 | tail -3
ROWS=$(python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r4_pmc_rows.json").read().strip().splitlines()[-1])
print(int(d["config"]["valid_rows_per_step"] * 2))
PY
)
python scripts/pmc_traffic.py gpurun_out/pmc_${TAG}_summary.txt k_geo_rows_f2p $ROWS $TAG | cut -c1-200
cp profiles/geo_rows_traffic.json gpurun_out/geo_rows_traffic.json
(timeout 600 python scripts/soak_mode2.py --mode 3 --repeats 6000) 2>/dev/null | tail -1 | tee gpurun_out/soak_$TAG.jsonl | cut -c1-400
(timeout 600 python scripts/soak_mode2.py --mode 3 --repeats 3000 --mask ellipsoid) 2>/dev/null | tail -1 | tee -a gpurun_out/soak_$TAG.jsonl | cut -c1-300
(timeout 1200 python scripts/fuzz_parity.py 200) > gpurun_out/fuzz_$TAG.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/fuzz_$TAG.log | cut -c1-500; cp gpurun_out/fuzz_parity.json gpurun_out/fuzz_parity_200_scenes_default_$TAG.json
(timeout 1800 python scripts/fuzz_parity.py 400 7) > gpurun_out/fuzz400_$TAG.log 2>&1; echo "fuzz400 rc=$?"; tail -1 gpurun_out/fuzz400_$TAG.log | cut -c1-500; cp gpurun_out/fuzz_parity.json gpurun_out/fuzz_400scenes_seed7_$TAG.json
(timeout 600 python scripts/rccl_one_rank.py 5) 2>/dev/null | grep '^{' | tail -1 | tee gpurun_out/rccl_one_rank_$TAG.json | cut -c1-300
(timeout 900 python scripts/range_gate.py) > gpurun_out/range_gate_$TAG.log 2>&1; echo "range rc=$?"; tail -2 gpurun_out/range_gate_$TAG.log | cut -c1-300
(timeout 900 python scripts/render_orbit.py --frames 200; timeout 900 python scripts/render_orbit.py --frames 200 --with-encoders) > gpurun_out/orbit_$TAG.txt 2>&1; grep -v amdgpu gpurun_out/orbit_$TAG.txt | tail -2
(timeout 600 python scripts/bench_dropin_train.py) 2>&1 | tail -1 | tee gpurun_out/dropin_train_$TAG.txt
