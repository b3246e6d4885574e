#!/bin/bash
# round-4 measurement evidence beside the suite / bench run (gpu_round4_run.sh): PMC counter passes of the bench command (traffic
# json regenerated from them), soak of the default rows kernel, eager PyTorch on the same GPU, the 200-frame orbit with and
# without encoder stand-ins, the training step through the drop-in.   usage: gpu_round4_evidence.sh TAG
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r04_e}
bash scripts/gpu_pmc.sh $TAG 2>&1 | tail -3
ROWS=$(python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r4_pmc_rows.json").read().strip().splitlines()[-1])
print(int(d["config"]["valid_rows_per_step"] * 2))
PY
)
python scripts/pmc_traffic.py gpurun_out/pmc_${TAG}_summary.txt k_geo_rows_f2p $ROWS $TAG | cut -c1-200
cp profiles/geo_rows_traffic.json gpurun_out/geo_rows_traffic.json
grep -A12 "## k_geo_rows_f2p" gpurun_out/pmc_${TAG}_summary.txt | head -40
(timeout 900 python scripts/fuzz_parity.py 200) > gpurun_out/fuzz_$TAG.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/fuzz_$TAG.log | cut -c1-400; cp gpurun_out/fuzz_parity.json gpurun_out/fuzz_parity_200_scenes_default_$TAG.json
(KPN_GEO_ROWS_MODE=0 KPN_FUSE_MODE=0 timeout 900 python scripts/fuzz_parity.py 200) > gpurun_out/fuzz32_$TAG.log 2>&1; echo "fuzz fp32 rc=$?"; tail -1 gpurun_out/fuzz32_$TAG.log | cut -c1-400; cp gpurun_out/fuzz_parity.json gpurun_out/fuzz_parity_200_scenes_fp32_kernels_$TAG.json
(timeout 600 python scripts/soak_mode2.py --mode 3 --repeats 8000) 2>/dev/null | tail -1 | tee gpurun_out/soak_mode3_$TAG.jsonl
(timeout 600 python scripts/soak_mode2.py --mode 2 --repeats 600) 2>/dev/null | tail -1 | tee -a gpurun_out/soak_mode3_$TAG.jsonl
(timeout 600 python scripts/bench_torch_eager.py) > gpurun_out/eager_$TAG.json 2>gpurun_out/eager_$TAG.err; tail -1 gpurun_out/eager_$TAG.json | cut -c1-300
(timeout 900 python scripts/render_orbit.py --frames 200; timeout 900 python scripts/render_orbit.py --frames 200 --with-encoders) > gpurun_out/orbit_$TAG.txt 2>&1; grep -v amdgpu gpurun_out/orbit_$TAG.txt | tail -2
(timeout 600 python scripts/bench_dropin_train.py) 2>&1 | tail -2 | tee gpurun_out/dropin_train_$TAG.txt
