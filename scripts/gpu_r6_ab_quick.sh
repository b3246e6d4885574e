#!/bin/bash
# Round 6 quick A/B on the GPU box: density-first tests, then the bench frame at three density biases in the three modes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "density_first or zero_density" 2>&1 | tail -3
for db in 0 -20 -30; do for arm in KPN_DENSITY_FIRST=2 KPN_DENSITY_FIRST=1 KPN_DENSITY_FIRST=0; do
  line=$(env $arm timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 --density-bias $db 2>/dev/null | tail -1)
  python - "$arm" "$db" "$line" <<'PY' | tee -a gpurun_out/r06_density_first_ab.txt
import json, sys
d = json.loads(sys.argv[3])
print(f"density_bias {sys.argv[2]:>4s} {sys.argv[1]:22s}: {d['ms_per_step']:.3f} ms/frame, rows kernel {d['roofline']['avg_launch_ms']:.3f} ms, sigma_zero_fraction {d.get('sigma_zero_fraction'):.3f}, passes df/fused {d['density_first']['passes_density_first']}/{d['density_first']['passes_fused_kernel']}")
PY
done; done
