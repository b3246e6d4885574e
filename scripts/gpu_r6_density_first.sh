#!/bin/bash
# Round 6: the density-first per-point passes on the MI355X — the bit-equality test, the parity suite, and an A/B of the bench frame
# (headline density and the partly empty hulls) against the fused per-point kernel (KPN_NO_DENSITY_FIRST=1).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "density_first or zero_density or capped_row or headline or goldens or golden" 2>&1 | tail -5 | tee gpurun_out/r06_a_density_first_tests.txt
for db in 0 -20 -30; do
  for rep in 1 2; do
    for arm in base KPN_NO_DENSITY_FIRST=1; do
      envs=""; [ "$arm" != "base" ] && envs="$arm"
      line=$(env $envs timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 2 --density-bias $db 2>/dev/null | tail -1)
      python - "$arm" "$db" "$rep" "$line" <<'PY' | tee -a gpurun_out/r06_a_density_first_ab.txt
import json, sys
d = json.loads(sys.argv[4])
print(f"density_bias {sys.argv[2]:>4s} {sys.argv[1]:26s} rep {sys.argv[3]}: {d['ms_per_step']:.3f} ms/frame, rows kernel {d['roofline']['avg_launch_ms']:.3f} ms x {d['roofline']['launches']}, sigma_zero_fraction {d.get('sigma_zero_fraction')}")
PY
    done
  done
done
