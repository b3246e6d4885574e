#!/bin/bash
# A/B of library builds on the bench frame (GPU box): arguments are "product" or paths of experimental builds (exp_libs/*.so).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "product" ]; then line=$(timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1)
  else line=$(KPN_EXPERIMENT_LIB=$lib timeout 600 python scripts/bench_variant.py --no-secondary --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1); fi
  echo "$line" | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print(f\"$lib rep $rep: {d['ms_per_step']:.3f} ms/frame, rows kernel {d['roofline']['avg_launch_ms']:.3f} ms x {d['roofline']['launches']}, frac {d['roofline']['frac']:.3f}, clock {d['roofline'].get('effective_clock_ghz')}, per-point kernel {d.get('secondary', {}).get('fuse_kernel_ms', '')}\")"
done; done
