#!/bin/bash
# A/B of environment knobs on the bench frame (GPU box): each argument is "NAME=VALUE[,NAME=VALUE]" (or "base"); prints ms per frame
# and the rows kernel's launch time.   usage: gpu_ab_env.sh base KPN_NO_POOL=1 ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do
for arm in "$@"; do
  envs=""; [ "$arm" != "base" ] && envs=$(echo $arm | tr ',' ' ')
  line=$(env $envs timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1)
  echo "$line" > gpurun_out/ab_${arm//[^A-Za-z0-9_=]/_}_$rep.json
  python - "$arm" <<PY
import json, sys
d = json.loads(open("gpurun_out/ab_" + "".join(c if (c.isalnum() or c in "_=") else "_" for c in sys.argv[1]) + "_$rep.json").read())
print(f"{sys.argv[1]:40s} rep $rep: {d['ms_per_step']:.3f} ms/frame, rows kernel {d['roofline']['avg_launch_ms']:.3f} ms x {d['roofline']['launches']} launches, frac {d['roofline']['frac']:.3f}, workspace {d['config']['render_workspace_bytes']/2**30:.2f} GiB, surplus {d['roofline']['surplus_launches']}")
PY
done; done
