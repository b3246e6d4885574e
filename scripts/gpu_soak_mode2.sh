#!/bin/bash
# Soak of a pair-tile rows kernel (MODE=3 default, MODE=2) over the product build and the placement variants.  Usage: [MODE=2] gpu_soak_mode2.sh <tag> <points> <repeats>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=$1; PTS=${2:-4000000}; REP=${3:-80}
: > gpurun_out/soak_mode2_$TAG.jsonl
for mask in dense ellipsoid; do timeout 600 python scripts/soak_mode2.py --mode ${MODE:-3} --points $PTS --repeats $REP --mask $mask >> gpurun_out/soak_mode2_$TAG.jsonl 2>/dev/null; done
for l in exp_libs/soak_*.so; do timeout 600 python scripts/soak_mode2.py --mode ${MODE:-3} --points $PTS --repeats $REP --lib $l >> gpurun_out/soak_mode2_$TAG.jsonl 2>/dev/null; done
python - <<PY
import json
rows=[json.loads(l) for l in open("gpurun_out/soak_mode2_$TAG.jsonl")]
print(len(rows), "builds/scenes;", "row evaluations %.3e" % sum(r["row_evaluations"] for r in rows), "; differing points", sum(r["differing_points_total"] for r in rows), "; off vs fp32", sum(r["points_off_vs_fp32_mfma"] for r in rows), "; seconds %.0f" % sum(r["seconds"] for r in rows))
PY
