cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "density_first" 2>&1 | tail -1
for rep in 1 2; do for db in 0 -30; do for lib in product exp_libs/before_touch.so; do
  if [ "$lib" = "product" ]; then line=$(KPN_DENSITY_FIRST=1 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 --density-bias $db 2>/dev/null | tail -1)
  else line=$(KPN_DENSITY_FIRST=1 KPN_EXPERIMENT_LIB=$lib python scripts/bench_variant.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 --density-bias $db 2>/dev/null | tail -1); fi
  python - "$lib" "$db" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[3])
print(f"density first always, bias {sys.argv[2]:>4s} {sys.argv[1]:26s}: {d['ms_per_step']:.3f} ms/frame")
PY
done; done; done
