cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6)
for cap in 3072 2048 1920 1792; do
  line=$(KPN_ROW_SCRATCH_MIB=$cap timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1)
  python - "$cap" "$line" <<'PY' | tee -a gpurun_out/r06_c_workspace_ab.txt
import json, sys
d = json.loads(sys.argv[2])
print(f"scratch cap {sys.argv[1]:>5s} MiB: {d['ms_per_step']:.3f} ms/frame, rows kernel {d['roofline']['avg_launch_ms']:.3f} ms x {d['roofline']['launches']}, workspace {d['config']['render_workspace_bytes']/1e9:.3f} e9 B, surplus {d['roofline']['surplus_launches']}")
PY
done
for cap in 3072 1920; do
  line=$(KPN_ROW_SCRATCH_MIB=$cap timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1)
  python - "$cap" "$line" <<'PY' | tee -a gpurun_out/r06_c_workspace_ab.txt
import json, sys
d = json.loads(sys.argv[2])
print(f"(repeat) scratch cap {sys.argv[1]:>5s} MiB: {d['ms_per_step']:.3f} ms/frame, workspace {d['config']['render_workspace_bytes']/1e9:.3f} e9 B")
PY
done
