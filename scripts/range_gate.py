#!/usr/bin/env python
"""The range / precision sweep of tests/test_gpu_range.py as a report (GPU box): per case the worst error of the default kernels
(rows mode 3 + fuse mode 1 under the range guard) and of the fp32 kernels (modes 0 / 0) against the C oracle, whether the
fp32-range kernels took over, and the parity gate's verdicts.  Writes gpurun_out/range_gate.json.  TEST INFRASTRUCTURE (imports
the oracle)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keypointnerf_amd import ops  # noqa: E402
from tests.test_gpu_range import range_cases, run_case  # noqa: E402

rows = []
for name, scene, sd, must in range_cases():
    r = run_case(ops, scene, sd, nominal=(must is False))
    r["case"] = name
    rows.append(r)
    e, e32 = max(r["err_default"].values()), max(r["err_fp32"].values())
    print(f"{name:58s} default {e:9.2e}  fp32 {e32:9.2e}  took over: {str(r['took_over']):5s} gate: default {r['gate_default_ok']} fp32 {r['gate_fp32_ok']}"
          + ("" if r["oracle_finite"] else "  (oracle not finite)"), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"what": "operand-range sweep: default kernels (two fp16 pieces, range guard) and fp32 kernels vs the C oracle, 400 rays x (24 + 16) samples, V = 3",
           "cases": rows}, open(os.path.join(ROOT, "gpurun_out", "range_gate.json"), "w"), indent=1)
