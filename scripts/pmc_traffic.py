#!/usr/bin/env python
"""profiles/geo_rows_traffic.json entry of one rows kernel from a PMC summary (scripts/gpu_pmc.sh -> pmc_<tag>_summary.txt):
HBM bytes per (point, view) row = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024 / rows (FETCH_SIZE doubled per MI355X_MICROARCH.md,
section HBM; separate --pmc passes).   usage: pmc_traffic.py <summary.txt> <kernel name> <rows in the profiled command> <tag>"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
summary, kernel, rows, tag = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
vals, cur = {}, None
for line in open(summary):
    m = re.match(r"## (.*)", line)
    if m:
        cur = m.group(1).strip()
        continue
    m = re.match(r"\s+(\S+)\s+(\S+)", line)
    if m and cur == kernel:
        vals[m.group(1)] = float(m.group(2))
fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
path = os.path.join(ROOT, "profiles", "geo_rows_traffic.json")
tr = json.load(open(path))
tr[kernel] = {"hbm_bytes_per_row": (2.0 * fetch + write) * 1024.0 / rows, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "rows": rows,
              "how": f"(2*FETCH_SIZE + WRITE_SIZE) KB * 1024 / rows over the dispatches of `bash scripts/gpu_pmc.sh {tag}` (separate --pmc passes, "
                     f"profiles/{tag}_pmc_counters.txt); FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM"}
json.dump(tr, open(path, "w"), indent=1)
print(kernel, tr[kernel])
