#!/usr/bin/env python
"""Aggregate rocprofv3 counter_collection.csv files: per kernel, sum of each counter over dispatches.
Usage: pmc_summary.py <dir> [kernel-substring ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    filt = sys.argv[2:] or ["k_geo_rows", "k_fuse_color", "k_mask_compact", "k_density_h", "k_colour_h", "k_row_records"]
    agg = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            ndisp[k].add((os.path.basename(f), row["Dispatch_Id"]))
    for k in agg:
        if not any(s in k for s in filt):
            continue
        print(f"## {k}")
        for c, v in sorted(agg[k].items()):
            print(f"  {c:34s} {v:.6g}")


if __name__ == "__main__":
    main()
