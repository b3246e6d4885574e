#!/usr/bin/env python
"""Summarise a rocprofv3 results database (rocpd sqlite) as a per-kernel table:
calls, total/avg/min/max duration, share.  Usage: rocprof_summary.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            f"from kernels group by {name_col} order by 3 desc"))
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, t, a, mn, mx in rows:
        short = n.split("(")[0][:70]
        lines.append(f"| {short} | {c} | {t/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*t/total:.1f} |")
        if mn < 20e3 and mx > 1e6:
            # the field kernels are launched once per possible batch of the capped row scratch; launches beyond the number of
            # batches a pass needs return at once (a few us).  Statistics of the launches that processed rows:
            c2, t2, a2, mn2, mx2 = list(cur.execute(f"select count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                                                    f"from kernels where {name_col} = ? and end-start > 50000", (n,)))[0]
            lines.append(f"| {short} — launches that processed rows (> 50 us) | {c2} | {t2/1e6:.3f} | {a2/1e3:.1f} | {mn2/1e3:.1f} | {mx2/1e3:.1f} | {100*t2/total:.1f} |")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
