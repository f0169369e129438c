#!/usr/bin/env python3
"""Average duration per kernel from a rocprofv3 results database (rocprofv3 --kernel-trace -d DIR -o NAME -- prog).

usage: kernel_times.py DIR_OR_DB [--step]     --step: timeline of the last train step (between the last two k_dw_reduce)
"""
import glob
import sqlite3
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    f = path if path.endswith(".db") else sorted(glob.glob(path + "/**/*.db", recursive=True))[-1]
    c = sqlite3.connect(f).cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(c.execute(f"select s.display_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    if "--step" in sys.argv:
        idx = [i for i, r in enumerate(rows) if "k_dw_reduce" in r[0]]
        seg = rows[idx[-2] + 1: idx[-1] + 1]
        t0 = seg[0][1]
        for n, s, e in seg:
            if e - s > 20000:
                print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  {n[:80]}")
        print(f"step {(seg[-1][2] - t0) / 1e3:.1f} us")
        return
    agg = defaultdict(list)
    for n, s, e in rows:
        agg[n].append(e - s)
    tot = sum(sum(v) for v in agg.values())
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:25]:
        print(f"{sum(v) / 1e6:9.3f} ms {100 * sum(v) / tot:5.1f}%  n={len(v):4d}  avg {sum(v) / len(v) / 1e3:9.1f} us  min {min(v) / 1e3:9.1f}  {n[:90]}")


if __name__ == "__main__":
    main()
