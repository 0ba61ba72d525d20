#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel-trace [+ pmc]) as markdown: per-kernel calls / total /
average duration, and per-kernel mean of every collected counter.  usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print("| kernel | calls | total ms | avg us | % |", file=out)
print("|---|---:|---:|---:|---:|", file=out)
for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"| `{name.split('(')[0]}` | {calls} | {tot / 1e3:.3f} | {avg:.1f} | {pct:.2f} |", file=out)
try:
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                      "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
except sqlite3.Error as e:
    rows = []
    print(f"\n(no counters: {e})", file=out)
if rows:
    print("\n| kernel | counter | dispatches | mean per dispatch | sum |", file=out)
    print("|---|---|---:|---:|---:|", file=out)
    for kn, cn, n, avg, tot in rows:
        print(f"| `{kn.split('(')[0]}` | {cn} | {n} | {avg:.6g} | {tot:.6g} |", file=out)
