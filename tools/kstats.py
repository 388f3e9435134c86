#!/usr/bin/env python3
"""tools/kstats.py <rocprofv3 results .db> [n] -- per-kernel totals of a --kernel-trace run (count, total ms, average us, share)"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute(f"select s.kernel_name,count(*),sum(d.end-d.start)/1e6,avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for r in rows[:top]:
    print(f"{r[0][:100]:100s} n={r[1]:6d} total {r[2]:9.1f} ms  avg {r[3]:10.1f} us {100 * r[2] / tot:5.1f}%")
