#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output): tools/kstats.py <results.db> [filter]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else "pgps"
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start) "
     f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc")
print("%-90s %7s %12s %12s %12s" % ("kernel", "calls", "avg_us", "min_us", "max_us"))
for name, n, avg, mn, mx, tot in con.execute(q):
    if flt in name:
        print("%-90s %7d %12.1f %12.1f %12.1f" % (name[:90], n, avg / 1e3, mn / 1e3, mx / 1e3))
