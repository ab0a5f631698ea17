"""Per-kernel summary of a rocprofv3 rocpd (sqlite) kernel trace: python tools/rocpd_stats.py results.db [--timeline N]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 "
     "from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 6 desc" % (kd, ks))
print("%-36s %6s %10s %10s %10s %10s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_ms"))
for r in db.execute(q):
    print("%-36s %6d %10.1f %10.1f %10.1f %10.2f" % (r[0].split('(')[0][:36], r[1], r[2], r[3], r[4], r[5]))
