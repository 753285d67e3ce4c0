"""Per-kernel durations from a rocprofv3 results database: python tools/kernstats.py <db> [name filter]."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 "
     f"from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 6 desc")
print('%-70s %6s %9s %9s %9s %9s' % ('kernel', 'calls', 'avg us', 'min us', 'max us', 'total ms'))
for r in c.execute(q):
    if flt in r[0]:
        print('%-70s %6d %9.1f %9.1f %9.1f %9.2f' % (r[0][:70], r[1], r[2], r[3], r[4], r[5]))
