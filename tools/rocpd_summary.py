"""Summarise rocprofv3 rocpd (.db) output: kernel stats (top_kernels view) and, when present,
per-kernel means of collected counters.   python tools/rocpd_summary.py results.db [...]"""
import sqlite3
import sys

for path in sys.argv[1:]:
    c = sqlite3.connect(path)
    print("# %s" % path)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    if rows:
        print("kernel,calls,total_us,avg_us,percent")
        for name, calls, tot, avg, pct in rows:
            print('"%s",%d,%.3f,%.3f,%.2f' % (name.split("(")[0], calls, tot, avg, pct))
    rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                          "group by kernel_name, counter_name order by kernel_name, counter_name"))
    if rows:
        print("kernel,counter,dispatches,mean")
        for k, n, cnt, v in rows:
            print('"%s",%s,%d,%.6g' % (k.split("(")[0], n, cnt, v))
