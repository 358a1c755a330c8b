"""Per-kernel mean time of the config-5 step under rocprofv3 (reads the rocpd db written by the run).
   rocprofv3 --kernel-trace -d DIR -o NAME -- python tools/time_config5.py ; python tools/prof_config5.py DIR/NAME_results.db"""
import sqlite3, sys
for path in sys.argv[1:]:
    c = sqlite3.connect(path)
    rows = c.execute("select name, grid_x, grid_y, count(*), avg(end-start)/1000.0 from kernels where name like '%roi_align%' "
                     "group by name, grid_x, grid_y order by 5 desc")
    print(path)
    for name, gx, gy, n, us in rows:
        print("  %-44s grid (%6d,%3d) x%d  %8.1f us" % (name.split("(")[0][:44], gx, gy, n, us))
