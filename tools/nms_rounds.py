"""Per-launch NMS kernel durations of one bench step from a rocpd db."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
ra = [i for i, r in enumerate(rows) if "roi_align" in r[0]]
i0 = ra[len(ra) // 2]
d = [round((r[2] - r[1]) / 1e3, 1) for r in rows[i0 + 1:ra[len(ra) // 2 + 1] + 1] if "nms" in r[0]]
print(d[:8], "sum", round(sum(d), 1))
