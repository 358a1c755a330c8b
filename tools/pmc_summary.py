"""Summarise a rocprofv3 --pmc counter_collection.csv: mean per dispatch of each counter per kernel."""
import collections, csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print("%-42s %-40s n=%d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
