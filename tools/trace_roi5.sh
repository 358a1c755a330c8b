# kernel-trace of tools/roi5.py -> per-kernel mean time.  usage: bash tools/trace_roi5.sh TAG [env...]
T=${1:-t5}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O -o t --output-format csv -- python $R/tools/roi5.py 10 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
grep config5 $O/run.log
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$O/t_kernel_stats.csv")) if "roi_align" in r["Name"]]
for r in rows:
    print("%-60s calls %4s  avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
