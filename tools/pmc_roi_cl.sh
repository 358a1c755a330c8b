# rocprofv3 passes over config 5's MultiScaleRoIAlign on the lanes-are-channels path (tools/roi5.py with
# FRCNN_ROI_LEVELS_PATH=cl): one kernel trace, then counters in separate --pmc passes.
# usage (GPU box, repo root): bash tools/pmc_roi_cl.sh TAG [extra env assignments are inherited]  -> gpurun_out/TAG/*
T=${1:-clpmc}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$T
mkdir -p $O
export FRCNN_ROI_LEVELS_PATH=${FRCNN_ROI_LEVELS_PATH:-cl}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/tools/roi5.py 10 > $O/trace.log 2>&1 || exit 1
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
         "FETCH_SIZE" \
         "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $C -d $O/pmc$i -o p --output-format csv -- python $R/tools/roi5.py 3 > $O/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pmc$i.log; }
done
python $R/tools/pmc_summary.py $O > $O/summary.txt 2>&1
cat $O/trace/*kernel_stats.csv > $O/kernel_stats.csv 2>/dev/null
echo done
