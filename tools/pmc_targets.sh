# rocprofv3 kernel trace + SQ / traffic counters of one stage driven by tools/prof_kernels.py: targets (K5
# iou_label_encode_kernel at BASELINE configs[2], the default), roipool (K7), topk, nms, bwd / bwd5 (RoIAlign backward at
# the headline shape / at configs[4]).  usage (GPU box, repo root): bash tools/pmc_targets.sh TAG [WHAT] -> gpurun_out/TAG/*
T=${1:-k5pmc}
WHAT=${2:-targets}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/tools/prof_kernels.py $WHAT 20 > $O/trace.log 2>&1 || exit 1
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
         "SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $C -d $O/pmc$i -o p --output-format csv -- python $R/tools/prof_kernels.py $WHAT 5 > $O/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pmc$i.log; }
done
python $R/tools/pmc_summary.py $O > $O/summary.txt 2>&1
cat $O/trace/*kernel_stats.csv > $O/kernel_stats.csv 2>/dev/null
echo done
