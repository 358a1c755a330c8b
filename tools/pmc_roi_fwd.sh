# rocprofv3 SQ counters of the headline RoIAlign launch (tools/prof_kernels.py roi), separate --pmc passes.
# usage (GPU box, repo root): bash tools/pmc_roi_fwd.sh TAG   -> gpurun_out/TAG/*
T=${1:-fwdpmc}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
         "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $C -d $O/pmc$i -o p --output-format csv -- python $R/tools/prof_kernels.py roi 3 > $O/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pmc$i.log; exit 1; }
done
python $R/tools/pmc_summary.py $O > $O/summary.txt 2>&1
grep "roi_align_plane" $O/summary.txt
echo done
