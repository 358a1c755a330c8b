set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in base NORESOLVE PROLOGUE; do
  if [ $v = base ]; then unset FRCNN_EXP_LIB; else export FRCNN_EXP_LIB=$R/tools/bin/nms_$v.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace -d $R/gpurun_out/exp_scan_$v -o t --output-format csv -- python $R/tools/prof_kernels.py nms 3 > $R/gpurun_out/exp_scan_$v.log 2>&1
done
echo done
