# Round-end refresh (GPU box, repo root):  bash tools/final_profiles.sh TAG   -> gpurun_out/TAG_*
#   1. GPU test suite
#   2. bench.py as the driver runs it (all extras)                                  -> TAG_bench.json
#   3. rocprofv3 --kernel-trace --stats of `bench.py --no-extras --cpu-sample 0`: ONE workload, ONE schedule, so
#      the AverageNs of roi_align_plane_kernel<4,64> IS the headline launch            -> TAG_ser/s_kernel_stats.csv
#   4. the same for --pipelined                                                       -> TAG_pipe/p_kernel_stats.csv
#   5. HBM traffic of the dominant kernel: two --pmc passes (WRITE_SIZE, FETCH_SIZE)   -> TAG_pmc_w, TAG_pmc_f
#   6. derived roofline figure next to the stats                                      -> TAG_roofline.txt
#   7. kernel stats of the whole default bench.py run: the extras (K5 targets, K7 max RoIPool, RoIAlign backward,
#      config 5 incl. both RoIAlign families, selective top-k, NMS sets)                -> TAG_extras/x_kernel_stats.csv
T=${1:-final}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/${T}_tests.log 2>&1 || { tail -20 $O/${T}_tests.log; exit 1; }
tail -1 $O/${T}_tests.log
timeout -k 10 400 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err || { tail -5 $O/${T}_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/${T}_ser -o s --output-format csv -- python $R/bench.py --no-extras --cpu-sample 0 > $O/${T}_ser.json 2> $O/${T}_ser.err || { tail -5 $O/${T}_ser.err; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/${T}_pipe -o p --output-format csv -- python $R/bench.py --no-extras --cpu-sample 0 --pipelined > $O/${T}_pipe.json 2> $O/${T}_pipe.err || { tail -5 $O/${T}_pipe.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/${T}_pmc_w -o w --output-format csv -- python $R/tools/prof_kernels.py roi 3 > $O/${T}_pmc_w.log 2>&1 || { tail -5 $O/${T}_pmc_w.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/${T}_pmc_f -o f --output-format csv -- python $R/tools/prof_kernels.py roi 3 > $O/${T}_pmc_f.log 2>&1 || { tail -5 $O/${T}_pmc_f.log; exit 1; }
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $O/${T}_extras -o x --output-format csv -- python $R/bench.py --cpu-sample 0 > $O/${T}_extras.json 2> $O/${T}_extras.err || { tail -5 $O/${T}_extras.err; exit 1; }
cd $R && python tools/roofline_from_stats.py $T > $O/${T}_roofline.txt && cat $O/${T}_roofline.txt
echo done
