# End-of-round refresh: GPU suite, bench in both schedules, rocprofv3 kernel stats of the same commands.
# usage (on the GPU box, from the repo root): bash tools/final_profiles.sh TAG   -> gpurun_out/TAG_*
set -e
T=${1:-final}
R=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $R/gpurun_out/${T}_tests.log 2>&1 || { tail -20 $R/gpurun_out/${T}_tests.log; exit 1; }
tail -1 $R/gpurun_out/${T}_tests.log
timeout -k 10 300 python bench.py > $R/gpurun_out/${T}_bench.json 2> $R/gpurun_out/${T}_bench.err
timeout -k 10 300 python bench.py --pipelined > $R/gpurun_out/${T}_bench_pipe.json 2> $R/gpurun_out/${T}_bench_pipe.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_ser -o s --output-format csv -- python $R/bench.py > $R/gpurun_out/${T}_ser.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_pipe -o p --output-format csv -- python $R/bench.py --pipelined > $R/gpurun_out/${T}_pipe.log 2>&1
echo done
