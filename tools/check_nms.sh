set -e
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "nms or rpn_forward or full_size or roi_head or determinism" > gpurun_out/s13_tests.log 2>&1 || { tail -20 gpurun_out/s13_tests.log; exit 1; }
tail -1 gpurun_out/s13_tests.log
timeout -k 10 120 python tools/time_roi_head_eval.py 2>&1 | tail -1
timeout -k 10 300 python bench.py > gpurun_out/s13_bench.json 2> gpurun_out/s13_bench.err
python -c "
import json;d=json.loads(open('gpurun_out/s13_bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['stage_ms']);s=d['secondary_workload'];print(s['value'],s['ms_per_step'],s['stage_ms']);print(d['nms_ms_per_img_n12000_batch16'])"
