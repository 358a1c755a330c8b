set -e
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "roi_align or multiscale or roi_pooling or full_size or roi_head" > gpurun_out/s15_tests.log 2>&1 || { tail -20 gpurun_out/s15_tests.log; exit 1; }
tail -1 gpurun_out/s15_tests.log
timeout -k 10 200 python tools/time_config5.py > gpurun_out/s15_c5.log 2>&1
grep -v "^ *\"\|^[{}]" gpurun_out/s15_c5.log | tail -3
python -c "
import json,re
t=open('gpurun_out/s15_c5.log').read()
i=t.find('{')
d=json.loads(t[i:]) if i>=0 else {}
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,dict)}); print(d.get('stage_ms'))
" || true
