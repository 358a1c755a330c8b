"""The NMS micro-benchmark of bench.py alone (realistic / clustered / lattice / bug-compatible sets).
   python tools/time_nms_sets.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
r = bench.nms_microbench(bench.Path(dev), dev)["nms_ms_per_img_batch16"]
for n, sets in r.items():
    for k, v in sets.items():
        print(n, k, json.dumps(v))
