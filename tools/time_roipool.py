"""K7 timing alone: python tools/time_roipool.py  (FRCNN_EXP_LIB for A/B builds)"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import objectdetection_amd._lib as _l
if os.environ.get("FRCNN_EXP_LIB"):
    _l.LIB_PATH = os.path.abspath(os.environ["FRCNN_EXP_LIB"])
import bench
dev = torch.device("cuda:0")
print(json.dumps(bench.roi_pool_microbench(bench.Path(dev), dev), indent=1))
