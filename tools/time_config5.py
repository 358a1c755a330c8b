"""Stage times at SURVEY 8d config 5: 800x1344, levels 13x21..100x168 (A = 200,907), k = 30,000, P = 2,000,
C = 256, 4-level MultiScaleRoIAlign.   python tools/time_config5.py [B]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import _lib as _l
if os.environ.get("FRCNN_EXP_LIB"):  # A/B builds of the library (experiments only)
    _l.LIB_PATH = os.path.abspath(os.environ["FRCNN_EXP_LIB"])
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
print(json.dumps(bench.config5_microbench(torch.device("cuda:0"), B), indent=1))
