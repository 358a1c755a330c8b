"""RoIAlign backward at the headline shape alone.  python tools/time_roi_bwd.py [reps]   (FRCNN_EXP_LIB for A/B builds)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import objectdetection_amd._lib as _l
if os.environ.get("FRCNN_EXP_LIB"):
    _l.LIB_PATH = os.path.abspath(os.environ["FRCNN_EXP_LIB"])
from objectdetection_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
B, P, C, H, W = 16, 2000, 512, 37, 62
rng = np.random.default_rng(17)
cx, cy = rng.uniform(0, 1000, (B, P)), rng.uniform(0, 600, (B, P))
bw, bh = 16 * 2 ** rng.uniform(0, 5, (B, P)), 16 * 2 ** rng.uniform(0, 5, (B, P))
props = torch.as_tensor(np.stack([(cx - bw / 2).clip(0, 1000), (cy - bh / 2).clip(0, 600), (cx + bw / 2).clip(0, 1000),
                                  (cy + bh / 2).clip(0, 600)], 2).astype(np.float32)).to(dev)
g = torch.randn((B * P, C, 7, 7), device=dev)
cfg = ((7, 7), 2, (1.0 / 16,), 4, 4)
fn = lambda: ops.roi_align_levels_backward(cfg, props, None, g, [(B, C, H, W)])
out = fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out = fn()
e1.record(); torch.cuda.synchronize()
print("roi_align backward %.3f ms, checksum %.6e" % (e0.elapsed_time(e1) / reps, float(out[0].double().sum())))
