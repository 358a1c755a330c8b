"""Config-5 MultiScaleRoIAlign alone (16 x 2,000 RoIs, C = 256, levels 100x168 .. 13x21):
proposals are computed once, then the four level launches run `reps` times.  For rocprofv3 and A/B builds.
    python tools/roi5.py [reps] [B]        (FRCNN_EXP_LIB=path: an experimental build of the library)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import objectdetection_amd._lib as _l
if os.environ.get("FRCNN_EXP_LIB"):
    _l.LIB_PATH = os.path.abspath(os.environ["FRCNN_EXP_LIB"])
import bench
from objectdetection_amd import ops
from objectdetection_amd.ROIHead.roi_pooling import ROIPooling

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
warm = int(os.environ.get("ROI5_WARM", "3"))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
x = bench.config5_inputs(B, dev)
tv, ti = ops.topk_desc_nchw(x["cls"], 30000)
boxes, _, _, count = ops.decode_clip_filter_nchw(x["anchors"], x["reg"], ti, tv, x["Himg"], x["Wimg"], 1e-3)
keep, nk = ops.nms_sorted(boxes, count, 0.7, 2000)
props = ops.gather_boxes(boxes, keep, nk)
if os.environ.get("ROI5_DUMP"):
    import numpy as np
    np.savez(os.environ["ROI5_DUMP"], props=props.cpu().numpy(), nk=nk.cpu().numpy())
pool = ROIPooling(['1', '2', '3', '4'], [7, 7], 2)
feats = x["feats"][::-1]
if os.environ.get("ROI5_CL_MAPS"):   # maps already channel-innermost (torch.channels_last): the cl path reads them in place
    feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
fd = {str(i + 1): f for i, f in enumerate(feats)}
if os.environ.get("ROI5_COMPARE"):   # the cl path against the tile path on the same inputs
    ops.ROI_LEVELS_PATH = "tiles"
    a, ta = pool.forward_padded(fd, props, nk, [(x["Himg"], x["Wimg"])])
    ops.ROI_LEVELS_PATH = "cl"
    b_, tb = pool.forward_padded(fd, props, nk, [(x["Himg"], x["Wimg"])])
    n_ = int(ta)
    d = (a[:n_] - b_[:n_]).abs()
    print("compare: total %d / %d, max abs diff %.3e, max |a| %.3e, rows differing by > 1e-4: %d"
          % (n_, int(tb), float(d.max()), float(a[:n_].abs().max()), int((d.flatten(1).max(1).values > 1e-4).sum())))
    ops.ROI_LEVELS_PATH = os.environ.get("FRCNN_ROI_LEVELS_PATH", "auto")
fn = lambda: pool.forward_padded(fd, props, nk, [(x["Himg"], x["Wimg"])])
for _ in range(warm):
    out, total = fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out, total = fn()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
nb = int(total) * x["C"] * 49 * 4
print("config5 roi_align: %.4f ms per batch, %d RoIs, %.1f GB/s written, hbm_frac %.3f, checksum %.6e"
      % (ms, int(total), nb / ms / 1e6, nb / ms / 1e6 / 8000, float(out[:int(total)].double().sum())))
