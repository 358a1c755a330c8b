"""RoIs per pyramid level and their spans in cells at SURVEY 8d config 5 (which RoIAlign kernel gets what)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from objectdetection_amd import ops
dev = torch.device("cuda:0")
x = bench.config5_inputs(16, dev)
scores, deltas = ops.rpn_flatten(x["cls"], x["reg"])
tv, ti = ops.topk_desc(scores, 30000)
boxes, _, _, count = ops.decode_clip_filter(x["anchors"], deltas, ti, tv, x["Himg"], x["Wimg"], 1e-3)
keep, nk = ops.nms_sorted(boxes, count, 0.7, 2000)
props = ops.gather_boxes(boxes, keep, nk).reshape(-1, 4).cpu()
w, h = props[:, 2] - props[:, 0], props[:, 3] - props[:, 1]
lvl = torch.floor(4 + torch.log2(torch.sqrt(w * h) / 224) + 1e-6).clamp(3, 6).long()
for l in range(3, 7):
    m = lvl == l
    sc = 1.0 / (1 << l)
    sw, sh = w[m] * sc, h[m] * sc
    span = torch.maximum(sw, sh)
    print("P%d: %5d RoIs (%.1f%%)  span cells: median %.1f  p90 %.1f  max %.1f  >31: %.1f%%  >47: %.1f%%" % (
        l, int(m.sum()), 100.0 * m.float().mean(), span.median() if m.any() else 0, span.quantile(0.9) if m.any() else 0,
        span.max() if m.any() else 0, 100.0 * (span > 31).float().mean() if m.any() else 0,
        100.0 * (span > 47).float().mean() if m.any() else 0))
# finer histogram of the P3 spans (cells), for choosing window sizes
m = lvl == 3
span = torch.maximum(w[m], h[m]) / 8
for t in (9, 13, 17, 21, 25, 29, 33, 41, 49):
    print("P3 span <= %2d cells: %5.1f %%" % (t, 100.0 * (span <= t).float().mean()))
m = lvl == 4
span = torch.maximum(w[m], h[m]) / 16
for t in (17, 25, 33, 49):
    print("P4 span <= %2d cells: %5.1f %%" % (t, 100.0 * (span <= t).float().mean()))
