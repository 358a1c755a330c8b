"""Times one stage with HIP events (A/B experiments): python tools/time_stage.py roi|nms|topk [reps] [workload]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import objectdetection_amd._lib as _l
if os.environ.get("FRCNN_EXP_LIB"):  # A/B builds of the library (experiments only)
    _l.LIB_PATH = os.path.abspath(os.environ["FRCNN_EXP_LIB"])
import bench

what = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
workload = sys.argv[3] if len(sys.argv) > 3 else "realistic"
dev = torch.device("cuda:0"); path = bench.Path(dev); B = 16
cls, reg, feat = (t.to(dev) for t in bench.make_batch(0, B, workload))
anchors = path.anchors(B, feat); ops = path.ops
scores, deltas = ops.rpn_flatten([cls], [reg])
top_val, top_idx = ops.topk_desc(scores, bench.PRE_NMS)
boxes, _, _, count = ops.decode_clip_filter(anchors, deltas, top_idx, top_val, bench.IMG_H, bench.IMG_W, bench.MIN_SIZE)
keep, n_keep = ops.nms_sorted(boxes, count, bench.NMS_THR, bench.POST_NMS)
props = ops.gather_boxes(boxes, keep, n_keep)
fn = {"roi": lambda: path.pool.forward_padded({'1': feat}, props, n_keep, [(bench.IMG_H, bench.IMG_W)]),
      "nms": lambda: ops.nms_sorted(boxes, count, bench.NMS_THR, bench.POST_NMS),
      "topk": lambda: ops.topk_desc(scores, bench.PRE_NMS)}[what]
if what == "roi" and os.environ.get("ROI_STATS"):
    m = torch.arange(props.shape[1], device=dev)[None, :] < n_keep[:, None]
    w = ((props[..., 2] - props[..., 0]) / 16.0)[m]; h = ((props[..., 3] - props[..., 1]) / 16.0)[m]
    q = torch.tensor([0.1, 0.5, 0.9], device=dev)
    print("roi w px q10/50/90", torch.quantile(w, q).tolist(), "mean", float(w.mean()),
          "h", torch.quantile(h, q).tolist(), "mean", float(h.mean()),
          "frac both<14", float(((w < 14) & (h < 14)).float().mean()), "frac w<14", float((w < 14).float().mean()))
for _ in range(3): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): fn()
e1.record(); torch.cuda.synchronize()
print(what, workload, "ms", e0.elapsed_time(e1) / reps, "rois", int(n_keep.sum()))
