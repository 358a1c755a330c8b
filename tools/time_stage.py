"""Times one stage with HIP events (A/B experiments): python tools/time_stage.py roi|nms|topk [reps] [workload]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for _ in range(3): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): fn()
e1.record(); torch.cuda.synchronize()
print(what, workload, "ms", e0.elapsed_time(e1) / reps, "rois", int(n_keep.sum()))
