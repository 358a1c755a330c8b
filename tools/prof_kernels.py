"""Small driver for rocprofv3 runs: exercises one stage of the hot path a few times.

    python tools/prof_kernels.py roi|nms|topk|targets [reps]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if os.environ.get("FRCNN_EXP_LIB"):  # A/B builds of the library (experiments only)
    import objectdetection_amd._lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["FRCNN_EXP_LIB"])


def main():
    what = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    workload = sys.argv[3] if len(sys.argv) > 3 else "realistic"
    dev = torch.device("cuda:0")
    path = bench.Path(dev)
    B = 16
    cls, reg, feat = (t.to(dev) for t in bench.make_batch(0, B, workload))
    anchors = path.anchors(B, feat)
    ops = path.ops
    scores, deltas = ops.rpn_flatten([cls], [reg])
    top_val, top_idx = ops.topk_desc(scores, bench.PRE_NMS)
    boxes, _, _, count = ops.decode_clip_filter(anchors, deltas, top_idx, top_val, bench.IMG_H, bench.IMG_W, bench.MIN_SIZE)
    keep, n_keep = ops.nms_sorted(boxes, count, bench.NMS_THR, bench.POST_NMS)
    props = ops.gather_boxes(boxes, keep, n_keep)
    if what == "targets":  # K5 at BASELINE configs[2]: 16 x 20,646 anchors x 100 boxes
        g = torch.Generator().manual_seed(3)
        wh = torch.rand(B, 100, 2, generator=g) * 240 + 16
        xy = torch.rand(B, 100, 2, generator=g) * (torch.tensor([bench.IMG_W, bench.IMG_H]) - wh)
        gt = torch.cat([xy, xy + wh], 2).to(dev)
    torch.cuda.synchronize()
    for _ in range(reps):
        if what == "targets":
            ops.iou_label_encode(anchors, gt, None, 0.3, 0.7)
        elif what == "roi":
            path.pool.forward_padded({'1': feat}, props, n_keep, [(bench.IMG_H, bench.IMG_W)])
        elif what == "nms":
            ops.nms_sorted(boxes, count, bench.NMS_THR, bench.POST_NMS)
        elif what == "topk":
            ops.topk_desc(scores, bench.PRE_NMS)
    torch.cuda.synchronize()
    print("done", what, reps, int(n_keep.sum()))


if __name__ == "__main__":
    main()
