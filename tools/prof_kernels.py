"""Small driver for rocprofv3 runs: exercises one stage of the hot path a few times.

    python tools/prof_kernels.py roi|nms|topk|targets|roipool|bwd|bwd5 [reps]
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
    if what == "roipool":  # K7 at the headline shape (bench.roipool_microbench's boxes)
        rng = np.random.default_rng(11)
        n = B * bench.POST_NMS
        cx, cy = rng.uniform(0, bench.IMG_W, n), rng.uniform(0, bench.IMG_H, n)
        w, h = 16 * 2 ** rng.uniform(0, 5, n), 16 * 2 ** rng.uniform(0, 5, n)
        pp = np.stack([(cx - w / 2).clip(0, bench.IMG_W), (cy - h / 2).clip(0, bench.IMG_H), (cx + w / 2).clip(0, bench.IMG_W),
                       (cy + h / 2).clip(0, bench.IMG_H)], 1).astype(np.float32)
        pool_props = torch.as_tensor(pp.reshape(B, bench.POST_NMS, 4)).to(dev)
    if what == "bwd":  # RoIAlign backward at the headline shape, the step's own proposals
        cfg = ((7, 7), 2, (1.0 / 16,), 4, 4)
        total = int(n_keep.sum())
        g_out = torch.randn((B * bench.POST_NMS, feat.shape[1], 7, 7), device=dev)
        shapes = [tuple(feat.shape)]
    if what == "bwd5":  # RoIAlign backward at BASELINE configs[4] (four FPN levels)
        x = bench.config5_inputs(B, dev)
        tv, ti = ops.topk_desc_nchw(x["cls"], 30000)
        b5, _, _, c5 = ops.decode_clip_filter_nchw(x["anchors"], x["reg"], ti, tv, x["Himg"], x["Wimg"], 1e-3)
        k5, nk5 = ops.nms_sorted(b5, c5, 0.7, 2000)
        p5 = ops.gather_boxes(b5, k5, nk5)
        from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
        feats5 = [f.requires_grad_(True) for f in x["feats"][::-1]]
        pool5 = ROIPooling(['1', '2', '3', '4'], [7, 7], 2)
        out5, _ = pool5.forward_padded({str(i + 1): f for i, f in enumerate(feats5)}, p5, nk5, [(x["Himg"], x["Wimg"])])
        g5 = torch.randn_like(out5)
    torch.cuda.synchronize()
    for _ in range(reps):
        if what == "roipool":
            ops.roi_pool_batched(feat, pool_props, None, (7, 7), 1.0 / 16)
        elif what == "bwd":
            ops.roi_align_levels_backward(cfg, props, n_keep, g_out, shapes)
        elif what == "bwd5":
            for f in feats5:
                f.grad = None
            out5.backward(g5, retain_graph=True)
        elif what == "targets":
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
