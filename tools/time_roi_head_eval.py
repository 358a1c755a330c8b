"""Time ROIHead.forward (eval post-process) at B = 16 x 2,000 proposals, 21 classes, small heads.
   python tools/time_roi_head_eval.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd.ROIHead import ROIHead, ROIPooling, TwoMLPHead, FasterRCNNPredictor

dev = torch.device("cuda:0")
B, P, C, K = 16, 2000, 64, 21
rng = np.random.default_rng(0)
feats = [torch.randn(B, C, 37, 62, device=dev)]
def boxes(n):
    c = rng.uniform(0, 1, (n, 2)) * [1000, 600]; s = 16 * 2 ** rng.uniform(0, 5, (n, 2))
    b = np.concatenate([c - s / 2, c + s / 2], 1); b[:, 0::2] = b[:, 0::2].clip(0, 1000); b[:, 1::2] = b[:, 1::2].clip(0, 600)
    return torch.as_tensor(b.astype(np.float32)).to(dev)
props = [boxes(P) for _ in range(B)]
torch.manual_seed(0)
params = dict(pos_threshold=0.5, neg_threshold=0.5, num_sample=128, pos_fraction=0.25, regression_weights=[10., 10., 5., 5.],
              score_thresh=0.05, proposal_min_size=1e-3, nms_threshold=0.5, num_post_nms=100,
              roi_pooling=ROIPooling(['1'], [7, 7], 2), mlp_head=TwoMLPHead(C * 49, 256).to(dev),
              predictor=FasterRCNNPredictor(256, K).to(dev))
head = ROIHead(params).eval()
with torch.no_grad():
    for _ in range(2): res, _ = head(props, feats, (B, 600, 1000), None, [(600, 1000)] * B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): res, _ = head(props, feats, (B, 600, 1000), None, [(600, 1000)] * B)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("ROIHead eval forward: %.2f ms per batch of %d (detections/img %s)" % (dt * 1e3, B, [int(r['score'].numel()) for r in res][:4]))
