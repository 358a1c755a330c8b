"""Time RPN.forward (training) at B = 16, 600x1000, one 37x62 level, 100 GT boxes per image.
   python tools/time_training_branches.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd.RPN import RPN, AnchorGenerator, RPNHead

dev = torch.device("cuda:0")
B = 16
rng = np.random.default_rng(0)
torch.manual_seed(0)
params = dict(anchor_generator=AnchorGenerator(), rpn_head=RPNHead(64, 9).to(dev), num_pre_nms=12000,
              num_post_nms=2000, nms_threshold=0.7, proposal_min_size=1e-3, pos_threshold=0.7,
              neg_threshold=0.3, num_sample=256, pos_fraction=0.5)
rpn = RPN(params).train()
feats = [torch.randn(B, 64, 37, 62, device=dev)]
targets = []
for _ in range(B):
    wh = rng.uniform(16, 256, (100, 2)); xy = rng.uniform(0, 1, (100, 2)) * ([1000, 600] - wh)
    targets.append({'box': np.concatenate([xy, xy + wh], 1).tolist()})
for _ in range(2):
    props, loss = rpn((B, 600, 1000), feats, targets)
    (loss["cls_loss"] + loss["reg_loss"]).backward()   # also absorbs MIOpen's one-time search
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    props, loss = rpn((B, 600, 1000), feats, targets)
    (loss["cls_loss"] + loss["reg_loss"]).backward()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("RPN.forward(train)+backward: %.2f ms per batch of %d  loss %.4f %.4f" % (dt * 1e3, B, float(loss["cls_loss"]), float(loss["reg_loss"])))
rpn.eval()
with torch.no_grad():
    for _ in range(2): rpn((B, 600, 1000), feats, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): rpn((B, 600, 1000), feats, None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("RPN.forward(eval): %.2f ms per batch of %d" % (dt * 1e3, B))

