"""Stage times at SURVEY 8d config 5: 800x1344, levels 13x21..100x168 (A = 200,907), k = 30,000, P = 2,000,
C = 256, 4-level MultiScaleRoIAlign.   python tools/time_config5.py [B]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import ops
from objectdetection_amd.RPN.anchor_generator import AnchorGenerator
from objectdetection_amd.ROIHead.roi_pooling import ROIPooling

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
Himg, Wimg, C, NA = 800, 1344, 256, 9
fsz = [(13, 21), (25, 42), (50, 84), (100, 168)]
g = torch.Generator().manual_seed(7)
A = sum(h * w for h, w in fsz) * NA
perm = (torch.stack([torch.randperm(A, generator=g) for _ in range(B)]).float() / A - 0.5) * 8
cls, reg, feats, off = [], [], [], 0
ag = AnchorGenerator()
for h, w in fsz:
    n = h * w * NA
    cls.append(perm[:, off:off + n].reshape(B, h, w, NA).permute(0, 3, 1, 2).contiguous().to(dev)); off += n
    feats.append(torch.randn(B, C, h, w, generator=g).to(dev))
anchors = torch.cat([a[0].reshape(-1, 4) for a in ag((B, Himg, Wimg), feats)], 0)
# proposal-realistic deltas under the reference decode (see bench.synth_image)
wa, ha = (anchors[:, 2] - anchors[:, 0]).cpu(), (anchors[:, 3] - anchors[:, 1]).cpu()
off = 0
for h, w in fsz:
    n = h * w * NA
    cx, cy = torch.rand(B, n, generator=g) * Wimg, torch.rand(B, n, generator=g) * Himg
    # object size 16 .. 724 px, aspect ratio 1:3 .. 3:1 (independent w and h would make a third of the
    # fine-level RoIs 5:1 and longer, which no detector proposes)
    size = 16 * 2 ** (torch.rand(B, n, generator=g) * 5.5)
    ratio = 2 ** ((torch.rand(B, n, generator=g) * 2 - 1) * 1.585)
    wt, ht = size * ratio.sqrt(), size / ratio.sqrt()
    w_, h_ = wa[off:off + n], ha[off:off + n]
    d = torch.stack([(cx - w_ / 2) / w_, (cy - h_ / 2) / h_, torch.log(wt / w_), torch.log(ht / h_)], 2)
    reg.append(d.reshape(B, h, w, 4 * NA).permute(0, 3, 1, 2).contiguous().to(dev)); off += n
pool = ROIPooling(['1', '2', '3', '4'], [7, 7], 2)
fd = {str(i + 1): f for i, f in enumerate(feats[::-1])}   # fine to coarse

def step(ev):
    def mark(n):
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.append((n, e))
    mark("start")
    scores, deltas = ops.rpn_flatten(cls, reg); mark("flatten")
    tv, ti = ops.topk_desc(scores, 30000); mark("topk")
    boxes, _, _, count = ops.decode_clip_filter(anchors, deltas, ti, tv, Himg, Wimg, 1e-3); mark("decode")
    keep, nk = ops.nms_sorted(boxes, count, 0.7, 2000); mark("nms")
    props = ops.gather_boxes(boxes, keep, nk); mark("gather")
    out, total = pool.forward_padded(fd, props, nk, [(Himg, Wimg)]); mark("roi_align")
    return nk, total

for _ in range(2): step([])
acc = {}
for _ in range(5):
    ev = []; nk, total = step(ev); torch.cuda.synchronize()
    for (a, e0), (b, e1) in zip(ev[:-1], ev[1:]): acc.setdefault(b, []).append(e0.elapsed_time(e1))
print("config5 B=%d A=%d keeps/img %.0f rois %d" % (B, A, nk.float().mean(), int(total)))
print({k: round(float(np.mean(v)), 3) for k, v in acc.items()}, "sum %.3f ms" % sum(np.mean(v) for v in acc.values()))
