"""NMS time on the headline boxes as a function of the bound (64-candidate blocks covered by the rounds).
python tools/nms_bound_sweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from objectdetection_amd import ops, _lib
dev = torch.device("cuda:0")
_lib.load()
path = bench.Path(dev)
cls, reg, feat = (t.to(dev) for t in bench.make_batch(0, 16, "realistic"))
anchors = path.anchors(16, feat)
tv, ti = ops.topk_desc_nchw([cls], bench.PRE_NMS)
boxes, _, _, count = ops.decode_clip_filter_nchw(anchors, [reg], ti, tv, bench.IMG_H, bench.IMG_W, bench.MIN_SIZE)
ref = ops.nms_sorted(boxes, count, bench.NMS_THR, bench.POST_NMS)


class Fixed(ops.NmsRounds):
    def __init__(self, b):
        super().__init__()
        self.b = b

    def bound(self, key, B):
        return self.b, self._slot(key, B)


for b in (0, 188, 64, 48, 40, 38, 36, 34, 33, 32, 24, 16):
    r = Fixed(b)
    for _ in range(3):
        keep, nk = ops.nms_sorted(boxes, count, bench.NMS_THR, bench.POST_NMS, rounds=r)
    torch.cuda.synchronize()
    assert torch.equal(nk, ref[1]) and torch.equal(keep, ref[0])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        keep, nk = ops.nms_sorted(boxes, count, bench.NMS_THR, bench.POST_NMS, rounds=r)
    e1.record(); torch.cuda.synchronize()
    print("max_blocks %3d: %.4f ms per batch   (blocks with work: max %d)" % (b, e0.elapsed_time(e1) / 20, int(r._slot(list(r._used)[0], 16).max())))
