"""Does RoIAlign's time depend on WHICH output buffer it writes?  (a) the same 3.2 GB block every launch,
(b) two blocks alternating, (c) three.  Experiment."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0"); path = bench.Path(dev); B = 16; ops = path.ops
cls, reg, feat = (t.to(dev) for t in bench.make_batch(0, B, "realistic"))
anchors = path.anchors(B, feat)
S, D = ops.rpn_flatten([cls], [reg])
TV, TI = ops.topk_desc(S, bench.PRE_NMS)
BX, _, _, CNT = ops.decode_clip_filter(anchors, D, TI, TV, bench.IMG_H, bench.IMG_W, bench.MIN_SIZE)
KEEP, NK = ops.nms_sorted(BX, CNT, bench.NMS_THR, bench.POST_NMS)
PROPS = ops.gather_boxes(BX, KEEP, NK)
torch.cuda.synchronize()
def run(hold, reps=40):
    held = []
    for _ in range(5):
        held.append(path.pool.forward_padded({'1': feat}, PROPS, NK, [(bench.IMG_H, bench.IMG_W)]))
        held = held[-hold:] if hold else []
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        held.append(path.pool.forward_padded({'1': feat}, PROPS, NK, [(bench.IMG_H, bench.IMG_W)]))
        held = held[-hold:] if hold else []
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for hold in (0, 1, 2, 0, 1, 2):
    print("results held: %d -> %.4f ms per launch (back to back)" % (hold, run(hold)))
