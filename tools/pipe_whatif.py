"""What would the pipelined step cost if flatten and decode were folded into the top-k launch?  Emulated by
hoisting their outputs out of the loop (the chain then is top-k -> NMS -> gather).  Experiment."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0"); path = bench.Path(dev); B = 16; ops = path.ops; pipe = path.pipe
cls, reg, feat = (t.to(dev) for t in bench.make_batch(0, B, "realistic"))
anchors = path.anchors(B, feat)
scores, deltas = ops.rpn_flatten([cls], [reg])
tv, ti = ops.topk_desc(scores, bench.PRE_NMS)
boxes, _, _, count = ops.decode_clip_filter(anchors, deltas, ti, tv, bench.IMG_H, bench.IMG_W, bench.MIN_SIZE)
torch.cuda.synchronize()
def step(full):
    def proposals():
        if full:
            s, d = ops.rpn_flatten([cls], [reg])
            v, i = ops.topk_desc(s, bench.PRE_NMS)
            bx, _, _, cnt = ops.decode_clip_filter(anchors, d, i, v, bench.IMG_H, bench.IMG_W, bench.MIN_SIZE)
        else:
            v, i = ops.topk_desc(scores, bench.PRE_NMS)
            bx, cnt = boxes, count
        keep, nk = ops.nms_sorted(bx, cnt, bench.NMS_THR, bench.POST_NMS)
        return ops.gather_boxes(bx, keep, nk), nk
    return pipe.submit(proposals, lambda p, n: path.pool.forward_padded({'1': feat}, p, n, [(bench.IMG_H, bench.IMG_W)]))
for full in (True, False, True, False):
    for _ in range(5): step(full)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): r = step(full)
    torch.cuda.synchronize()
    print("full chain" if full else "no flatten/decode launches", "%.3f ms/step" % ((time.perf_counter() - t0) / 50 * 1e3))
