"""Which proposal stage stretches RoIAlign when two batches are in flight?  The pipelined step is run with
parts of the proposal chain replaced by their precomputed outputs; RoIAlign's HIP-event time is reported
for each variant.  Experiment."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0"); path = bench.Path(dev); B = 16; ops = path.ops; pipe = path.pipe
cls, reg, feat = (t.to(dev) for t in bench.make_batch(0, B, "realistic"))
anchors = path.anchors(B, feat)
S, D = ops.rpn_flatten([cls], [reg])
TV, TI = ops.topk_desc(S, bench.PRE_NMS)
BX, _, _, CNT = ops.decode_clip_filter(anchors, D, TI, TV, bench.IMG_H, bench.IMG_W, bench.MIN_SIZE)
KEEP, NK = ops.nms_sorted(BX, CNT, bench.NMS_THR, bench.POST_NMS)
PROPS = ops.gather_boxes(BX, KEEP, NK)
torch.cuda.synchronize()

def make(stages):
    def proposals():
        s, d = ops.rpn_flatten([cls], [reg]) if "flatten" in stages else (S, D)
        v, i = ops.topk_desc(s, bench.PRE_NMS) if "topk" in stages else (TV, TI)
        if "decode" in stages:
            bx, _, _, cnt = ops.decode_clip_filter(anchors, d, i, v, bench.IMG_H, bench.IMG_W, bench.MIN_SIZE)
        else:
            bx, cnt = BX, CNT
        keep, nk = ops.nms_sorted(bx, cnt, bench.NMS_THR, bench.POST_NMS) if "nms" in stages else (KEEP, NK)
        p = ops.gather_boxes(bx, keep, nk) if "gather" in stages else PROPS
        return p, nk
    return proposals

def roi_timed(ev):
    def roi(p, n):
        e0 = torch.cuda.Event(enable_timing=True); e0.record(pipe.s_roi)
        r = path.pool.forward_padded({'1': feat}, p, n, [(bench.IMG_H, bench.IMG_W)])
        e1 = torch.cuda.Event(enable_timing=True); e1.record(pipe.s_roi)
        ev.append((e0, e1))
        return r
    return roi

variants = [("full chain", "flatten topk decode nms gather"), ("no proposal work", ""), ("top-k only", "topk"),
            ("nms only", "nms"), ("flatten + decode only", "flatten decode"), ("all but top-k", "flatten decode nms gather"),
            ("all but nms", "flatten topk decode gather"), ("full chain", "flatten topk decode nms gather")]
for name, st in variants:
    fn = make(set(st.split()))
    for _ in range(10): pipe.submit(fn, roi_timed([]))
    torch.cuda.synchronize(); ev = []; t0 = time.perf_counter()
    for _ in range(100): pipe.submit(fn, roi_timed(ev))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100 * 1e3
    ra = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    print("%-24s step %.3f ms   RoIAlign %.3f ms" % (name, dt, ra))
