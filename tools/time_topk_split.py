"""top-k time against the number of slices:  python tools/time_topk_split.py B n k"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetection_amd import ops
B, n, k = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
s = ((torch.stack([torch.randperm(n, generator=g) for _ in range(B)]).float() / n) - 0.5) * 8
s = s.to(dev)
for ms_ in (1, 2, 4, 8, 16, 32, 0):
    for _ in range(3): ops.topk_desc(s, k, max_slices=ms_)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.topk_desc(s, k, max_slices=ms_)
    e1.record(); torch.cuda.synchronize()
    print("B=%d n=%d k=%d max_slices=%2d  %.4f ms" % (B, n, k, ms_, e0.elapsed_time(e1) / 20))
