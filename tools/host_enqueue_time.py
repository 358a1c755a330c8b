"""How long the HOST needs to enqueue one step of the headline path (no synchronisation inside the loop): if that
approaches the GPU's time per step the bench is host-bound.  python tools/host_enqueue_time.py [B]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from objectdetection_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
_lib.load()
path = bench.Path(dev)
cls, reg, feat = (t.to(dev) for t in bench.make_batch(0, B, "realistic"))
anchors = path.anchors(B, feat)
for _ in range(30):
    path.step(anchors, cls, reg, feat)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    path.step(anchors, cls, reg, feat)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("B=%d: host enqueue %.3f ms per step, GPU done after %.3f ms per step" % (B, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
