"""RoIAlign backward at the headline shape alone, checked through the adjoint identity against the forward kernel.
python tools/time_roi_bwd.py [reps]   (FRCNN_EXP_LIB for A/B builds)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import objectdetection_amd._lib as _l
if os.environ.get("FRCNN_EXP_LIB"):
    _l.LIB_PATH = os.path.abspath(os.environ["FRCNN_EXP_LIB"])
from objectdetection_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
B, P, C, H, W = 16, 2000, 512, 37, 62
rng = np.random.default_rng(17)
cx, cy = rng.uniform(0, 1000, (B, P)), rng.uniform(0, 600, (B, P))
bw, bh = 16 * 2 ** rng.uniform(0, 5, (B, P)), 16 * 2 ** rng.uniform(0, 5, (B, P))
props = torch.as_tensor(np.stack([(cx - bw / 2).clip(0, 1000), (cy - bh / 2).clip(0, 600), (cx + bw / 2).clip(0, 1000),
                                  (cy + bh / 2).clip(0, 600)], 2).astype(np.float32)).to(dev)
torch.manual_seed(3)
g = torch.randn((B * P, C, 7, 7), device=dev)
cfg = ((7, 7), 2, (1.0 / 16,), 4, 4)
fn = lambda: ops.roi_align_levels_backward(cfg, props, None, g, [(B, C, H, W)])
out = fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out = fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("roi_align backward %.3f ms (%.0f GB/s of grad_out), checksum %.6e" % (ms, g.numel() * 4 / ms / 1e6, float(out[0].double().sum())))
out2 = fn()
print("deterministic:", bool(torch.equal(out[0], out2[0])))
# <RoIAlign(f), g> = <f, RoIAlign^T(g)> per (image, channel), the forward being the parity-tested plane kernel
f = torch.randn((B, C, H, W), device=dev)
y, total = ops._roi_align_levels(cfg, props, None, [f])
lhs = (y.double() * g.double()).sum(dim=(2, 3)).view(B, P, C).sum(dim=1)
rhs = (f.double() * out[0].double()).sum(dim=(2, 3))
scale = (y.double().abs() * g.double().abs()).sum(dim=(2, 3)).view(B, P, C).sum(dim=1)
err = ((lhs - rhs).abs() / scale).max()
print("adjoint identity: max |lhs - rhs| / sum|terms| = %.3e over %d (image, channel) pairs" % (float(err), B * C))
