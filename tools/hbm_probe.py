"""Achievable HBM rates on this box with stock torch kernels: pure write (fill_), copy (read+write), read (sum)."""
import torch
dev = torch.device("cuda:0")
n = 800 * 1024 * 1024  # 3.2 GB of fp32
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
def t(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(1.0)); print("fill  3.2GB: %.3f ms  %.2f TB/s written" % (ms, n * 4 / ms / 1e9))
ms = t(lambda: a.zero_()); print("zero  3.2GB: %.3f ms  %.2f TB/s written" % (ms, n * 4 / ms / 1e9))
ms = t(lambda: b.copy_(a)); print("copy  3.2GB: %.3f ms  %.2f TB/s read+write" % (ms, 2 * n * 4 / ms / 1e9))
ms = t(lambda: a.sum()); print("sum   3.2GB: %.3f ms  %.2f TB/s read" % (ms, n * 4 / ms / 1e9))
