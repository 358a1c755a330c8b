"""frcnn_nms_sorted_static_f32 -- the entry point that allocates nothing (its workspace is a device array inside
the library) -- against frcnn_nms_sorted_f32 with a caller workspace: same keep lists, bit for bit, on two
calls in a row, the second on a smaller input (the array keeps the first call's contents)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _boxes(B, n, seed, W=1000.0, H=600.0):
    rng = np.random.default_rng(seed)
    cx, cy = rng.uniform(0, W, (B, n)), rng.uniform(0, H, (B, n))
    w, h = 16 * 2 ** rng.uniform(0, 5, (B, n)), 16 * 2 ** rng.uniform(0, 5, (B, n))
    b = np.stack([(cx - w / 2).clip(0, W), (cy - h / 2).clip(0, H), (cx + w / 2).clip(0, W), (cy + h / 2).clip(0, H)], 2)
    return torch.as_tensor(b.astype(np.float32)).cuda()


def _both(lib, boxes, counts, thr, max_keep):
    B, n = boxes.shape[0], boxes.shape[1]
    P = ctypes.c_void_p
    st = P(torch.cuda.current_stream().cuda_stream)
    cnt = torch.as_tensor(counts, dtype=torch.int32).cuda() if counts is not None else None
    cp = P(cnt.data_ptr()) if cnt is not None else P(0)
    res = []
    for static in (False, True):
        keep = torch.full((B, max_keep), -7, dtype=torch.int32, device="cuda")
        nk = torch.full((B,), -7, dtype=torch.int32, device="cuda")
        if static:
            rc = lib.frcnn_nms_sorted_static_f32(P(boxes.data_ptr()), cp, B, n, thr, max_keep, P(keep.data_ptr()),
                                                 P(nk.data_ptr()), st)
        else:
            nbytes = lib.frcnn_nms_workspace_bytes(B, n)
            ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device="cuda")
            rc = lib.frcnn_nms_sorted_f32(P(boxes.data_ptr()), cp, B, n, thr, max_keep, P(keep.data_ptr()),
                                          P(nk.data_ptr()), P(ws.data_ptr()), nbytes, st)
        assert rc == 0, lib.frcnn_last_error_string()
        torch.cuda.synchronize()
        res.append((keep.cpu().numpy(), nk.cpu().numpy()))
    return res


def test_static_scratch_nms_matches_workspace_nms():
    from objectdetection_amd import _lib
    lib = _lib.load()
    assert lib.frcnn_nms_static_scratch_bytes() == 16 << 20
    # first call: the round path (n > 2048) over several rounds, ragged counts; second call: smaller in every way
    for B, n, counts, thr, max_keep, seed in [(5, 7000, [7000, 6500, 1, 0, 3000], 0.7, 600, 1),
                                              (2, 3000, None, 0.5, 2000, 2)]:
        (k0, n0), (k1, n1) = _both(lib, _boxes(B, n, seed), counts, thr, max_keep)
        assert np.array_equal(n0, n1)
        assert n0.max() > 100
        for b in range(B):
            assert np.array_equal(k0[b, :n0[b]], k1[b, :n1[b]])
    # a size whose workspace exceeds the array is refused, loudly, before anything is launched
    null = ctypes.c_void_p(0)
    rc = lib.frcnn_nms_sorted_static_f32(null, null, 64, 30000, 0.7, 2000, null, null, null)
    assert rc == -1 and b"scratch" in lib.frcnn_last_error_string()
