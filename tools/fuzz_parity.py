"""Randomised parity runs: the HIP path against the CPU oracle on shapes nobody chose by hand.
    python tools/fuzz_parity.py [seconds] [seed]
Draws (op, shape, data kind) at random until the time is up: top-k (every plan: registers, full sort, split, selective),
the RPN proposal half through its module (one-call path and staged), NMS (lazy, rounds, bounded rounds + tail through the C ABI), multi-level RoIAlign forward and
backward (plane, window, direct kernels; ragged counts; channel tails), max RoIPool.  Index / integer results must be
bit-exact, floating point within 1e-5 (tests/test_gpu_parity.py's bars).  Prints one line per failure with the draw's
seed so that it can be replayed (FUZZ_ONLY=op FUZZ_CASE=seed), and a summary; exit status 1 on any failure.
The oracle is test infrastructure: this tool is a checker, nothing in the product path imports it."""
import os, sys, time, traceback
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from objectdetection_amd import ops, _lib
from tests.test_gpu_parity import (T, tie_free_scores, realistic_boxes, lattice_boxes, assert_bits_equal,
                                    synth_head_outputs, _FixedHead)

dev = torch.device("cuda:0")
RTOL = 1e-5


def scores_of(rng, n, kind):
    if kind == "tie_free":
        return tie_free_scores(rng, n)
    if kind == "ties":
        return (rng.integers(-40, 40, n) * 0.125).astype(np.float32)
    s = rng.standard_normal(n).astype(np.float32)   # specials
    s[rng.integers(0, n, max(1, n // 50))] = np.inf
    s[rng.integers(0, n, max(1, n // 50))] = -np.inf
    s[rng.integers(0, n, max(1, n // 70))] = np.nan
    s[rng.integers(0, n, max(1, n // 70))] = -0.0
    return s


def size(rng, hi, lo_big):
    """log-uniform in [1, hi], half of the draws from [lo_big, hi] (the multi-workgroup plans)"""
    lo = lo_big if rng.random() < 0.5 else 1
    return int(np.exp(rng.uniform(np.log(lo), np.log(hi))))


def fuzz_topk(rng):
    n = size(rng, 260000, 6000)
    k = int(rng.choice([1, max(1, n // 97), max(1, n // 8), max(1, n // 4), max(1, n // 4 + 1), max(1, n // 2), n,
                        int(rng.integers(1, n + 1))]))
    B = int(rng.integers(1, 5)) if n > 50000 else int(rng.integers(1, 18))
    kind = str(rng.choice(["tie_free", "ties", "specials"]))
    s = np.stack([scores_of(rng, n, kind) for _ in range(B)])
    vals, idx = ops.topk_desc(T(s, dev), k)
    for b in range(B):
        v, i = orc.topk(s[b], k)
        assert np.array_equal(idx[b].cpu().numpy(), i), f"indices differ (image {b})"
        assert_bits_equal(vals[b].cpu().numpy(), v)
    return f"B={B} n={n} k={k} {kind}"


def boxes_of(rng, n, kind):
    if kind == "realistic":
        return realistic_boxes(rng, n)
    if kind == "lattice":
        return lattice_boxes(n)
    if kind == "heavy":   # everything overlaps: few keeps, every round works
        c = rng.uniform(100, 140, (n, 2)).astype(np.float32)
        wh = rng.uniform(150, 170, (n, 2)).astype(np.float32)
        return np.concatenate([c, c + wh], 1)
    base = realistic_boxes(rng, 40)   # near-duplicates + degenerate boxes
    b = base[rng.integers(0, 40, n)] + rng.normal(0, 2.0, (n, 4)).astype(np.float32)
    b[::50] = np.float32(7.0)
    return b


def fuzz_nms(rng):
    n = size(rng, 30000, 1500)
    B = int(rng.integers(1, 7))
    thr = float(rng.choice([0.3, 0.5, 0.7, 0.9, float(rng.uniform(0.05, 0.95))]))
    P = int(rng.choice([1, 50, 300, 1000, 2000, n]))
    kinds = [str(rng.choice(["realistic", "lattice", "heavy", "clustered"])) for _ in range(B)]
    boxes = np.stack([boxes_of(rng, n, kd) for kd in kinds])
    ns = [int(rng.choice([0, 1, n, n, int(rng.integers(0, n + 1))])) for _ in range(B)]
    bound = int(rng.choice([0, 0, 1, 7, 16, 17, 33, 80, int(rng.integers(1, 500))]))
    lib = _lib.load()
    nbytes = lib.frcnn_nms_workspace_bytes(B, n)
    ws = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device=dev)
    keep = torch.full((B, P), -7, dtype=torch.int32, device=dev)
    n_keep = torch.full((B,), -7, dtype=torch.int32, device=dev)
    used = torch.full((B,), -7, dtype=torch.int32).pin_memory()
    bx = T(boxes, dev)
    cnt = T(np.asarray(ns, np.int32), dev, torch.int32)
    rc = lib.frcnn_nms_sorted_rounds_f32(bx.data_ptr(), cnt.data_ptr(), B, n, thr, P, keep.data_ptr(), n_keep.data_ptr(),
                                         ws.data_ptr(), nbytes, bound, used.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.frcnn_last_error_string()
    torch.cuda.synchronize()
    for b in range(B):
        ref = orc.nms(boxes[b][:ns[b]], -np.arange(ns[b], dtype=np.float32), thr, P)
        assert int(n_keep[b]) == len(ref), f"image {b} ({kinds[b]}, {ns[b]} boxes): {int(n_keep[b])} keeps, oracle {len(ref)}"
        assert np.array_equal(keep[b, :len(ref)].cpu().numpy(), ref), f"image {b} ({kinds[b]}): keep list differs"
    return f"B={B} n={n} thr={thr:.3f} P={P} bound={bound} {kinds}"


def random_boxes(rng, k, Wimg, Himg):
    b = realistic_boxes(rng, k, Wimg, Himg)
    m = max(1, k // 12)
    # small ones (fine levels), some outside / degenerate / huge
    i = rng.integers(0, k, m)
    c = np.stack([rng.uniform(0, Wimg, m), rng.uniform(0, Himg, m)], 1)
    s = rng.uniform(1, 60, (m, 2))
    b[i] = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
    if k > 4:
        b[0] = [-50.0, -40.0, 90.0, 70.0]
        b[1] = [Wimg - 30.0, Himg - 20.0, Wimg + 60.0, Himg + 80.0]
        b[2] = [10.0, 10.0, 5.0, 20.0]
        b[3] = [0.0, 0.0, Wimg, Himg]
    return b


def pyramid(rng):
    if rng.random() < 0.5:   # one level
        stride = int(rng.choice([8, 16, 32]))
        h, w = int(rng.integers(2, 900 // stride + 1)), int(rng.integers(2, 1400 // stride + 1))
        return (h * stride, w * stride), [(h, w)]   # (whole strides: the scale inferred from both axes must agree)
    Himg, Wimg = int(rng.choice([512, 800, 608])), int(rng.choice([640, 1344, 1024]))
    first = int(rng.choice([4, 8]))
    nl = int(rng.integers(2, 5))
    return (Himg, Wimg), [(-(-Himg // (first << l)), -(-Wimg // (first << l))) for l in range(nl)]


def fuzz_roi_align(rng):
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    img, fsz = pyramid(rng)
    B = int(rng.integers(1, 5))
    C = int(rng.choice([1, 3, 7, 8, 16, 20, 33, 64]))
    n = [int(rng.choice([0, 1, 5, 64, 65, 200, int(rng.integers(1, 400))])) for _ in range(B)]
    if sum(n) == 0:
        n[0] = 3
    feats = [rng.standard_normal((B, C, h, w)).astype(np.float32) for h, w in fsz]
    props = [random_boxes(rng, k, img[1], img[0]) if k else np.zeros((0, 4), np.float32) for k in n]
    names = [str(i + 1) for i in range(len(fsz))]
    ref, _ = orc.multiscale_roi_align(feats, props, [img], (7, 7), 2)
    pool = ROIPooling(names, [7, 7], 2)
    d = {nm: T(f, dev) for nm, f in zip(names, feats)}
    P = max(n)
    padded = np.zeros((B, P, 4), np.float32)
    for b in range(B):
        padded[b, :n[b]] = props[b]
    out, total = pool.forward_padded(d, T(padded, dev), T(np.asarray(n, np.int32), dev, torch.int32), [img])
    assert int(total) == sum(n)
    np.testing.assert_allclose(out[:sum(n)].cpu().numpy(), ref, rtol=RTOL, atol=1e-5)
    return f"img={img} levels={fsz} B={B} C={C} n={n}"


def fuzz_roi_align_cl(rng):
    """MultiScaleRoIAlign through csrc/roi_cl.hip (lanes = channels): C a multiple of 64, maps channels-last (read in
    place) or NCHW (copied first), both against the oracle and bit for bit against each other."""
    from objectdetection_amd import ops
    img, fsz = pyramid(rng)
    B = int(rng.integers(1, 4))
    C = int(rng.choice([64, 64, 128, 192]))
    if sum(h * w for h, w in fsz) * C * B > 40e6:   # keeps the oracle's share of a draw small
        C, B = 64, min(B, 2)
    n = [int(rng.choice([0, 1, 5, 64, 65, int(rng.integers(1, 200))])) for _ in range(B)]
    if sum(n) == 0:
        n[0] = 3
    feats = [rng.standard_normal((B, C, h, w)).astype(np.float32) for h, w in fsz]
    props = [random_boxes(rng, k, img[1], img[0]) if k else np.zeros((0, 4), np.float32) for k in n]
    names = [str(i + 1) for i in range(len(fsz))]
    ref, _ = orc.multiscale_roi_align(feats, props, [img], (7, 7), 2)
    pool = ops.MultiScaleRoIAlign(names, (7, 7), 2)
    fd, scales, k_min, k_max = pool._levels({nm: T(f, dev) for nm, f in zip(names, feats)}, [img])
    cfg = ((7, 7), 2, tuple(float(s_) for s_ in scales), int(k_min), int(k_max))
    P = max(n)
    padded = np.zeros((B, P, 4), np.float32)
    for b in range(B):
        padded[b, :n[b]] = props[b]
    pd, cd = T(padded, dev), T(np.asarray(n, np.int32), dev, torch.int32)
    cl_first = bool(rng.random() < 0.5)
    outs = []
    for cl_maps in (cl_first, not cl_first):
        maps = [f.contiguous(memory_format=torch.channels_last) for f in fd] if cl_maps else fd
        out, total = ops._roi_align_levels(cfg, pd, cd, maps, path="cl")
        assert int(total) == sum(n)
        np.testing.assert_allclose(out[:sum(n)].cpu().numpy(), ref, rtol=RTOL, atol=1e-5)
        outs.append(out[:sum(n)].clone())
    assert torch.equal(outs[0], outs[1])
    return f"img={img} levels={fsz} B={B} C={C} n={n}"


def fuzz_roi_backward(rng):
    img, fsz = pyramid(rng)
    B = int(rng.integers(1, 4))
    C = int(rng.choice([1, 2, 3, 7, 8, 20, 33]))
    n = [int(rng.choice([1, 5, 64, 65, 130, int(rng.integers(1, 300))])) for _ in range(B)]
    names = [str(i) for i in range(len(fsz))]
    feats = [torch.randn(B, C, h, w, device=dev, requires_grad=True) for h, w in fsz]
    boxes = [random_boxes(rng, k, img[1], img[0]) for k in n]
    pool = ops.MultiScaleRoIAlign(names, (7, 7), 2)
    out = pool(dict(zip(names, feats)), [T(b, dev) for b in boxes], [img])
    G = torch.randn(out.shape, device=dev)
    out.backward(G)
    ref = orc.multiscale_roi_align_backward(G.cpu().numpy(), [tuple(f.shape) for f in feats], boxes, [img])
    for f, r in zip(feats, ref):
        np.testing.assert_allclose(f.grad.cpu().numpy(), r, rtol=RTOL, atol=1e-5 * max(1.0, float(np.abs(r).max())))
    return f"img={img} levels={fsz} B={B} C={C} n={n}"


def fuzz_roi_pool(rng):
    img, fsz = pyramid(rng)
    h, w = fsz[-1] if len(fsz) == 1 else fsz[int(rng.integers(0, len(fsz)))]
    B = int(rng.integers(1, 4))
    C = int(rng.choice([1, 5, 16, 20, 48]))
    n = [int(rng.choice([0, 1, 64, 65, int(rng.integers(1, 300))])) for _ in range(B)]
    if sum(n) == 0:
        n[0] = 2
    feat = rng.standard_normal((B, C, h, w)).astype(np.float32)
    feat[:, :, ::3, ::2] = np.round(feat[:, :, ::3, ::2])   # ties for the first-max rule
    props = [random_boxes(rng, k, img[1], img[0]) if k else np.zeros((0, 4), np.float32) for k in n]
    scale = orc.infer_scale((h, w), img)
    rois = np.concatenate([np.concatenate([np.full((len(p), 1), b, np.float32), p], 1) for b, p in enumerate(props)])
    ref, ref_arg = orc.roi_pool(feat, rois, (7, 7), scale)
    P = max(n)
    padded = np.zeros((B, P, 4), np.float32)
    for b in range(B):
        padded[b, :n[b]] = props[b]
    try:
        out, arg, total = ops.roi_pool_batched(T(feat, dev), T(padded, dev), T(np.asarray(n, np.int32), dev, torch.int32),
                                               (7, 7), scale)
        assert int(total) == sum(n)
    except _lib.FrcnnError as e:   # planes beyond LDS are the roi-list entry point's (documented)
        assert "does not fit LDS" in str(e), str(e)
        out, arg = ops.roi_pool(T(feat, dev), T(rois, dev), (7, 7), scale)
    assert np.array_equal(arg[:sum(n)].cpu().numpy(), ref_arg), "argmax differs"
    assert_bits_equal(out[:sum(n)].cpu().numpy(), ref)
    return f"img={img} map={(h, w)} B={B} C={C} n={n}"


def fuzz_rpn(rng):
    """RPN.forward (eval): anchors -> top-k -> decode / clip / filter -> NMS -> gather, in place on the head's NCHW
    outputs where the shape allows (one C call), staged otherwise.  As tests/test_gpu_parity.py's
    test_rpn_forward_vs_oracle: the oracle's NMS is fed the HIP path's decoded boxes (exp ULPs cannot flip a keep)."""
    from objectdetection_amd.RPN import RPN, AnchorGenerator
    B = int(rng.integers(1, 5))
    if rng.random() < 0.5:
        stride = int(rng.choice([16, 32]))
        h, w = int(rng.integers(4, 40)), int(rng.integers(4, 64))
        info, fsz = (B, h * stride, w * stride), [(h, w)]
    else:
        base = int(rng.choice([256, 512]))
        nl = int(rng.integers(2, 5))
        first = int(rng.choice([4, 8]))
        info = (B, base, base)
        fsz = [(base // (first << l), base // (first << l)) for l in range(nl)][::-1]
    A = 9 * sum(h * w for h, w in fsz)
    pre = int(rng.choice([max(1, A // 7), max(1, A // 2), A, A + 5, int(rng.integers(1, A + 1))]))
    post = int(rng.choice([1, 100, 1000, 2000]))
    thr = float(rng.choice([0.5, 0.7, 0.9]))
    min_size = float(rng.choice([1e-3, 2.0, 30.0]))
    list_cls, list_regs = synth_head_outputs(rng, B, fsz)
    params = dict(anchor_generator=AnchorGenerator(), rpn_head=_FixedHead([T(c, dev) for c in list_cls],
                                                                         [T(r, dev) for r in list_regs]),
                  num_pre_nms=pre, num_post_nms=post, nms_threshold=thr, proposal_min_size=min_size,
                  pos_threshold=0.7, neg_threshold=0.3, num_sample=256, pos_fraction=0.5)
    rpn = RPN(params).eval()
    feats = [torch.zeros(B, 1, h, w, device=dev) for h, w in fsz]
    for _ in range(2):   # the second call runs with the first one's NMS round bound
        props, loss = rpn(info, feats, None)
    anchors = torch.cat([a[0].reshape(-1, 4) for a in params['anchor_generator'](info, feats)], 0)
    scores, deltas = ops.rpn_flatten(params['rpn_head'].c, params['rpn_head'].r)
    k = min(pre, scores.shape[1])
    top_idx = np.stack([orc.topk(scores[b].cpu().numpy(), k)[1] for b in range(B)])
    top_val = np.stack([orc.topk(scores[b].cpu().numpy(), k)[0] for b in range(B)])
    boxes, fscores, src, count = ops.decode_clip_filter(anchors, deltas, T(top_idx, dev, torch.int32), T(top_val, dev),
                                                        info[1], info[2], min_size)
    for b in range(B):
        c = int(count[b])
        hip_boxes = boxes[b, :c].cpu().numpy()
        keep_ref = orc.nms(hip_boxes, fscores[b, :c].cpu().numpy(), thr)[:post]
        assert props[b].shape == (len(keep_ref), 4), f"image {b}: {props[b].shape[0]} proposals, oracle {len(keep_ref)}"
        assert_bits_equal(props[b].cpu().numpy(), hip_boxes[keep_ref])
    return f"info={info} levels={fsz} pre={pre} post={post} thr={thr} min={min_size}"


def fuzz_targets(rng):
    """IoU matrix + threshold labels + argmax + delta encodings (utils.box_iou / tagging_box / box_deviation fused):
    shared or per-image box_1, ragged box counts, duplicates and exact hits, both threshold pairs of the reference."""
    B = int(rng.integers(1, 6))
    A = size(rng, 30000, 2000)
    M = int(rng.choice([1, 2, 37, 100, int(rng.integers(1, 130))]))
    ms = [int(rng.choice([1, M, int(rng.integers(1, M + 1))])) for _ in range(B)]
    per_image = rng.random() < 0.5
    box1 = np.stack([realistic_boxes(rng, A) for _ in range(B)]) if per_image else realistic_boxes(rng, A)
    w = rng.uniform(8, 300, (B, M)); h = rng.uniform(8, 300, (B, M))
    x1 = rng.uniform(0, 1000 - w); y1 = rng.uniform(0, 600 - h)
    gt = np.stack([x1, y1, x1 + w, y1 + h], 2).astype(np.float32)
    if M > 1:
        gt[:, 1] = gt[:, 0]   # a duplicate box: first-max argmax
    for b in range(B):        # exact hits and near hits: IoU = 1 and around the thresholds
        src = box1[b] if per_image else box1
        j = rng.integers(0, A, min(A, 3 * ms[b]))
        src[j] = gt[b, rng.integers(0, ms[b], len(j))] + (rng.normal(0, 4, (len(j), 4)) * (rng.random((len(j), 1)) < 0.7)).astype(np.float32)
    lab = rng.integers(1, 21, (B, M)).astype(np.float32)
    neg, pos = (0.3, 0.7) if rng.random() < 0.5 else (0.5, 0.5)
    out = ops.iou_label_encode(T(box1, dev), T(gt, dev), T(np.asarray(ms, np.int32), dev, torch.int32), neg, pos,
                               T(lab, dev), [10.0, 10.0, 5.0, 5.0])
    for b in range(B):
        l, a, mx, dv = orc.iou_label_encode(box1[b] if per_image else box1, gt[b, :ms[b]], neg, pos, lab[b, :ms[b]],
                                            [10.0, 10.0, 5.0, 5.0])
        assert_bits_equal(out["label"][b].cpu().numpy(), l)
        assert np.array_equal(out["argmax"][b].cpu().numpy(), a), f"image {b}: argmax differs"
        assert_bits_equal(out["max_iou"][b].cpu().numpy(), mx)
        np.testing.assert_allclose(out["dev"][b].cpu().numpy(), dv, rtol=RTOL, atol=1e-6)
    return f"B={B} A={A} M={M} ms={ms} per_image={per_image} thr=({neg},{pos})"


OPS = {"targets": fuzz_targets, "rpn": fuzz_rpn, "topk": fuzz_topk, "nms": fuzz_nms, "roi_align": fuzz_roi_align, "roi_align_cl": fuzz_roi_align_cl, "roi_backward": fuzz_roi_backward,
       "roi_pool": fuzz_roi_pool}


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = os.environ.get("FUZZ_ONLY")
    case = os.environ.get("FUZZ_CASE")
    orc.build()
    names = [only] if only else list(OPS)
    runs = {n: 0 for n in names}
    fails = []
    t0 = time.time()
    i = 0
    while time.time() - t0 < budget:
        name = names[i % len(names)]
        seed = int(case) if case else seed0 * 100003 + i
        rng = np.random.default_rng(seed)
        try:
            desc = OPS[name](rng)
            runs[name] += 1
            if case:
                print("ok", name, seed, desc)
        except AssertionError as e:
            fails.append((name, seed))
            print(f"FAIL {name} seed {seed}: {str(e).strip().splitlines()[0][:300] if str(e).strip() else 'assert'}", flush=True)
            traceback.print_exc(limit=2)
        i += 1
        if case:
            break
        if i % 50 == 0:
            print(f"[{time.time() - t0:5.0f} s] {i} draws, {len(fails)} failures", flush=True)
    print("fuzz summary:", runs, "failures:", fails)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
