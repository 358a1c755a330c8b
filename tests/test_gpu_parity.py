"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden
vectors.  Bit-exact for anchors, IoU, labels, argmax, top-k order, filter indices, NMS keep lists,
level indices and RoI-pool argmax; 1e-5 relative for exp/log/bilinear outputs."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda:0")


def T(a, dev, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(dev)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.int32)


def assert_bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape
    nan = np.isnan(a) & np.isnan(b)
    assert np.array_equal(bits(a)[~nan], bits(b)[~nan])


def tie_free_scores(rng, n):
    return ((rng.permutation(n).astype(np.float32) / np.float32(n)) - np.float32(0.5)) * np.float32(8)


def realistic_boxes(rng, n, W=1000.0, H=600.0):
    cx = rng.uniform(0, W, n)
    cy = rng.uniform(0, H, n)
    w = 16 * 2 ** rng.uniform(0, 5, n)
    h = 16 * 2 ** rng.uniform(0, 5, n)
    b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    b[:, 0::2] = b[:, 0::2].clip(0, W)
    b[:, 1::2] = b[:, 1::2].clip(0, H)
    return b.astype(np.float32)


# ------------------------------------------------------------------------------------ anchors
@pytest.mark.parametrize("name", ["smoke", "cfg1", "fpn", "small_full", "custom"])
def test_anchor_generator_golden(dev, name):
    from objectdetection_amd.RPN.anchor_generator import AnchorGenerator
    g = load_golden("anchors.npz")
    info = tuple(int(v) for v in g[f"{name}.info"])
    fsz = [tuple(int(v) for v in r) for r in g[f"{name}.fsz"]]
    if f"{name}.sizes" in g:
        ag = AnchorGenerator(tuple(tuple(float(v) for v in r) for r in g[f"{name}.sizes"]),
                             tuple(tuple(float(v) for v in r) for r in g[f"{name}.ratios"]))
    else:
        ag = AnchorGenerator()
    feats = [torch.zeros(info[0], 1, h, w, device=dev) for h, w in fsz]
    res = ag(info, feats)
    assert_bits_equal(ag.base_anchor.cpu().numpy(), g[f"{name}.base"])
    for l, a in enumerate(res):
        assert tuple(a.shape) == tuple(g[f"{name}.l{l}.shape"])
        assert a.stride(0) == 0 or info[0] == 1
        a0 = a[0].cpu().numpy()
        assert int(bits(a0).astype(np.int64).sum()) == int(g[f"{name}.l{l}.checksum"])
        if f"{name}.l{l}.full" in g:
            assert_bits_equal(a0, g[f"{name}.l{l}.full"])
        else:
            assert_bits_equal(a0[:32], g[f"{name}.l{l}.head"])
            assert_bits_equal(a0[-32:], g[f"{name}.l{l}.tail"])
    for b in range(1, info[0]):
        assert torch.equal(res[0][b], res[0][0])


# ---------------------------------------------------------------------------- targets (K5)
@pytest.mark.parametrize("name", ["m1", "m2", "m100", "m100w", "m7swap", "nan"])
def test_targets_golden(dev, name):
    from objectdetection_amd import utils as U
    g = load_golden("targets.npz")
    box1, box2 = T(g[f"{name}.box1"], dev), T(g[f"{name}.box2"], dev)
    neg, pos = (float(v) for v in g[f"{name}.thr"])
    labels = T(g[f"{name}.labels"], dev) if f"{name}.labels" in g else None
    weights = [float(v) for v in g[f"{name}.weights"]] if f"{name}.weights" in g else None
    if f"{name}.iou" in g:
        assert_bits_equal(U.box_iou(box1, box2).cpu().numpy(), g[f"{name}.iou"])
    lab = U.tagging_box(box1, box2, neg, pos, labels)
    assert lab.dtype == torch.float32 and lab.shape == (box1.shape[0],)
    assert_bits_equal(lab.cpu().numpy(), g[f"{name}.label"])
    from objectdetection_amd import ops
    out = ops.iou_label_encode(box1, box2[None], None, neg, pos, labels[None] if labels is not None else None, weights)
    assert_bits_equal(out["max_iou"][0].cpu().numpy(), g[f"{name}.max"])
    assert np.array_equal(out["argmax"][0].cpu().numpy(), g[f"{name}.argmax"])
    if f"{name}.dev" in g:
        dev_out = U.box_deviation(box1, box2, weights).cpu().numpy()
        ref = g[f"{name}.dev"]
        fin = np.isfinite(ref)
        assert np.array_equal(np.isnan(dev_out), np.isnan(ref))
        assert np.array_equal(dev_out[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)])
        np.testing.assert_allclose(dev_out[fin], ref[fin], rtol=RTOL, atol=1e-6)


def test_targets_batched_vs_oracle(dev, oracle):
    """Config-3 shape: shared anchors (A=20,646), B=3 images with ragged GT counts."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(3)
    anchors = oracle.anchors_all_levels((1, 600, 1000), [(37, 62)])
    ms = [100, 1, 37]
    gts = []
    for m in ms:
        w = rng.uniform(16, 256, m); h = rng.uniform(16, 256, m)
        x1 = rng.uniform(0, 1000 - w); y1 = rng.uniform(0, 600 - h)
        gts.append(np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32))
    M = max(ms)
    gt = np.zeros((3, M, 4), np.float32)
    lab = np.zeros((3, M), np.float32)
    for i, g_ in enumerate(gts):
        gt[i, :ms[i]] = g_
        lab[i, :ms[i]] = rng.integers(1, 21, ms[i])
    out = ops.iou_label_encode(T(anchors, dev), T(gt, dev), T(ms, dev, torch.int32), 0.3, 0.7, T(lab, dev),
                               [10.0, 10.0, 5.0, 5.0])
    for i in range(3):
        l, a, mx, dv = oracle.iou_label_encode(anchors, gts[i], 0.3, 0.7, lab[i, :ms[i]], [10.0, 10.0, 5.0, 5.0])
        assert_bits_equal(out["label"][i].cpu().numpy(), l)
        assert np.array_equal(out["argmax"][i].cpu().numpy(), a)
        assert_bits_equal(out["max_iou"][i].cpu().numpy(), mx)
        np.testing.assert_allclose(out["dev"][i].cpu().numpy(), dv, rtol=RTOL, atol=1e-6)
    assert (out["label"] > 0).sum() > 0


def test_proposal_targets_config3_shape_vs_oracle(dev, oracle):
    """BASELINE configs[2], the ProposalTarget half at its own shape: per-image box_1 (B, 2,000 proposals, 4) against
    100 boxes per image with class labels, thresholds 0.5 / 0.5, weights [10,10,5,5], batch 16 in one launch."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(33)
    B, P, M = 16, 2000, 100
    props = np.stack([realistic_boxes(rng, P) for _ in range(B)])
    w = rng.uniform(16, 256, (B, M)); h = rng.uniform(16, 256, (B, M))
    x1 = rng.uniform(0, 1000 - w); y1 = rng.uniform(0, 600 - h)
    gt = np.stack([x1, y1, x1 + w, y1 + h], 2).astype(np.float32)
    props[:, :M] = gt + rng.normal(0, 3, gt.shape).astype(np.float32)      # proposals near the boxes: positives exist
    lab = rng.integers(1, 21, (B, M)).astype(np.float32)
    out = ops.iou_label_encode(T(props, dev), T(gt, dev), None, 0.5, 0.5, T(lab, dev), [10.0, 10.0, 5.0, 5.0])
    n_pos = 0
    for i in range(B):
        l, a, mx, dv = oracle.iou_label_encode(props[i], gt[i], 0.5, 0.5, lab[i], [10.0, 10.0, 5.0, 5.0])
        assert_bits_equal(out["label"][i].cpu().numpy(), l)
        assert np.array_equal(out["argmax"][i].cpu().numpy(), a)
        assert_bits_equal(out["max_iou"][i].cpu().numpy(), mx)
        np.testing.assert_allclose(out["dev"][i].cpu().numpy(), dv, rtol=RTOL, atol=1e-6)
        n_pos += int((l > 0).sum())
    assert n_pos > B * 50


# ------------------------------------------------------------------- decode / clip / filter
@pytest.mark.parametrize("K", [1, 3])
def test_box_regression_golden(dev, K):
    from objectdetection_amd import utils as U
    g = load_golden("decode.npz")
    out = U.box_regression(T(g[f"k{K}.box"], dev), T(g[f"k{K}.regs"], dev)).cpu().numpy()
    np.testing.assert_allclose(out, g[f"k{K}.decoded"], rtol=RTOL, atol=1e-4)


@pytest.mark.parametrize("K", [1, 3])
def test_box_regression_backward_golden(dev, oracle, K):
    """decode is differentiable like the reference's; gradients match what the
    reference's own autograd produced (decode_grad.npz) and the CPU restatement."""
    from objectdetection_amd import utils as U
    g = load_golden("decode_grad.npz")
    box = T(g[f"k{K}.box"], dev).requires_grad_(True)
    regs = T(g[f"k{K}.regs"], dev).requires_grad_(True)
    out = U.box_regression(box, regs)
    assert out.requires_grad
    out.backward(T(g[f"k{K}.upstream"], dev))
    np.testing.assert_allclose(regs.grad.cpu().numpy(), g[f"k{K}.grad_regs"], rtol=RTOL, atol=1e-3)
    np.testing.assert_allclose(box.grad.cpu().numpy(), g[f"k{K}.grad_box"], rtol=RTOL, atol=1e-3)
    ref_regs, ref_box = oracle.box_regression_backward(g[f"k{K}.box"], g[f"k{K}.regs"], g[f"k{K}.upstream"])
    np.testing.assert_allclose(regs.grad.cpu().numpy(), ref_regs, rtol=RTOL, atol=1e-4)
    np.testing.assert_allclose(box.grad.cpu().numpy(), ref_box, rtol=RTOL, atol=1e-4)
    # only the deltas need a gradient (the RPN case: anchors are constants); no-grad inputs stay on the plain path
    regs2 = T(g[f"k{K}.regs"], dev).requires_grad_(True)
    U.box_regression(T(g[f"k{K}.box"], dev), regs2).backward(T(g[f"k{K}.upstream"], dev))
    assert torch.equal(regs2.grad, regs.grad)
    assert not U.box_regression(T(g[f"k{K}.box"], dev), T(g[f"k{K}.regs"], dev)).requires_grad


def test_clip_small_golden(dev):
    from objectdetection_amd import utils as U
    g = load_golden("decode.npz")
    h, w = (int(v) for v in g["clip.image_size"])
    clipped = U.box_clip(T(g["clip.in"], dev), (h, w))
    assert_bits_equal(clipped.cpu().numpy(), g["clip.out"])
    for t in ("0.001", "16.0"):
        kept, idx = U.remove_small_box(clipped, float(t))
        assert idx.dtype == torch.int64
        assert np.array_equal(idx.cpu().numpy(), g[f"small.{t}.index"])
        assert_bits_equal(kept.cpu().numpy(), g[f"small.{t}.box"])
    # score prefilter form (utils.py:227-229): indices are relative to the score-filtered rows
    score = torch.linspace(0, 1, clipped.shape[0], device=dev)
    kept2, idx2 = U.remove_small_box(clipped, 16.0, score, 0.5)
    sel = clipped[score > 0.5]
    ref_idx = torch.where(((sel[:, 2] - sel[:, 0]) > 16.0) & ((sel[:, 3] - sel[:, 1]) > 16.0))[0]
    assert torch.equal(idx2, ref_idx) and torch.equal(kept2, sel[ref_idx])


def test_empty_inputs(dev):
    from objectdetection_amd import utils as U, ops
    e = torch.zeros((0, 4), device=dev)
    assert U.box_clip(e, (10, 10)).shape == (0, 4)
    assert U.box_regression(e, torch.zeros((0, 4), device=dev)).shape == (0, 4)
    kept, idx = U.remove_small_box(e, 1e-3)
    assert kept.shape == (0, 4) and idx.shape == (0,)
    assert ops.nms(e, torch.zeros((0,), device=dev), 0.5).shape == (0,)
    all_small = torch.tensor([[1.0, 1.0, 1.0, 5.0], [2.0, 2.0, 3.0, 2.0]], device=dev)
    kept, idx = U.remove_small_box(all_small, 1e-3)
    assert kept.shape == (0, 4)
    with pytest.raises(ValueError):
        U.tagging_box(torch.ones((3, 4), device=dev), e, 0.3, 0.7)
    with pytest.raises(Exception):
        U.box_iou(torch.ones(3, 4), torch.ones(2, 4))  # CPU tensors: no fallback


# --------------------------------------------------------------------------------- top-k (K2)
@pytest.mark.parametrize("B,n,k", [(1, 100, 100), (2, 5000, 1000), (3, 20646, 12000), (1, 70000, 30000), (2, 777, 1),
                                   # edges of the full-sort path (k >= n/2, 8192 < n <= 21504) and just outside it
                                   (2, 21504, 21504), (1, 8193, 4097), (1, 21505, 12000), (1, 8192, 8192),
                                   (2, 20646, 10322)])
def test_topk_vs_oracle(dev, oracle, B, n, k):
    from objectdetection_amd import ops
    rng = np.random.default_rng(n + k)
    s = np.stack([tie_free_scores(rng, n) for _ in range(B)])
    vals, idx = ops.topk_desc(T(s, dev), k)
    for b in range(B):
        v, i = oracle.topk(s[b], k)
        assert np.array_equal(idx[b].cpu().numpy(), i)
        assert_bits_equal(vals[b].cpu().numpy(), v)


def test_topk_ties_and_specials(dev, oracle):
    """Heavy ties, +-0, +-inf and NaN: the build's rule is descending key, then ascending index."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(9)
    n = 9000
    s = rng.integers(-3, 4, n).astype(np.float32) * 0.5
    s[::97] = np.nan
    s[5::101] = np.inf
    s[7::103] = -np.inf
    s[11::107] = -0.0
    for k in (1, 64, 4500, n):
        vals, idx = ops.topk_desc(T(s[None], dev), k)
        v, i = oracle.topk(s, k)
        assert np.array_equal(idx[0].cpu().numpy(), i)
        assert_bits_equal(vals[0].cpu().numpy(), v)
    # the same on the full-sort path at its largest size, two images with different contents
    n = 21504
    s2 = rng.integers(-40, 41, (2, n)).astype(np.float32) * 0.25
    s2[0, ::89] = np.nan
    s2[1, 3::97] = -0.0
    s2[1, 4::97] = 0.0
    s2[0, 6::131] = np.inf
    for k in (n // 2, 15000, n):
        vals, idx = ops.topk_desc(T(s2, dev), k)
        for b in range(2):
            v, i = oracle.topk(s2[b], k)
            assert np.array_equal(idx[b].cpu().numpy(), i)
            assert_bits_equal(vals[b].cpu().numpy(), v)


@pytest.mark.parametrize("kind", ["tie_free", "all_equal", "two_values", "ascending", "descending", "specials"])
@pytest.mark.parametrize("B,n,k", [(1, 8193, 100), (2, 20646, 12000), (1, 50000, 50000), (3, 50000, 300),
                                   (1, 200907, 30000), (16, 9000, 4500), (1, 805707, 30000), (2, 1300000, 5000)])
def test_topk_split_matches_single_workgroup(dev, oracle, kind, B, n, k):
    """The multi-workgroup path (slice sorts + windowed merge) against one workgroup per image and the
    oracle, whatever the split: adversarial key distributions make the merge windows empty (tie-free,
    sorted inputs), huge (all keys equal: the windows do not fit LDS and are searched in global memory)
    or lopsided."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(n * 7 + k)
    if kind == "tie_free":
        s = np.stack([tie_free_scores(rng, n) for _ in range(B)])
    elif kind == "all_equal":
        s = np.full((B, n), 0.375, np.float32)
    elif kind == "two_values":
        s = rng.integers(0, 2, (B, n)).astype(np.float32)
    elif kind == "ascending":
        s = np.tile(np.arange(n, dtype=np.float32) * 0.5, (B, 1))
    elif kind == "descending":
        s = np.tile(-np.arange(n, dtype=np.float32), (B, 1))
    else:
        s = rng.integers(-30, 31, (B, n)).astype(np.float32) * 0.125
        s[:, ::89] = np.nan
        s[:, 3::97] = -0.0
        s[:, 6::131] = np.inf
        s[:, 9::137] = -np.inf
    dev_s = T(s, dev)
    ref_v, ref_i = ops.topk_desc(dev_s, k, max_slices=1)
    v0, i0 = oracle.topk(s[0], k)
    assert np.array_equal(ref_i[0].cpu().numpy(), i0)
    assert_bits_equal(ref_v[0].cpu().numpy(), v0)
    for max_slices in (0, 2, 5, 64):
        v, i = ops.topk_desc(dev_s, k, max_slices=max_slices)
        assert torch.equal(i, ref_i), (kind, max_slices)
        assert_bits_equal(v.cpu().numpy(), ref_v.cpu().numpy())
    _, i = ops.topk_desc(dev_s, k, want_values=False)
    assert torch.equal(i, ref_i)


@pytest.mark.parametrize("fsz,B,k,P", [([(37, 62)], 3, 12000, 2000), ([(13, 21), (25, 42), (50, 84), (100, 168)], 2, 30000, 2000),
                                        ([(40, 50)], 1, 18000, 300)])
def test_rpn_proposals_one_call_matches_stages(dev, fsz, B, k, P):
    """frcnn_rpn_proposals_nchw_f32 (RPN/rpn.py:54-85 in one library call, intermediates in one workspace) against the
    four stage entry points: same proposals, same counts, bit for bit -- with a fresh workspace full of 0xFF, a
    reused one, and with the NMS rounds bounded by an NmsRounds."""
    from objectdetection_amd import ops
    from objectdetection_amd.RPN.anchor_generator import AnchorGenerator
    rng = np.random.default_rng(len(fsz) * 100 + B)
    na, Himg, Wimg = 9, 600.0 if len(fsz) == 1 else 800.0, 1000.0 if len(fsz) == 1 else 1344.0
    A = sum(h * w for h, w in fsz) * na
    sc = np.stack([tie_free_scores(rng, A) for _ in range(B)])
    cls, reg, off = [], [], 0
    for h, w in fsz:
        n = h * w * na
        cls.append(T(sc[:, off:off + n].reshape(B, h, w, na).transpose(0, 3, 1, 2).copy(), dev))
        reg.append(T((rng.standard_normal((B, 4 * na, h, w)) * 0.3).astype(np.float32), dev))
        off += n
    anchors = torch.cat([a[0].reshape(-1, 4) for a in AnchorGenerator()((B, int(Himg), int(Wimg)), cls)], 0)
    tv, ti = ops.topk_desc_nchw(cls, k)
    boxes, _, _, count = ops.decode_clip_filter_nchw(anchors, reg, ti, tv, Himg, Wimg, 1e-3)
    keep, nk = ops.nms_sorted(boxes, count, 0.7, P)
    ref = ops.gather_boxes(boxes, keep, nk)
    lib_bytes = ops._lib.load().frcnn_rpn_proposals_workspace_bytes(B, A, k, P)
    ws = torch.full((lib_bytes,), 0xFF, dtype=torch.uint8, device=dev)
    rounds = ops.NmsRounds()
    for _ in range(3):
        props, n_keep, ws2 = ops.rpn_proposals_nchw(cls, reg, anchors, k, Himg, Wimg, 1e-3, 0.7, P, rounds=rounds,
                                                    workspace=ws)
        torch.cuda.synchronize()
        assert ws2 is ws
        assert torch.equal(n_keep, nk)
        assert torch.equal(props, ref)


def test_rpn_stages_golden(dev):
    """RPN/rpn.py:55-83 on the HIP path vs what the reference's own functions produced."""
    from objectdetection_amd import ops
    from objectdetection_amd.RPN.anchor_generator import AnchorGenerator
    g = load_golden("rpn_stages.npz")
    info = tuple(int(v) for v in g["info"])
    fsz = [tuple(int(v) for v in r) for r in g["fsz"]]
    list_cls = [T(g[f"cls{l}"], dev) for l in range(len(fsz))]
    list_regs = [T(g[f"reg{l}"], dev) for l in range(len(fsz))]
    ag = AnchorGenerator()
    anchors = torch.cat([a[0].reshape(-1, 4) for a in ag(info, list_cls)], 0)
    scores, deltas = ops.rpn_flatten(list_cls, list_regs)
    ref_scores = torch.cat([c.permute(0, 2, 3, 1).flatten(1, 3) for c in list_cls], dim=1)
    ref_deltas = torch.cat([r.permute(0, 2, 3, 1).flatten(1, 2) for r in list_regs], dim=1).reshape(info[0], -1, 4)
    assert torch.equal(scores, ref_scores) and torch.equal(deltas, ref_deltas)
    k = int(g["num_pre_nms"])
    top_val, top_idx = ops.topk_desc(scores, k)
    boxes, fscores, src, count = ops.decode_clip_filter(anchors, deltas, top_idx, top_val, info[1], info[2],
                                                        float(g["min_size"]))
    for b in range(info[0]):
        assert np.array_equal(top_idx[b].cpu().numpy(), g[f"b{b}.top_index"])
        assert_bits_equal(top_val[b].cpu().numpy(), g[f"b{b}.top_score"])
        c = int(count[b])
        assert np.array_equal(src[b, :c].cpu().numpy(), g[f"b{b}.small_index"])
        np.testing.assert_allclose(boxes[b, :c].cpu().numpy(), g[f"b{b}.filtered"], rtol=RTOL, atol=1e-4)
        assert_bits_equal(fscores[b, :c].cpu().numpy(), g[f"b{b}.filtered_score"])


@pytest.mark.parametrize("B,A,k,min_size", [(3, 20646, 12000, 40.0), (2, 20646, 12000, 1e-3), (2, 5000, 1024, 30.0),
                                            (1, 5000, 1025, 30.0), (2, 3000, 700, 25.0), (1, 70000, 30000, 60.0),
                                            (2, 9000, 4096, 1e9)])
def test_decode_clip_filter_gaps(dev, B, A, k, min_size):
    """Fused decode / clip / small-box filter with many rows dropped in every tile (the gap-closing pass),
    with none dropped, with all dropped, and at the tile boundaries: against the same stages done one by
    one (box_regression, box_clip, remove_small_box -- each pinned by the reference's goldens)."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(A + k)
    anchors = T(realistic_boxes(rng, A), dev)
    deltas = T((rng.standard_normal((B, A, 4)) * 0.4).astype(np.float32), dev)
    top_idx = T(np.stack([rng.permutation(A)[:k] for _ in range(B)]).astype(np.int32), dev, torch.int32)
    top_score = T(rng.standard_normal((B, k)).astype(np.float32), dev)
    boxes, scores, src, count = ops.decode_clip_filter(anchors, deltas, top_idx, top_score, 600.0, 1000.0, min_size)
    cnt = count.cpu().numpy()
    for b in range(B):
        sel = top_idx[b].long()
        dec = ops.box_clip(ops.box_regression(anchors[sel], deltas[b][sel]), 600.0, 1000.0)
        kept, index, n = ops.remove_small_box(dec, min_size)
        n = int(n)
        assert int(cnt[b]) == n
        assert torch.equal(src[b, :n].long(), index[:n].long())
        assert_bits_equal(boxes[b, :n].cpu().numpy(), kept[:n].cpu().numpy())
        assert_bits_equal(scores[b, :n].cpu().numpy(), top_score[b][index[:n].long()].cpu().numpy())
    if min_size > 1e8:
        assert (cnt == 0).all()
    elif min_size > 1:
        assert ((cnt > 0) & (cnt < k)).all()


@pytest.mark.parametrize("levels,na,B,k,per_image_anchors,min_size", [
    ([(37, 62)], 9, 4, 12000, False, 1e-3),                                      # the headline shape
    ([(37, 62)], 9, 2, 20646, False, 2.0),                                       # k = A
    ([(50, 76), (25, 38), (13, 19), (7, 10)], 3, 3, 9000, True, 2.0),            # A = 15201 over four levels
    ([(64, 64), (32, 32), (16, 16), (8, 8), (4, 4), (2, 2)], 2, 2, 6000, False, 2.0),   # six levels
    ([(100, 168), (50, 84), (25, 42), (13, 21)], 9, 2, 30000, False, 1e-3),      # config 5: A = 200,907
])
def test_rpn_in_place_matches_flattened(dev, levels, na, B, k, per_image_anchors, min_size):
    """Top-k and decode reading the head's NCHW outputs in place return the same bits as rpn_flatten +
    topk_desc + decode_clip_filter: ties, +-inf and filtered rows included."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(k + na)
    list_cls, list_regs = [], []
    for (h, w) in levels:
        c = rng.standard_normal((B, na, h, w)).astype(np.float32)
        c = np.round(c * 64) / 64            # many exact ties: order must fall back to the flat index
        c[:, :, ::5, ::7] = np.float32(0.25)
        c[0, 0, 0, 0] = np.inf
        c[-1, -1, -1, -1] = -np.inf
        list_cls.append(T(c, dev))
        list_regs.append(T((rng.standard_normal((B, 4 * na, h, w)) * 0.5).astype(np.float32), dev))
    A = na * sum(h * w for h, w in levels)
    assert ops.rpn_in_place_supported(list_cls, k)
    assert not ops.rpn_in_place_supported(list_cls, k, max_slices=1) or A > 21504
    anc = realistic_boxes(rng, A if not per_image_anchors else B * A)
    anc[::9, 2:] = anc[::9, :2] + 0.5        # tiny anchors: some decoded boxes fall under min_size
    anchors = T(anc.reshape(B, A, 4) if per_image_anchors else anc, dev)
    scores, deltas = ops.rpn_flatten(list_cls, list_regs)
    tv, ti = ops.topk_desc(scores, k)
    b0, s0, r0, c0 = ops.decode_clip_filter(anchors, deltas, ti, tv, 600.0, 1000.0, min_size)
    tv1, ti1 = ops.topk_desc_nchw(list_cls, k)
    b1, s1, r1, c1 = ops.decode_clip_filter_nchw(anchors, list_regs, ti1, tv1, 600.0, 1000.0, min_size)
    assert torch.equal(ti, ti1)
    assert_bits_equal(tv.cpu().numpy(), tv1.cpu().numpy())
    assert torch.equal(c0, c1)
    cnt = c0.cpu().numpy()
    assert (cnt > 0).all()
    for b in range(B):
        n = int(cnt[b])
        assert torch.equal(r0[b, :n], r1[b, :n])
        assert_bits_equal(b0[b, :n].cpu().numpy(), b1[b, :n].cpu().numpy())
        assert_bits_equal(s0[b, :n].cpu().numpy(), s1[b, :n].cpu().numpy())


def test_rpn_in_place_rejects_small_images(dev):
    from objectdetection_amd import ops
    cls = torch.zeros((1, 3, 20, 20), device=dev)
    assert not ops.rpn_in_place_supported([cls], 1000)
    with pytest.raises(RuntimeError, match="outside the split path's range"):
        ops.topk_desc_nchw([cls], 1000)


def test_resize_pad_batch_matches_per_image(dev):
    """The one-launch batch assembly writes the same bits as one resize_pad launch per image (40 images: two
    table loads), identity sizes, up- and down-scaling mixed."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(4)
    images, sizes = [], []
    for i in range(40):
        h, w = int(rng.integers(5, 90)), int(rng.integers(5, 120))
        images.append(T(rng.standard_normal((3, h, w)).astype(np.float32), dev))
        sizes.append((h, w) if i % 3 == 0 else (int(rng.integers(4, 96)), int(rng.integers(4, 128))))
    H, W = 96, 128
    a = torch.full((40, 3, H, W), 7.0, device=dev)
    b = torch.full((40, 3, H, W), -7.0, device=dev)
    ops.resize_pad_batch(images, sizes, a)
    for i, (im, (h, w)) in enumerate(zip(images, sizes)):
        ops.resize_pad_into(im, h, w, b[i])
    assert torch.equal(a, b)


# ---------------------------------------------------------------------------------- NMS (K4)
def lattice_boxes(n):
    side = int(np.ceil(np.sqrt(n)))
    ys, xs = np.divmod(np.arange(n), side)
    return np.stack([xs * 5.0, ys * 5.0, xs * 5.0 + 4, ys * 5.0 + 4], 1).astype(np.float32)


@pytest.mark.parametrize("kind,n,thr,max_keep", [
    ("realistic", 12000, 0.7, 2000), ("realistic", 3000, 0.5, 3000), ("realistic", 1000, 0.3, 50),
    ("lattice", 5000, 0.7, 2000), ("clustered", 4000, 0.7, 2000), ("realistic", 65, 0.7, 100),
    ("realistic", 64, 0.7, 64), ("realistic", 1, 0.7, 5), ("realistic", 30000, 0.7, 2000),
])
def test_nms_vs_oracle(dev, oracle, kind, n, thr, max_keep):
    from objectdetection_amd import ops
    rng = np.random.default_rng(n)
    if kind == "realistic":
        boxes = realistic_boxes(rng, n)
    elif kind == "lattice":
        boxes = lattice_boxes(n)
    else:  # many near-duplicates + degenerate boxes (NaN IoU must never suppress)
        base = realistic_boxes(rng, 40)
        boxes = base[rng.integers(0, 40, n)] + rng.normal(0, 2.0, (n, 4)).astype(np.float32)
        boxes[::50] = np.float32(7.0)
    scores = tie_free_scores(rng, n)
    ref = oracle.nms(boxes, scores, thr)
    got = ops.nms(T(boxes, dev), T(scores, dev), thr)
    assert got.dtype == torch.int64
    assert np.array_equal(got.cpu().numpy(), ref)
    # batched, pre-sorted, early-exit form
    order = oracle.argsort_desc(scores)
    sb = boxes[order]
    two = np.stack([sb, sb[::-1].copy()])
    npi = np.asarray([n, max(1, n // 2)], np.int32)
    keep, nk = ops.nms_sorted(T(two, dev), T(npi, dev, torch.int32), thr, max_keep)
    ref0 = oracle.nms(sb, -np.arange(n, dtype=np.float32), thr, max_keep)
    ref1 = oracle.nms(two[1][:npi[1]], -np.arange(npi[1], dtype=np.float32), thr, max_keep)
    assert int(nk[0]) == len(ref0) and int(nk[1]) == len(ref1)
    assert np.array_equal(keep[0, :len(ref0)].cpu().numpy(), ref0)
    assert np.array_equal(keep[1, :len(ref1)].cpu().numpy(), ref1)


def test_nms_threshold_edge(dev, oracle):
    """IoU exactly equal to the threshold must NOT suppress (strict >), also for thresholds whose
    fp32 rounding differs from the double (0.3): compare in double like torchvision's CPU kernel."""
    from objectdetection_amd import ops
    boxes = np.asarray([[0, 0, 10, 10], [0, 0, 10, 5], [0, 0, 10, 3], [20, 20, 30, 30]], np.float32)
    scores = np.asarray([4, 3, 2, 1], np.float32)
    for thr in (0.5, 0.3, 0.29999, 0.30000001192092896, 0.0, 1.0):
        ref = oracle.nms(boxes, scores, thr)
        got = ops.nms(T(boxes, dev), T(scores, dev), thr).cpu().numpy()
        assert np.array_equal(got, ref), thr


# ---------------------------------------------------------------------------- RoI ops (K6/K7)
def random_rois(rng, K, B, W, H):
    b = rng.integers(0, B, K).astype(np.float32)
    boxes = realistic_boxes(rng, K, W, H)
    return np.concatenate([b[:, None], boxes], 1).astype(np.float32)


@pytest.mark.parametrize("C,H,W,scale,K", [(8, 37, 62, 1 / 16, 300), (3, 112, 112, 0.5, 96), (33, 13, 21, 1 / 64, 50)])
def test_roi_align_vs_oracle(dev, oracle, C, H, W, scale, K):
    from objectdetection_amd import ops
    rng = np.random.default_rng(C * H)
    B = 2
    feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
    rois = random_rois(rng, K, B, W / scale, H / scale)
    rois[0, 1:] = [0, 0, W / scale, H / scale]      # whole image
    rois[1, 1:] = [5, 5, 5, 5]                      # empty box -> roi size clamps to 1
    rois[2, 1:] = [-40, -40, 30, 30]                # partly outside
    rois[3, 1:] = [W / scale + 50, 10, W / scale + 90, 60]  # fully outside -> zeros
    ref = oracle.roi_align(feat, rois, (7, 7), scale, 2)
    got = ops.roi_align(T(feat, dev), T(rois, dev), (7, 7), scale, 2).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=1e-5)
    assert np.all(got[3] == 0)


@pytest.mark.parametrize("C,H,W,Himg,Wimg,counts", [
    (40, 37, 62, 600, 1000, [300, 0, 77]),      # LDS-plane kernel, 16-channel chunks, ragged + empty image
    (3, 112, 112, 256, 256, [64, 32]),          # the reference's own roi_pooling smoke shapes (scale 1/2 -> fallback)
    (33, 13, 21, 800, 1344, [50, 50]),          # odd channel count: tail chunk of 1 channel
    (6, 50, 84, 800, 1344, [400]),              # 8-channel chunks
])
def test_roi_pooling_single_level_vs_oracle(dev, oracle, C, H, W, Himg, Wimg, counts):
    """ROIPooling.forward (list API) on one level == torchvision roi_align on the concatenated list."""
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    rng = np.random.default_rng(C + H)
    B = len(counts)
    feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
    props = [realistic_boxes(rng, n, Wimg, Himg) for n in counts]
    if counts[0] >= 4:
        props[0][0] = [0, 0, Wimg, Himg]
        props[0][1] = [5, 5, 5, 5]
        props[0][2] = [-40, -40, 30, 30]
        props[0][3] = [Wimg + 50, 10, Wimg + 90, 60]
    ref, _ = oracle.multiscale_roi_align([feat], props, [(Himg, Wimg)], (7, 7), 2)
    pool = ROIPooling(['1'], [7, 7], 2)
    got = pool({'1': T(feat, dev)}, [T(p, dev) for p in props], [(Himg, Wimg)])
    assert got.shape == ref.shape
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=RTOL, atol=1e-5)
    # determinism of the batched kernel
    got2 = pool({'1': T(feat, dev)}, [T(p, dev) for p in props], [(Himg, Wimg)])
    assert torch.equal(got, got2)


def _bins_with_a_sample_outside(props, scale, H, W):
    """(n, 49) bool: bins one of whose 2 x 2 samples lies outside [-1, H] x [-1, W] (fp32, the kernels' expression)."""
    f = np.float32
    p = np.asarray(props, np.float32)
    sw, sh = p[:, 0] * f(scale), p[:, 1] * f(scale)
    bw = np.maximum(p[:, 2] * f(scale) - sw, f(1)) / f(7)
    bh = np.maximum(p[:, 3] * f(scale) - sh, f(1)) / f(7)
    out_y = np.zeros((len(p), 7), bool)
    out_x = np.zeros((len(p), 7), bool)
    for q in range(7):
        for i in range(2):
            y = sh + f(q) * bh + (f(i) + f(0.5)) * bh / f(2)
            x = sw + f(q) * bw + (f(i) + f(0.5)) * bw / f(2)
            out_y[:, q] |= (y < -1) | (y > H)
            out_x[:, q] |= (x < -1) | (x > W)
    return (out_y[:, :, None] | out_x[:, None, :]).reshape(len(p), 49)


@pytest.mark.parametrize("fsz", [[(37, 62)], [(100, 168), (50, 84), (25, 42)]])
def test_roi_align_non_finite_features_stay_where_documented(dev, oracle, fsz):
    """DESIGN.md section 8.6, pinned so that it cannot widen silently.  A sample outside [-1, H] x [-1, W] carries
    weight 0; torchvision's device kernel returns before reading anything for it (the oracle's rule,
    oracle/frcnn_oracle.c), its CPU kernel multiplies feature[0] by that 0, and the kernels here multiply SOME cell of the
    map by it.  So with Inf / NaN in a map: (1) every bin without such a sample agrees with the oracle, finite or
    not; (2) a bin WITH one may come out non-finite where the oracle's is finite -- and nothing else may;
    (3) finite maps are not affected.  RoIs clipped to the image (everything the RPN emits) have no such sample."""
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    rng = np.random.default_rng(77)
    B, C = 2, 4
    Himg, Wimg = (600, 1000) if len(fsz) == 1 else (800, 1344)
    feats = [rng.standard_normal((B, C, h, w)).astype(np.float32) for h, w in fsz]
    for f_ in feats:
        f_[:, 1, 0, 0] = np.inf                  # channel 1: Inf in the corner cell (where weightless taps tend to land)
        f_[:, 2, -1, -1] = np.nan                # channel 2: NaN in the opposite corner
        f_[:, 2, f_.shape[2] // 2, f_.shape[3] // 2] = -np.inf
    n = 120
    props = [realistic_boxes(rng, n, Wimg, Himg) for _ in range(B)]
    for p_ in props:                             # a third of the boxes stick out of the image on some side
        k = n // 3
        p_[:k, 0] -= rng.uniform(20, 200, k)
        p_[:k, 2] += rng.uniform(20, 400, k)
        p_[k // 2:k, 1] -= rng.uniform(20, 200, k - k // 2)
        p_[k // 2:k, 3] += rng.uniform(20, 300, k - k // 2)
    ref, lvl = oracle.multiscale_roi_align(feats, props, [(Himg, Wimg)], (7, 7), 2)
    pool = ROIPooling([str(i + 1) for i in range(len(fsz))], [7, 7], 2)
    got = pool({str(i + 1): T(f_, dev) for i, f_ in enumerate(feats)}, [T(p_, dev) for p_ in props],
               [(Himg, Wimg)]).cpu().numpy()
    allp = np.concatenate(props)
    lvl = np.zeros(len(allp), int) if len(fsz) == 1 else np.asarray(lvl)
    outside = np.zeros((len(allp), 49), bool)
    for l, (h, w) in enumerate(fsz):
        sel = lvl == l
        outside[sel] = _bins_with_a_sample_outside(allp[sel], 2.0 ** round(float(np.log2(h / Himg))), h, w)
    assert outside.any() and (~outside).any()
    got, ref = got.reshape(len(allp), C, 49), ref.reshape(len(allp), C, 49)
    inside = np.broadcast_to(~outside[:, None, :], got.shape)
    # (1) bins whose samples are all inside: the oracle's value, finite or not
    assert np.array_equal(np.isfinite(got[inside]), np.isfinite(ref[inside]))
    both = inside & np.isfinite(ref)
    np.testing.assert_allclose(got[both], ref[both], rtol=RTOL, atol=1e-5)
    assert (inside & ~np.isfinite(ref)).any()     # (Inf or NaN may differ: a merged tap adds 0 x Inf to an Inf)
    # (2) bins with a sample outside: finite ones agree; non-finite ones are the documented divergence
    edge = ~inside & np.isfinite(got) & np.isfinite(ref)
    np.testing.assert_allclose(got[edge], ref[edge], rtol=RTOL, atol=1e-5)
    assert not (np.isfinite(got) & ~np.isfinite(ref)).any()          # never finite where the oracle is not
    # (3) the finite channels are untouched
    np.testing.assert_allclose(got[:, [0, 3]], ref[:, [0, 3]], rtol=RTOL, atol=1e-5)


def test_multiscale_roi_align_vs_oracle(dev, oracle):
    """FPN pyramid, fine-to-coarse order: level assignment bit-exact, values 1e-5."""
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    from objectdetection_amd import ops
    rng = np.random.default_rng(5)
    B, C = 2, 16
    Himg, Wimg = 800, 1344
    fsz = [(200, 336), (100, 168), (50, 84), (25, 42)]
    feats = [rng.standard_normal((B, C, h, w)).astype(np.float32) for h, w in fsz]
    props = [realistic_boxes(rng, 150, Wimg, Himg), realistic_boxes(rng, 90, Wimg, Himg)]
    props[0][0] = [100, 100, 324, 324]   # sqrt(area) == 224 exactly -> level 4
    props[0][1] = [0, 0, 112, 112]       # exact power of two below
    props[0][2] = [10, 10, 5, 20]        # negative area -> NaN level -> zeros
    ref, ref_lvl = oracle.multiscale_roi_align(feats, props, [(Himg, Wimg)], (7, 7), 2)
    pool = ROIPooling(['1', '2', '3', '4'], [7, 7], 2)
    d = {str(i + 1): T(f, dev) for i, f in enumerate(feats)}
    got = pool(d, [T(p, dev) for p in props], [(Himg, Wimg)])
    rois = torch.cat([torch.cat([torch.full((len(p), 1), float(i)), torch.as_tensor(p)], 1) for i, p in enumerate(props)])
    lvl = ops.level_map(rois.to(dev), 2, 5)
    assert np.array_equal(lvl.cpu().numpy(), ref_lvl)
    assert len(np.unique(ref_lvl)) >= 4
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=RTOL, atol=1e-5)
    # padded, sync-free form gives the same rows
    P = 150
    padded = np.zeros((B, P, 4), np.float32)
    padded[0] = props[0]; padded[1, :90] = props[1]
    out, total = pool.forward_padded(d, T(padded, dev), T([150, 90], dev, torch.int32), [(Himg, Wimg)])
    assert int(total) == 240
    np.testing.assert_allclose(out[:240].cpu().numpy(), ref, rtol=RTOL, atol=1e-5)
    # coarse-to-fine order (the reference backbone's order): k_min > k_max -> nothing matches
    pool_rev = ROIPooling(['1', '2', '3', '4'], [7, 7], 2)
    d_rev = {str(i + 1): T(f, dev) for i, f in enumerate(feats[::-1])}
    ref_rev, _ = oracle.multiscale_roi_align(feats[::-1], props, [(Himg, Wimg)], (7, 7), 2)
    got_rev = pool_rev(d_rev, [T(p, dev) for p in props], [(Himg, Wimg)])
    np.testing.assert_allclose(got_rev.cpu().numpy(), ref_rev, rtol=RTOL, atol=1e-5)


def test_multiscale_roi_align_config5_scale(dev, oracle):
    """SURVEY 8d config 5 shapes: C = 256, pyramid 100x168 .. 13x21 on an 800x1344 image, 2,000 RoIs.
    The 100x168 level is too large for the LDS-resident kernel and takes the direct path."""
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    rng = np.random.default_rng(23)
    B, C, Himg, Wimg = 1, 256, 800, 1344
    fsz = [(100, 168), (50, 84), (25, 42), (13, 21)]
    feats = [rng.standard_normal((B, C, h, w)).astype(np.float32) for h, w in fsz]
    props = [realistic_boxes(rng, 2000, Wimg, Himg)]
    props[0][:40, :2] = rng.uniform(0, 300, (40, 2))          # some large boxes for the coarsest level
    props[0][:40, 2:] = props[0][:40, :2] + rng.uniform(460, 500, (40, 2))
    props[0][40:50] = np.concatenate([rng.uniform(0, 20, (10, 2)), rng.uniform(0, 20, (10, 2)) + [1300, 760]], 1)
    ref, ref_lvl = oracle.multiscale_roi_align(feats, props, [(Himg, Wimg)], (7, 7), 2)
    assert len(np.unique(ref_lvl)) == 4
    pool = ROIPooling(['1', '2', '3', '4'], [7, 7], 2)
    d = {str(i + 1): T(f, dev) for i, f in enumerate(feats)}
    out, total = pool.forward_padded(d, T(np.stack(props), dev), T([2000], dev, torch.int32), [(Himg, Wimg)])
    assert int(total) == 2000
    np.testing.assert_allclose(out[:2000].cpu().numpy(), ref, rtol=RTOL, atol=1e-5)


def _cl_case_boxes(rng, counts, Wimg, Himg):
    props = []
    for n in counts:
        s = np.exp(rng.uniform(np.log(4), np.log(max(Himg, Wimg) * 1.2), (n, 2)))
        c = rng.uniform(-0.1, 1.1, (n, 2)) * [Wimg, Himg]
        props.append(np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32))
    return props


@pytest.mark.parametrize("C,fsz,stride", [
    (64, [(37, 62)], 16), (128, [(200, 336), (100, 168), (50, 84), (25, 42)], 4), (192, [(96, 128), (48, 64), (24, 32)], 8),
    (64, [(1, 1)], 64), (64, [(2, 70)], 16), (256, [(100, 168), (50, 84), (25, 42), (13, 21)], 8)])
def test_roi_align_lanes_are_channels_path_vs_oracle(dev, oracle, C, fsz, stride):
    """csrc/roi_cl.hip (lanes = channels, frcnn_roi_align_levels_cl_f32) against the oracle, on maps that are
    channels-last (read in place) and on NCHW maps (copied into the workspace first): boxes inside, across and
    outside the map, tiny and map-sized, a NaN-level box (zero rows), ragged counts incl. an image without boxes.
    The two layouts run the same arithmetic: their outputs are equal bit for bit."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(C + len(fsz) + fsz[0][1])
    B = 3
    Himg, Wimg = fsz[0][0] * stride, fsz[0][1] * stride
    feats = [rng.standard_normal((B, C, h, w)).astype(np.float32) for h, w in fsz]
    counts = [70, 0, 23]
    props = _cl_case_boxes(rng, counts, Wimg, Himg)
    props[0][0] = [10, 10, 5, 20]                                     # negative area: NaN level, zero rows
    props[0][1] = [0, 0, Wimg, Himg]                                  # the whole image
    props[0][2] = [Wimg - 3, Himg - 3, Wimg, Himg]                    # the last cell
    props[0][3] = [Wimg + 40, 10, Wimg + 90, 60]                      # beside the map
    props[0][4] = [33.3, 47.1, 33.3, 47.1]                            # zero size
    ref, _ = oracle.multiscale_roi_align(feats, props, [(Himg, Wimg)], (7, 7), 2)
    P = 70
    padded = np.zeros((B, P, 4), np.float32)
    for b, p_ in enumerate(props):
        padded[b, :len(p_)] = p_
    pool = ops.MultiScaleRoIAlign([str(i + 1) for i in range(len(fsz))], (7, 7), 2)
    levels = pool._levels({str(i + 1): T(f, dev) for i, f in enumerate(feats)}, [(Himg, Wimg)])
    fd, scales, k_min, k_max = levels
    cfg = ((7, 7), 2, tuple(float(s_) for s_ in scales), int(k_min), int(k_max))
    pd, cd = T(padded, dev), T(counts, dev, torch.int32)
    n = sum(counts)
    outs = []
    for cl_maps in (False, True):
        maps = [f.contiguous(memory_format=torch.channels_last) for f in fd] if cl_maps else fd
        out, total = ops._roi_align_levels(cfg, pd, cd, maps, path="cl")
        assert int(total) == n
        np.testing.assert_allclose(out[:n].cpu().numpy(), ref, rtol=RTOL, atol=2e-5,
                                   err_msg="C=%d planes=%s channels_last=%s" % (C, fsz, cl_maps))
        outs.append(out[:n].clone())
    assert torch.equal(outs[0], outs[1])
    # and the tile / plane kernels agree with it to the same tolerance (different summation order)
    tiles, _ = ops._roi_align_levels(cfg, pd, cd, fd, path="tiles")
    np.testing.assert_allclose(tiles[:n].cpu().numpy(), outs[0].cpu().numpy(), rtol=RTOL, atol=2e-5)
    # channels-last maps go down this path by themselves (auto) unless the plane kernel holds the map
    auto, _ = ops._roi_align_levels(cfg, pd, cd, [f.contiguous(memory_format=torch.channels_last) for f in fd])
    np.testing.assert_allclose(auto[:n].cpu().numpy(), ref, rtol=RTOL, atol=2e-5)


def test_roi_align_lanes_are_channels_rejects_what_it_cannot_do(dev):
    """C not a multiple of 64, other bin counts: the entry point refuses (FRCNN_EINVAL) and the wrapper falls back
    to the tile / plane kernels instead of computing something else."""
    from objectdetection_amd import ops, _lib
    lib = _lib.load()
    import ctypes
    hw = (ctypes.c_int32 * 2)(50, 84)
    assert lib.frcnn_roi_align_levels_cl_workspace_bytes(hw, 1, 2, 48, 10, 0) == 0
    assert lib.frcnn_roi_align_levels_cl_workspace_bytes(hw, 1, 2, 64, 10, 0) > \
        lib.frcnn_roi_align_levels_cl_workspace_bytes(hw, 1, 2, 64, 10, 1) > 0
    f = torch.randn(2, 48, 50, 84, device=dev)
    props = torch.tensor([[[8.0, 8.0, 100.0, 90.0]], [[30.0, 20.0, 300.0, 200.0]]], device=dev)
    cfg = ((7, 7), 2, (0.125,), 3, 3)
    a, _ = ops._roi_align_levels(cfg, props, None, [f], path="cl")      # falls back: C = 48
    b, _ = ops._roi_align_levels(cfg, props, None, [f], path="tiles")
    assert torch.equal(a, b)
    out = torch.empty(2, 64, 7, 7, device=dev)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    ptrs = (ctypes.c_void_p * 1)(f.data_ptr())
    sc = (ctypes.c_float * 1)(0.125)
    rc = lib.frcnn_roi_align_levels_cl_f32(ptrs, hw, sc, 1, 2, 48, 0, ctypes.c_void_p(props.data_ptr()), None, 1, 7, 7, 2,
                                           3, 3, ctypes.c_void_p(out.data_ptr()), None, ctypes.c_void_p(ws.data_ptr()),
                                           ws.numel(), None)
    assert rc != 0 and b"multiple of 64" in lib.frcnn_last_error_string()
    rc = lib.frcnn_roi_align_levels_cl_f32(ptrs, hw, sc, 1, 2, 64, 0, ctypes.c_void_p(props.data_ptr()), None, 1, 7, 7, 2,
                                           3, 3, ctypes.c_void_p(out.data_ptr()), None, ctypes.c_void_p(ws.data_ptr()),
                                           16, None)
    assert rc != 0 and b"workspace" in lib.frcnn_last_error_string()


def test_roi_align_lanes_are_channels_non_finite_features(dev, oracle):
    """The rule of test_roi_align_non_finite_features_stay_where_documented for csrc/roi_cl.hip: a cell row is added
    to a bin row only where it carries weight there, so bins whose samples all lie inside the map have the oracle's
    finiteness exactly; only a bin with a sample outside [-1, H] x [-1, W] may turn non-finite where the oracle's
    stays finite (a weightless x tap still multiplies some cell of the row)."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(78)
    B, C = 2, 64
    fsz = [(100, 168), (50, 84), (25, 42)]
    Himg, Wimg = 800, 1344
    feats = [rng.standard_normal((B, C, h, w)).astype(np.float32) for h, w in fsz]
    for f_ in feats:
        f_[:, 1, 0, 0] = np.inf
        f_[:, 2, -1, -1] = np.nan
        f_[:, 2, f_.shape[2] // 2, f_.shape[3] // 2] = -np.inf
        f_[:, 40, f_.shape[2] // 3, :] = np.inf                       # a whole row of one channel
    n = 120
    props = [realistic_boxes(rng, n, Wimg, Himg) for _ in range(B)]
    for p_ in props:
        k = n // 3
        p_[:k, 0] -= rng.uniform(20, 200, k)
        p_[:k, 2] += rng.uniform(20, 400, k)
        p_[k // 2:k, 1] -= rng.uniform(20, 200, k - k // 2)
        p_[k // 2:k, 3] += rng.uniform(20, 300, k - k // 2)
    ref, lvl = oracle.multiscale_roi_align(feats, props, [(Himg, Wimg)], (7, 7), 2)
    pool = ops.MultiScaleRoIAlign(['1', '2', '3'], (7, 7), 2)
    fd, scales, k_min, k_max = pool._levels({str(i + 1): T(f, dev) for i, f in enumerate(feats)}, [(Himg, Wimg)])
    cfg = ((7, 7), 2, tuple(float(s_) for s_ in scales), int(k_min), int(k_max))
    got, total = ops._roi_align_levels(cfg, T(np.stack(props), dev), None, fd, path="cl")
    got = got.cpu().numpy()
    allp = np.concatenate(props)
    lvl = np.asarray(lvl)
    outside = np.zeros((len(allp), 49), bool)
    for l, (h, w) in enumerate(fsz):
        sel = lvl == l
        outside[sel] = _bins_with_a_sample_outside(allp[sel], 2.0 ** round(float(np.log2(h / Himg))), h, w)
    assert outside.any() and (~outside).any()
    got, ref = got.reshape(len(allp), C, 49), ref.reshape(len(allp), C, 49)
    inside = np.broadcast_to(~outside[:, None, :], got.shape)
    assert np.array_equal(np.isfinite(got[inside]), np.isfinite(ref[inside]))
    both = inside & np.isfinite(ref)
    np.testing.assert_allclose(got[both], ref[both], rtol=RTOL, atol=1e-5)
    assert (inside & ~np.isfinite(ref)).any()
    edge = ~inside & np.isfinite(got) & np.isfinite(ref)
    np.testing.assert_allclose(got[edge], ref[edge], rtol=RTOL, atol=1e-5)
    assert not (np.isfinite(got) & ~np.isfinite(ref)).any()
    np.testing.assert_allclose(got[:, [0, 3, 63]], ref[:, [0, 3, 63]], rtol=RTOL, atol=1e-5)


@pytest.mark.parametrize("C,H,W", [(20, 100, 168), (16, 70, 49), (5, 200, 336)])
def test_roi_align_window_kernel_edges(dev, oracle, C, H, W):
    """Planes too large for LDS (window kernel + direct kernel for the RoIs that fit no window): RoIs on
    every border, a RoI wider than a window, one larger than the plane, degenerate ones, channel counts
    that are and are not a multiple of 16, three images with different counts."""
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    rng = np.random.default_rng(C + H)
    stride = 8
    Himg, Wimg = H * stride, W * stride
    B = 3
    feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
    props = []
    for b, n in enumerate((97, 1, 40)):
        s = rng.uniform(8, 200, (n, 2))
        c = rng.uniform(0, 1, (n, 2)) * [Wimg, Himg]
        bx = np.concatenate([c - s / 2, c + s / 2], 1)
        bx[:, 0::2] = bx[:, 0::2].clip(0, Wimg)
        bx[:, 1::2] = bx[:, 1::2].clip(0, Himg)
        props.append(bx.astype(np.float32))
    props[0][0] = [0, 0, 30, 30]                              # top-left corner
    props[0][1] = [Wimg - 20, Himg - 25, Wimg, Himg]          # bottom-right corner
    props[0][2] = [0, Himg - 40, 50, Himg]                    # bottom-left
    props[0][3] = [100, 60, 100 + 47 * stride, 90]            # 47 cells wide: fits no window -> direct kernel
    props[0][4] = [-50, -50, Wimg + 80, Himg + 80]            # larger than the plane
    props[0][5] = [50, 50, 50, 50]                            # zero size -> roi_w = roi_h = 1
    props[0][6] = [Wimg + 30, 10, Wimg + 60, 40]              # entirely right of the map: zeros
    props[0][7] = [128, 128, 128 + 33 * stride - 1, 128 + 33 * stride - 1]   # just inside the window limit
    props[0][8] = [120, 120, 120 + 34 * stride, 120 + 34 * stride]           # just outside it
    ref, _ = oracle.multiscale_roi_align([feat], props, [(Himg, Wimg)], (7, 7), 2)
    pool = ROIPooling(['1'], [7, 7], 2)
    P = 97
    padded = np.zeros((B, P, 4), np.float32)
    for b, p in enumerate(props):
        padded[b, :len(p)] = p
    out, total = pool.forward_padded({'1': T(feat, dev)}, T(padded, dev), T([97, 1, 40], dev, torch.int32),
                                     [(Himg, Wimg)])
    assert int(total) == 138
    np.testing.assert_allclose(out[:138].cpu().numpy(), ref, rtol=RTOL, atol=1e-5)


def test_roi_align_full_size_properties(dev, oracle):
    """BASELINE config 2 at full size (16 images x 2,000 RoIs, C = 512, 37x62: 3.2 GB of output), where the CPU
    oracle cannot check every value: (a) 96 RoIs spread over the batch against the oracle, all channels;
    (b) linearity in the features, RoIAlign(2 F + G) = 2 RoIAlign(F) + RoIAlign(G), on a strided sample;
    (c) a constant plane pools to that constant for every RoI inside the map; (d) determinism."""
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    rng = np.random.default_rng(2024)
    B, P, C, H, W, Himg, Wimg = 16, 2000, 512, 37, 62, 600, 1000
    g = torch.Generator().manual_seed(7)
    F = torch.randn(B, C, H, W, generator=g)
    G = torch.randn(B, C, H, W, generator=g)
    props = np.stack([realistic_boxes(rng, P, Wimg, Himg) for _ in range(B)])
    counts = np.asarray([P] * (B - 2) + [P - 37, 1], np.int32)
    pool = ROIPooling(['1'], [7, 7], 2)
    pd, cd = T(props, dev), T(counts, dev, torch.int32)
    Fd, Gd = F.to(dev), G.to(dev)
    outF, total = pool.forward_padded({'1': Fd}, pd, cd, [(Himg, Wimg)])
    n = int(total)
    assert n == int(counts.sum())
    assert bool(torch.isfinite(outF[:n]).all())
    # (a) sample against the oracle
    offs = np.concatenate([[0], np.cumsum(counts)])
    pick = [(b, j) for b in (0, 5, 14, 15) for j in rng.integers(0, counts[b], 24)]
    for b in (0, 5, 14, 15):
        js = sorted({j for bb, j in pick if bb == b})
        ref, _ = oracle.multiscale_roi_align([F[b:b + 1].numpy()], [props[b][js]], [(Himg, Wimg)], (7, 7), 2)
        got = outF[torch.as_tensor(offs[b] + np.asarray(js), device=dev)].cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=RTOL, atol=1e-5)
    # (d) determinism
    outF2, _ = pool.forward_padded({'1': Fd}, pd, cd, [(Himg, Wimg)])
    assert torch.equal(outF[:n], outF2[:n])
    del outF2
    # (b) linearity on every 97th row
    rows = torch.arange(0, n, 97, device=dev)
    a = outF[rows].clone()
    del outF
    outG, _ = pool.forward_padded({'1': Gd}, pd, cd, [(Himg, Wimg)])
    bb = outG[rows].clone()
    del outG
    outL, _ = pool.forward_padded({'1': 2 * Fd + Gd}, pd, cd, [(Himg, Wimg)])
    torch.testing.assert_close(outL[rows], 2 * a + bb, rtol=1e-4, atol=1e-4)
    del outL
    # (c) constant plane; proposals are clipped to the image, so every sample lies inside the map except on
    # the right / bottom border cells beyond W-1 / H-1 -- keep RoIs that end before them
    inside = (props[..., 2] < (W - 1) * 16) & (props[..., 3] < (H - 1) * 16)
    const = torch.full((B, C, H, W), 3.25, device=dev)
    outC, _ = pool.forward_padded({'1': const}, pd, cd, [(Himg, Wimg)])
    mask = np.concatenate([inside[b, :counts[b]] for b in range(B)])
    sel = torch.as_tensor(np.nonzero(mask)[0][::53], device=dev)
    torch.testing.assert_close(outC[sel], torch.full_like(outC[sel], 3.25), rtol=1e-6, atol=0)


def test_roi_align_shape_sweep(dev, oracle):
    """Seeded sweep over plane sizes, channel counts, strides and pyramid depths: every dispatch of the batched
    RoIAlign (16-, 8- and 4-channel LDS planes, the generic instantiation, windows, the direct kernel) against
    the oracle."""
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    rng = np.random.default_rng(777)
    cases = [  # (C, [(H, W) per level, fine to coarse], stride of the finest level)
        (16, [(37, 62)], 16), (32, [(36, 64)], 16), (7, [(37, 62)], 16), (33, [(20, 30)], 32),
        (16, [(48, 48)], 16), (16, [(49, 49)], 16), (12, [(64, 96)], 8), (40, [(75, 100)], 8),
        (16, [(96, 128), (48, 64), (24, 32)], 8), (24, [(128, 200), (64, 100), (32, 50), (16, 25)], 4),
        (3, [(8, 8)], 32), (64, [(1, 1)], 64), (16, [(2, 70)], 16),
    ]
    for C, fsz, stride in cases:
        B = int(rng.integers(1, 4))
        Himg, Wimg = fsz[0][0] * stride, fsz[0][1] * stride
        feats = [rng.standard_normal((B, C, h, w)).astype(np.float32) for h, w in fsz]
        counts = [int(rng.integers(0, 60)) for _ in range(B)]
        counts[int(rng.integers(0, B))] = 60
        props = []
        for n in counts:
            s = np.exp(rng.uniform(np.log(4), np.log(max(Himg, Wimg) * 1.2), (n, 2)))
            c = rng.uniform(-0.1, 1.1, (n, 2)) * [Wimg, Himg]
            props.append(np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32))
        names = [str(i + 1) for i in range(len(fsz))]
        ref, _ = oracle.multiscale_roi_align(feats, props, [(Himg, Wimg)], (7, 7), 2)
        pool = ROIPooling(names, [7, 7], 2)
        P = 60
        padded = np.zeros((B, P, 4), np.float32)
        for b, p in enumerate(props):
            padded[b, :len(p)] = p
        d = {k: T(f, dev) for k, f in zip(names, feats)}
        out, total = pool.forward_padded(d, T(padded, dev), T(counts, dev, torch.int32), [(Himg, Wimg)])
        assert int(total) == sum(counts), (C, fsz)
        np.testing.assert_allclose(out[:sum(counts)].cpu().numpy(), ref, rtol=RTOL, atol=2e-5,
                                   err_msg="C=%d planes=%s stride=%d" % (C, fsz, stride))
        got = pool(d, [T(p, dev) for p in props], [(Himg, Wimg)])      # the reference's list API
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=RTOL, atol=2e-5)


@pytest.mark.parametrize("B,C,hw", [(3, 512, (37, 62)), (5, 256, (37, 62)), (7, 64, (50, 84)), (3, 32, (100, 168))])
def test_roi_align_block_order_padding(dev, oracle, B, C, hw):
    """The 1-D, XCD-grouped grids of the plane and window kernels pad the number of (image, split) or
    (window, image) groups to a multiple of 8: batch sizes for which that padding is not empty."""
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    rng = np.random.default_rng(B * 1000 + C)
    h, w = hw
    stride = 16 if h < 60 else 8
    Himg, Wimg = h * stride, w * stride
    feat = rng.standard_normal((B, C, h, w)).astype(np.float32)
    P = 40
    counts = [int(rng.integers(1, P + 1)) for _ in range(B)]
    padded = np.zeros((B, P, 4), np.float32)
    props = []
    for b, n in enumerate(counts):
        sz = np.exp(rng.uniform(np.log(8), np.log(min(Himg, Wimg) * 0.6), (n, 2)))
        c = rng.uniform(0.05, 0.95, (n, 2)) * [Wimg, Himg]
        p = np.concatenate([c - sz / 2, c + sz / 2], 1).astype(np.float32)
        props.append(p)
        padded[b, :n] = p
    ref, _ = oracle.multiscale_roi_align([feat], props, [(Himg, Wimg)], (7, 7), 2)
    pool = ROIPooling(['1'], [7, 7], 2)
    out, total = pool.forward_padded({'1': T(feat, dev)}, T(padded, dev), T(counts, dev, torch.int32), [(Himg, Wimg)])
    assert int(total) == sum(counts)
    np.testing.assert_allclose(out[:sum(counts)].cpu().numpy(), ref, rtol=RTOL, atol=2e-5)


def test_nms_size_sweep(dev, oracle):
    """Seeded sweep over candidate counts around the lazy / round boundaries and keep budgets."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(4242)
    for n, max_keep, thr in [(1, 5, 0.5), (63, 63, 0.3), (64, 10, 0.7), (65, 65, 0.7), (1023, 1000, 0.5),
                             (1024, 2000, 0.7), (1025, 17, 0.7), (2048, 2000, 0.6), (2049, 2049, 0.7),
                             (3000, 1, 0.7), (4097, 300, 0.45), (9000, 2000, 0.7)]:
        B = 3
        boxes = np.stack([realistic_boxes(rng, n, 800, 600) for _ in range(B)])
        counts = np.asarray([n, max(1, n // 2), n - 1 if n > 1 else 1], np.int32)
        keep, nk = ops.nms_sorted(T(boxes, dev), T(counts, dev, torch.int32), thr, max_keep)
        for b in range(B):
            c = int(counts[b])
            ref = oracle.nms(boxes[b, :c], np.arange(c, 0, -1, dtype=np.float32), thr)[:max_keep]
            assert int(nk[b]) == len(ref), (n, max_keep, thr, b)
            assert np.array_equal(keep[b, :len(ref)].cpu().numpy(), ref), (n, max_keep, thr, b)


def test_nms_workspace_needs_no_clearing(dev, oracle):
    """The round path (n > 2048) through the C ABI with a workspace full of 0xFF bytes: round 0 reads nothing of
    it, so the caller need not clear it.  Images with 0, 1, 700 (one round), 2,500 and 3,000 candidates."""
    from objectdetection_amd import _lib
    rng = np.random.default_rng(31)
    ns = [0, 1, 700, 2500, 3000]
    B, n, P = len(ns), 3000, 600
    boxes = np.stack([realistic_boxes(rng, n, 320, 256) for _ in ns])
    bx = T(boxes, dev)
    cnt = T(np.asarray(ns, np.int32), dev, torch.int32)
    lib = _lib.load()
    nbytes = lib.frcnn_nms_workspace_bytes(B, n)
    assert nbytes > 0
    ws = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device=dev)
    keep = torch.full((B, P), -7, dtype=torch.int32, device=dev)
    n_keep = torch.full((B,), -7, dtype=torch.int32, device=dev)
    rc = lib.frcnn_nms_sorted_f32(bx.data_ptr(), cnt.data_ptr(), B, n, 0.7, P, keep.data_ptr(), n_keep.data_ptr(),
                                  ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.frcnn_last_error_string()
    torch.cuda.synchronize()
    nk = n_keep.cpu().numpy()
    for i, m in enumerate(ns):
        ref = oracle.nms(boxes[i, :m], np.arange(m, 0, -1, dtype=np.float32), 0.7)[:P] if m else np.zeros((0,), np.int64)
        assert nk[i] == len(ref), (i, nk[i], len(ref))
        assert np.array_equal(keep[i, :nk[i]].cpu().numpy(), ref)


@pytest.mark.parametrize("max_blocks", [1, 16, 20, 37, 80, 0])
def test_nms_round_bound_never_changes_the_result(dev, oracle, max_blocks):
    """frcnn_nms_sorted_rounds_f32 through the C ABI: with the rounds covering fewer 64-candidate blocks than the
    batch needs (whole rounds, or a last round cut short), the last round's tail finishes the images that are not done
    and the keep lists are those of the unbounded call (and of the oracle).  Images: empty, one round, max_keep reached in round 2, all 6 rounds needed (sparse boxes, few keeps),
    ragged.  rounds_used (pinned host memory the kernels write directly) = rounds that had work, or all of them
    for an image the tail had to finish."""
    from objectdetection_amd import _lib
    rng = np.random.default_rng(77)
    n, P = 6000, 900
    ns = [0, 700, 6000, 6000, 3500]
    boxes = np.stack([realistic_boxes(rng, n, 1000, 600) for _ in ns])
    # image 3: heavy overlap -> few keeps, every round has work
    c = rng.uniform(100, 140, (n, 2)).astype(np.float32)
    wh = rng.uniform(150, 170, (n, 2)).astype(np.float32)
    boxes[3] = np.concatenate([c, c + wh], 1)
    B = len(ns)
    bx = T(boxes, dev)
    cnt = T(np.asarray(ns, np.int32), dev, torch.int32)
    lib = _lib.load()
    nbytes = lib.frcnn_nms_workspace_bytes(B, n)
    st = torch.cuda.current_stream().cuda_stream

    def run(bound):
        ws = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device=dev)
        keep = torch.full((B, P), -7, dtype=torch.int32, device=dev)
        n_keep = torch.full((B,), -7, dtype=torch.int32, device=dev)
        used = torch.full((B,), -7, dtype=torch.int32).pin_memory()
        rc = lib.frcnn_nms_sorted_rounds_f32(bx.data_ptr(), cnt.data_ptr(), B, n, 0.7, P, keep.data_ptr(),
                                             n_keep.data_ptr(), ws.data_ptr(), nbytes, bound, used.data_ptr(), st)
        assert rc == 0, lib.frcnn_last_error_string()
        torch.cuda.synchronize()
        return keep.cpu().numpy(), n_keep.cpu().numpy(), used.numpy().copy()

    keep, nk, used = run(max_blocks)
    keep0, nk0, used0 = run(0)
    assert np.array_equal(nk, nk0) and np.array_equal(keep, keep0)
    for i, m in enumerate(ns):
        ref = oracle.nms(boxes[i, :m], np.arange(m, 0, -1, dtype=np.float32), 0.7)[:P] if m else np.zeros((0,), np.int64)
        assert nk[i] == len(ref), (i, nk[i], len(ref))
        assert np.array_equal(keep[i, :nk[i]], ref)
    # rounds_used counts 64-candidate blocks that had work: how far into the candidates the image got
    for u in (used, used0):
        assert u[0] == 0
        for i in range(1, B):
            assert 1 <= u[i] <= (ns[i] + 63) // 64, (i, u)
        assert u[1] <= 11 and u[3] > 80       # 700 boxes cannot fill 900 keeps; image 3 needs its last round
        assert u[2] < 80                      # the realistic image is done long before its candidates end


def test_nms_rounds_feed_forward(dev):
    """ops.nms_sorted(rounds=NmsRounds()): the first call enqueues every round, later calls one more than the
    finished calls needed; results identical, and a batch that needs more than the bound still comes out right."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(78)
    B, n, P = 3, 5000, 400
    easy = T(np.stack([realistic_boxes(rng, n, 1000, 600) for _ in range(B)]), dev)
    c = rng.uniform(100, 140, (B, n, 2)).astype(np.float32)
    hard = T(np.concatenate([c, c + rng.uniform(150, 170, (B, n, 2)).astype(np.float32)], 2), dev)
    ref_easy = ops.nms_sorted(easy, None, 0.7, P)
    ref_hard = ops.nms_sorted(hard, None, 0.7, P)
    r = ops.NmsRounds()
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), B, n, P, 0.7)
    assert r.bound(key, B)[0] == 0                      # nothing known: every round
    for expect_ref, boxes in ((ref_easy, easy), (ref_easy, easy), (ref_hard, hard), (ref_hard, hard), (ref_easy, easy)):
        keep, nk = ops.nms_sorted(boxes, None, 0.7, P, rounds=r)
        torch.cuda.synchronize()
        assert torch.equal(nk, expect_ref[1])
        for b in range(B):
            assert torch.equal(keep[b, :int(nk[b])], expect_ref[0][b, :int(nk[b])])
    assert r.bound(key, B)[0] >= 5000 // 64             # the hard batches are still among the last observations
    for _ in range(r.history):
        ops.nms_sorted(easy, None, 0.7, P, rounds=r)
        torch.cuda.synchronize()
    assert 1 <= r.bound(key, B)[0] <= 5 * 16            # only easy batches remembered: their reach + slack


def test_roi_pool_max_vs_oracle(dev, oracle):
    from objectdetection_amd import ops
    rng = np.random.default_rng(11)
    B, C, H, W = 2, 19, 37, 62
    feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
    rois = random_rois(rng, 200, B, 1000, 600)
    rois[0, 1:] = [990, 590, 1400, 900]   # mostly outside: empty bins -> 0 / -1
    ref, ref_arg = oracle.roi_pool(feat, rois, (7, 7), 1 / 16)
    got, arg = ops.roi_pool(T(feat, dev), T(rois, dev), (7, 7), 1 / 16)
    assert np.array_equal(arg.cpu().numpy(), ref_arg)
    assert_bits_equal(got.cpu().numpy(), ref)
    assert (ref_arg == -1).any()


@pytest.mark.parametrize("B,C,H,W,counts", [(2, 19, 37, 62, [150, 37]), (3, 32, 37, 62, [0, 600, 1]),
                                             (1, 5, 20, 30, [64]), (2, 16, 50, 84, [40, 40])])
def test_roi_pool_batched_vs_oracle(dev, oracle, B, C, H, W, counts):
    """The plane-scheme max RoIPool over padded proposals (K7 at batch shapes): values and argmax bit-exact against
    the oracle's roi_pool on the concatenated roi list; ragged counts, tail channels, a wider map (generic
    instantiation), RoIs running off the map (empty bins -> 0 / -1)."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(C * 7 + H)
    scale = 1 / 16
    feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
    feat[0, 0, :3, :5] = 1.25      # a flat patch: the first maximum must win
    P = max(counts)
    props = np.zeros((B, P, 4), np.float32)
    rows = []
    for b, n in enumerate(counts):
        bx = realistic_boxes(rng, n, W * 16.0, H * 16.0)
        if n > 2:
            bx[0] = [W * 16 - 20, H * 16 - 20, W * 16 + 400, H * 16 + 300]   # mostly outside
            bx[1] = [0, 0, 70, 40]                                            # over the flat patch
            bx[2] = [40.5, 24.5, 136.5, 120.5]                                # halves: round away from zero
        props[b, :n] = bx
        rows.append(np.concatenate([np.full((n, 1), b, np.float32), bx], 1))
    rois = np.concatenate(rows)
    ref, ref_arg = oracle.roi_pool(feat, rois, (7, 7), scale)
    out, arg, total = ops.roi_pool_batched(T(feat, dev), T(props, dev), T(counts, dev, torch.int32), (7, 7), scale)
    n = sum(counts)
    assert int(total) == n
    assert np.array_equal(arg[:n].cpu().numpy(), ref_arg)
    assert_bits_equal(out[:n].cpu().numpy(), ref)


def test_roi_pool_batched_full_size_and_module(dev, oracle):
    """BASELINE config 2 shape (16 x 2,000 RoIs, C = 512) through ROIPooling(mode='max').forward_padded: a sample of
    rows against the oracle, bit-exact; argmax always points at the value it reports."""
    from objectdetection_amd.ROIHead import ROIPooling
    rng = np.random.default_rng(2)
    B, C, H, W, P = 16, 512, 37, 62, 2000
    feat = torch.randn((B, C, H, W), generator=torch.Generator().manual_seed(3)).to(dev)
    props = np.stack([realistic_boxes(rng, P) for _ in range(B)])
    counts = np.full(B, P, np.int32)
    counts[3] = 1234
    pool = ROIPooling(['1'], [7, 7], 2, mode="max")
    out, total = pool.forward_padded({'1': feat}, T(props, dev), T(counts, dev, torch.int32), [(600, 1000)])
    n = int(counts.sum())
    assert int(total) == n
    arg = pool.last_argmax
    row0 = np.concatenate([[0], np.cumsum(counts)])
    sel = rng.choice(n, 40, replace=False)
    img = np.searchsorted(row0, sel, side="right") - 1
    rois = np.stack([np.concatenate([[b], props[b, r - row0[b]]]) for b, r in zip(img, sel)]).astype(np.float32)
    chans = rng.choice(C, 24, replace=False)
    ref, ref_arg = oracle.roi_pool(feat[:, chans].cpu().numpy(), rois, (7, 7), 1 / 16)
    got = out[torch.as_tensor(sel, device=dev)][:, torch.as_tensor(chans, device=dev)]
    got_arg = arg[torch.as_tensor(sel, device=dev)][:, torch.as_tensor(chans, device=dev)]
    assert np.array_equal(got_arg.cpu().numpy(), ref_arg)
    assert_bits_equal(got.cpu().numpy(), ref)
    # size-independent property on everything: the value IS the feature at argmax (or 0 / -1 for an empty bin)
    o, a = out[:n], arg[:n]
    bidx = torch.repeat_interleave(torch.arange(B, device=dev), torch.as_tensor(counts, device=dev).long())
    for c in (0, 77, 511):
        plane = feat[bidx, c].reshape(n, -1)
        ac = a[:, c].reshape(n, 49).long()
        picked = torch.gather(plane, 1, ac.clamp(min=0))
        want = torch.where(ac >= 0, picked, torch.zeros_like(picked))
        assert torch.equal(o[:, c].reshape(n, 49), want)


# ------------------------------------------------------------------------------ end to end
def synth_head_outputs(rng, B, fsz, na=9):
    list_cls, list_regs = [], []
    for h, w in fsz:
        n = h * w * na
        cls = np.stack([tie_free_scores(rng, n).reshape(h, w, na).transpose(2, 0, 1) for _ in range(B)])
        reg = rng.standard_normal((B, 4 * na, h, w)).astype(np.float32)
        reg[:, 0::4] *= 0.1; reg[:, 1::4] *= 0.1; reg[:, 2::4] *= 0.2; reg[:, 3::4] *= 0.2
        list_cls.append(np.ascontiguousarray(cls, np.float32) + np.float32(len(list_cls) * 1.7e-4))
        list_regs.append(reg)
    return list_cls, list_regs


class _FixedHead(torch.nn.Module):
    def __init__(self, list_cls, list_regs):
        super().__init__()
        self.c, self.r = list_cls, list_regs

    def forward(self, feats):
        return self.c, self.r


@pytest.mark.parametrize("info,fsz,pre,post", [((2, 600, 1000), [(37, 62)], 12000, 2000),
                                               ((2, 256, 256), [(112, 112), (64, 64), (32, 32)], 2000, 2000),
                                               # A = 200,907 anchors, k = 30,000
                                               ((1, 800, 1344), [(13, 21), (25, 42), (50, 84), (100, 168)], 30000, 2000)])
def test_rpn_forward_vs_oracle(dev, oracle, info, fsz, pre, post):
    """RPN.forward (eval) end to end.  Stages are compared on identical inputs: the oracle's NMS is
    fed the HIP path's decoded boxes, so libm-vs-device exp ULPs cannot flip a keep decision."""
    from objectdetection_amd.RPN import RPN, AnchorGenerator
    from objectdetection_amd import ops
    rng = np.random.default_rng(17)
    list_cls, list_regs = synth_head_outputs(rng, info[0], fsz)
    params = dict(anchor_generator=AnchorGenerator(), rpn_head=_FixedHead([T(c, dev) for c in list_cls],
                                                                         [T(r, dev) for r in list_regs]),
                  num_pre_nms=pre, num_post_nms=post, nms_threshold=0.7, proposal_min_size=1e-3,
                  pos_threshold=0.7, neg_threshold=0.3, num_sample=256, pos_fraction=0.5)
    rpn = RPN(params).eval()
    feats = [torch.zeros(info[0], 1, h, w, device=dev) for h, w in fsz]
    props, loss = rpn(info, feats, None)
    assert loss is None and len(props) == info[0]
    stages = {}
    ref = oracle.rpn_proposals(info, list_cls, list_regs, params, stages=stages)
    # stage-wise on identical inputs
    anchors = torch.cat([a[0].reshape(-1, 4) for a in params['anchor_generator'](info, feats)], 0)
    scores, deltas = ops.rpn_flatten(params['rpn_head'].c, params['rpn_head'].r)
    k = min(pre, scores.shape[1])
    top_val, top_idx = ops.topk_desc(scores, k)
    boxes, fscores, src, count = ops.decode_clip_filter(anchors, deltas, top_idx, top_val, info[1], info[2], 1e-3)
    for b in range(info[0]):
        st = stages["per_image"][b]
        assert np.array_equal(top_idx[b].cpu().numpy(), st["top_idx"])
        c = int(count[b])
        assert np.array_equal(src[b, :c].cpu().numpy(), st["small_index"])
        hip_boxes = boxes[b, :c].cpu().numpy()
        np.testing.assert_allclose(hip_boxes, st["filtered"], rtol=RTOL, atol=1e-4)
        keep_ref = oracle.nms(hip_boxes, fscores[b, :c].cpu().numpy(), 0.7)[:post]
        assert props[b].shape == (len(keep_ref), 4)
        assert_bits_equal(props[b].cpu().numpy(), hip_boxes[keep_ref])
        # and the fully independent oracle run agrees up to the exp tolerance on the same keeps
        if len(ref[b]) == len(keep_ref):
            np.testing.assert_allclose(props[b].cpu().numpy(), ref[b], rtol=RTOL, atol=1e-4)


def test_rpn_training_targets_and_sampling(dev, oracle):
    """RPN.forward training branch: labels/encodings from the fused kernel, sampling on the host
    RNG in the reference's call order (pos randperm first, negatives first in the index)."""
    from objectdetection_amd.RPN import RPN, AnchorGenerator, RPNHead
    info = (2, 256, 256)
    fsz = [(32, 32), (16, 16)]
    targets = [{'size': [3.0, 281, 500], 'class': ['dog'], 'box': [[104, 78, 375, 183], [133, 88, 197, 123]]},
               {'size': [3.0, 500.0, 375.0], 'class': ['cat'], 'box': [[20, 30, 150, 200]]}]
    torch.manual_seed(0)
    params = dict(anchor_generator=AnchorGenerator(), rpn_head=RPNHead(3, 9).to(dev), num_pre_nms=2000,
                  num_post_nms=2000, nms_threshold=0.7, proposal_min_size=1e-3, pos_threshold=0.5,
                  neg_threshold=0.3, num_sample=256, pos_fraction=0.5)
    rpn = RPN(params).train()
    feats = [torch.rand(2, 3, h, w, device=dev) for h, w in fsz]
    torch.manual_seed(123)
    props, loss = rpn(info, feats, targets)
    assert set(loss) == {"cls_loss", "reg_loss"} and torch.isfinite(loss["cls_loss"]) and torch.isfinite(loss["reg_loss"])
    loss["cls_loss"].backward()
    assert rpn.rpn_head.cls.weight.grad is not None
    # labels vs oracle (the sampling itself: test_rpn_training_half_sampling_golden below)
    anchors = oracle.anchors_all_levels(info, fsz)
    labels, regl = rpn.anchor_targets(torch.as_tensor(anchors).to(dev), targets, dev)
    for b, t in enumerate(targets):
        l, a, mx, dv = oracle.iou_label_encode(anchors, np.asarray(t['box'], np.float32), 0.3, 0.5)
        assert_bits_equal(labels[b].cpu().numpy(), l)
        np.testing.assert_allclose(regl[b].cpu().numpy(), dv, rtol=RTOL, atol=1e-6)
        assert int((l == 1).sum()) > 0


class _RecordedHead(torch.nn.Module):
    """Stands in for the (out of scope) conv head: returns the recorded outputs, on the autograd graph."""

    def __init__(self, cls, regs):
        super().__init__()
        self.c = torch.nn.ParameterList([torch.nn.Parameter(c) for c in cls])
        self.r = torch.nn.ParameterList([torch.nn.Parameter(r) for r in regs])

    def forward(self, feats):
        return list(self.c), list(self.r)


def test_rpn_training_half_sampling_golden(dev):
    """RPN.forward(training) against the reference's rpn.py:90-137 replayed with the reference's own utils
    functions under the same torch.manual_seed (tests/golden/make_golden.py: gen_train_sampling): the
    random_index each image actually sampled (per-image positives-then-negatives RNG order, negatives-first
    concat, Q9), the gathered labels and encodings, and both losses."""
    from objectdetection_amd.RPN import RPN, AnchorGenerator
    g = load_golden("train_sampling.npz")
    info = tuple(int(v) for v in g["rpn.info"])
    fsz = [tuple(int(v) for v in hw) for hw in g["rpn.fsz"]]
    S, f, neg, pos = g["rpn.params"]
    head = _RecordedHead([T(g[f"rpn.cls{l}"], dev) for l in range(len(fsz))],
                      [T(g[f"rpn.reg{l}"], dev) for l in range(len(fsz))])
    rpn = RPN(dict(anchor_generator=AnchorGenerator(), rpn_head=head, num_pre_nms=500, num_post_nms=100,
                   nms_threshold=0.7, proposal_min_size=1e-3, pos_threshold=float(pos), neg_threshold=float(neg),
                   num_sample=int(S), pos_fraction=float(f))).train()
    targets = [{"box": g[f"rpn.b{b}.gt"].tolist()} for b in range(info[0])]
    feats = [torch.zeros(info[0], 1, h, w, device=dev) for h, w in fsz]
    torch.manual_seed(int(g["rpn.seed"]))
    _, loss = rpn(info, feats, targets)
    assert len(rpn.last_sampled) == info[0]
    for b, smp in enumerate(rpn.last_sampled):
        n_pos, n_neg = g[f"rpn.b{b}.n_pos_neg"]
        idx = smp["random_index"].cpu().numpy()
        assert np.array_equal(idx, g[f"rpn.b{b}.random_index"]), "image %d sampled other anchors" % b
        assert_bits_equal(smp["cls_label"].cpu().numpy(), g[f"rpn.b{b}.cls_label"])
        np.testing.assert_allclose(smp["reg_label"].cpu().numpy(), g[f"rpn.b{b}.reg_label"], rtol=RTOL, atol=1e-6)
        # negatives first; Q9: a short positive side is not topped up with negatives
        want_pos, want_neg = min(int(S * f), int(n_pos)), min(int(S * (1 - f)), int(n_neg))
        lab = g[f"rpn.b{b}.cls_label"]
        assert len(idx) == want_pos + want_neg and (lab[:want_neg] == 0).all() and (lab[want_neg:] == 1).all()
    np.testing.assert_allclose(float(loss["cls_loss"]), float(g["rpn.cls_loss"]), rtol=1e-6)
    np.testing.assert_allclose(float(loss["reg_loss"]), float(g["rpn.reg_loss"]), rtol=1e-5)
    (loss["cls_loss"] + loss["reg_loss"]).backward()
    assert head.c[0].grad is not None and head.r[1].grad is not None


def test_roi_head_training_half_sampling_golden(dev):
    """ROIHead.forward(force_eval=False, training) against roi_head.py:99-124,141-152 replayed with the
    reference's utils functions under the same seed: sampled index lists, class labels, encodings, sampled
    proposals, and the two losses on recorded predictor outputs."""
    from objectdetection_amd.ROIHead import ROIHead, ROIPooling
    g = load_golden("train_sampling.npz")
    S, f, neg, pos = g["roi.params"]

    class Predictor(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.cls = torch.nn.Parameter(T(g["roi.predict_cls"], dev))
            self.reg = torch.nn.Parameter(T(g["roi.predict_reg"], dev))

        def forward(self, x):
            assert x.shape[0] == self.cls.shape[0]  # one row per sampled proposal, in sampling order
            return self.cls, self.reg

    pred = Predictor()
    head = ROIHead(dict(pos_threshold=float(pos), neg_threshold=float(neg), num_sample=int(S), pos_fraction=float(f),
                        regression_weights=[float(w) for w in g["roi.weights"]], score_thresh=0.05,
                        proposal_min_size=1e-3, nms_threshold=0.5, num_post_nms=100,
                        roi_pooling=ROIPooling(['1'], [7, 7], 2), mlp_head=torch.nn.Identity(), predictor=pred,
                        force_eval=False)).train()
    props = [T(g[f"roi.b{b}.proposals"], dev) for b in range(2)]
    targets = [{"class": g[f"roi.b{b}.class"].tolist(), "box": g[f"roi.b{b}.gt"].tolist()} for b in range(2)]
    feats = [torch.rand(2, 4, 12, 20, device=dev)]
    torch.manual_seed(int(g["roi.seed"]))
    _, loss = head(props, feats, (2, 192, 320), targets, [(192, 320)] * 2)
    for b, smp in enumerate(head.last_sampled):
        assert np.array_equal(smp["random_index"].cpu().numpy(), g[f"roi.b{b}.random_index"])
        assert_bits_equal(smp["cls_label"].cpu().numpy(), g[f"roi.b{b}.cls_label"])
        np.testing.assert_allclose(smp["reg_label"].cpu().numpy(), g[f"roi.b{b}.reg_label"], rtol=RTOL, atol=1e-6)
        assert_bits_equal(smp["proposal"].cpu().numpy(), g[f"roi.b{b}.sampled_proposals"])
    np.testing.assert_allclose(float(loss["cls_loss"]), float(g["roi.cls_loss"]), rtol=1e-6)
    np.testing.assert_allclose(float(loss["reg_loss"]), float(g["roi.reg_loss"]), rtol=1e-5)
    (loss["cls_loss"] + loss["reg_loss"]).backward()
    assert pred.cls.grad is not None and pred.reg.grad is not None


def test_roi_head_eval_and_targets(dev, oracle):
    """ROIHead: proposal targets (batched kernel) vs oracle; eval post-process runs and its NMS
    output satisfies the keep-set property."""
    from objectdetection_amd.ROIHead import ROIHead, ROIPooling, TwoMLPHead, FasterRCNNPredictor
    rng = np.random.default_rng(23)
    info = (2, 256, 256)
    feats = [torch.rand(2, 3, 128, 128, device=dev), torch.rand(2, 3, 64, 64, device=dev)]
    props = [T(realistic_boxes(rng, 64, 256, 256), dev), T(realistic_boxes(rng, 35, 256, 256), dev)]
    targets = [{'size': [3.0, 281, 500], 'class': [1.0, 2.0], 'box': [[104, 78, 175, 183], [33, 88, 197, 123]]},
               {'size': [3.0, 500.0, 375.0], 'class': [2.0], 'box': [[104, 78, 205, 183]]}]
    torch.manual_seed(1)
    params = dict(pos_threshold=0.3, neg_threshold=0.3, num_sample=512, pos_fraction=0.25,
                  regression_weights=[10., 10., 5., 5.], score_thresh=0.05, proposal_min_size=1e-3,
                  nms_threshold=0.5, num_post_nms=1000, roi_pooling=ROIPooling(['1', '2'], [7, 7], 2),
                  mlp_head=TwoMLPHead(3 * 49, 64).to(dev), predictor=FasterRCNNPredictor(64, 3).to(dev))
    head = ROIHead(params)
    tg = head.proposal_targets(props, targets, dev)
    for i in range(2):
        l, a, mx, dv = oracle.iou_label_encode(props[i].cpu().numpy(), np.asarray(targets[i]['box'], np.float32),
                                               0.3, 0.3, np.asarray(targets[i]['class'], np.float32),
                                               [10., 10., 5., 5.])
        assert_bits_equal(tg[i][0].cpu().numpy(), l)
        np.testing.assert_allclose(tg[i][1].cpu().numpy(), dv, rtol=RTOL, atol=1e-6)
    result, loss = head(props, feats, info, targets, [(250, 226), (256, 240)])
    assert loss == {} and len(result) == 2 and not head.training
    for r in result:
        assert r['proposal'].shape[0] == r['score'].shape[0] == r['label'].shape[0]
        assert (r['label'] > 0).all() and (r['score'] > 0.05).all()
        if r['proposal'].shape[0] > 1:
            iou = oracle.box_iou(r['proposal'].cpu().numpy(), r['proposal'].cpu().numpy())
            np.fill_diagonal(iou, 0)
            assert not (iou.astype(np.float64) > 0.5).any()
            s = r['score'].cpu().numpy()
            assert np.all(s[:-1] >= s[1:])
    # training branch as written (force_eval=False)
    params2 = dict(params, force_eval=False)
    head2 = ROIHead(params2).train()
    torch.manual_seed(5)
    _, loss2 = head2(props, feats, info, targets, [(250, 226), (256, 240)])
    assert set(loss2) == {"cls_loss", "reg_loss"}   # (what it sampled: test_roi_head_training_half_sampling_golden)


def test_roi_head_eval_postprocess_vs_oracle(dev, oracle):
    """SURVEY 8f row 1, batched: ROIHead.forward (eval) with fixed predictor outputs against the reference's
    per-image sequence restated with the oracle (roi_head.py:167-229: softmax, per-class box_regression,
    clip to each image's own size, score > thresh, small-box filter, label > 0, class-agnostic NMS, cut)."""
    from objectdetection_amd.ROIHead import ROIHead, ROIPooling
    rng = np.random.default_rng(99)
    K, Cf = 7, 4
    info = (3, 256, 320)
    sizes = [(250, 300), (256, 320), (200, 310)]
    nprops = [150, 1, 333]
    feats = [torch.rand(3, Cf, 32, 40, device=dev)]
    props = [realistic_boxes(rng, n, 320, 256) for n in nprops]
    N = sum(nprops)
    logits = (rng.standard_normal((N, K)) * 2.5).astype(np.float32)
    regs = (rng.standard_normal((N, K * 4)) * 0.3).astype(np.float32)
    regs[5, 10] = 9.0      # above the dw clamp
    regs[7, 8:12] = [0, 0, -40, -40]   # collapses: dropped by the small-box filter

    class Flat(torch.nn.Module):
        def forward(self, x):
            return x.flatten(1)

    class Fixed(torch.nn.Module):
        def forward(self, x):
            return T(logits, dev), T(regs, dev)

    params = dict(pos_threshold=0.5, neg_threshold=0.5, num_sample=128, pos_fraction=0.25,
                  regression_weights=[10., 10., 5., 5.], score_thresh=0.05, proposal_min_size=1e-3,
                  nms_threshold=0.5, num_post_nms=40, roi_pooling=ROIPooling(['1'], [7, 7], 2), mlp_head=Flat(),
                  predictor=Fixed())
    head = ROIHead(params)
    result, loss = head([T(p, dev) for p in props], feats, info, None, sizes)
    assert loss == {} and len(result) == 3
    e = np.exp(logits - logits.max(1, keepdims=True))
    prob = (e / e.sum(1, keepdims=True)).astype(np.float32)
    off = 0
    total_det = 0
    for i, n in enumerate(nprops):
        sc = torch.softmax(torch.as_tensor(logits[off:off + n]), -1).numpy().reshape(-1)     # same fp32 softmax as torch's
        box = oracle.box_clip(oracle.box_regression(props[i], regs[off:off + n]).reshape(-1, 4), sizes[i])
        lab = np.tile(np.arange(K), n)
        m = sc > np.float32(0.05)
        box, s2, lab = box[m], sc[m], lab[m]
        _, idx = oracle.remove_small_box(box, 1e-3)
        box, s2, lab = box[idx], s2[idx], lab[idx]
        m = lab > 0
        box, s2, lab = box[m], s2[m], lab[m]
        got = result[i]
        g_box, g_sc, g_lab = got['proposal'].cpu().numpy(), got['score'].cpu().numpy(), got['label'].cpu().numpy()
        assert got['label'].dtype == torch.int64
        if len(box) == 0:
            assert len(g_sc) == 0
        else:
            keep = oracle.nms(box, s2, 0.5)[:40]
            assert len(g_sc) == len(keep), (i, len(g_sc), len(keep))
            np.testing.assert_allclose(g_sc, s2[keep], rtol=1e-6)
            assert np.array_equal(g_lab, lab[keep])
            np.testing.assert_allclose(g_box, box[keep], rtol=RTOL, atol=1e-4)
            total_det += len(keep)
        off += n
    assert total_det > 20


def test_rpn_forward_reference_file_golden(dev):
    """RPN.forward of the package against what the reference's RPN/rpn.py ITSELF returned when run
    (tests/golden/make_golden.py gen_reference_modules; torchvision.ops pre-seeded with the oracle's nms): the
    proposal lists in eval mode on tie-free two-level inputs, and proposals + both losses in training mode under
    the same torch.manual_seed (rpn.py:46-142 as one call)."""
    from objectdetection_amd.RPN import RPN, AnchorGenerator
    g = load_golden("ref_modules.npz")
    rs = load_golden("rpn_stages.npz")
    info = tuple(int(v) for v in rs["info"])
    fsz = [tuple(int(v) for v in hw) for hw in rs["fsz"]]
    head = _FixedHead([T(rs[f"cls{l}"], dev) for l in range(len(fsz))], [T(rs[f"reg{l}"], dev) for l in range(len(fsz))])
    base = dict(anchor_generator=AnchorGenerator(), nms_threshold=0.7, proposal_min_size=1e-3, pos_threshold=0.7,
                neg_threshold=0.3, num_sample=256, pos_fraction=0.5)
    rpn = RPN(dict(base, rpn_head=head, num_pre_nms=int(rs["num_pre_nms"]), num_post_nms=60)).eval()
    props, loss = rpn(info, [torch.zeros(info[0], 1, h, w, device=dev) for h, w in fsz], None)
    assert loss is None
    for b in range(info[0]):
        want = g[f"rpn_eval.props{b}"]
        assert tuple(props[b].shape) == want.shape and len(want) > 0
        np.testing.assert_allclose(props[b].cpu().numpy(), want, rtol=RTOL, atol=1e-4)
    ts = load_golden("train_sampling.npz")
    info = tuple(int(v) for v in ts["rpn.info"])
    fsz = [tuple(int(v) for v in hw) for hw in ts["rpn.fsz"]]
    S, f, neg, pos = ts["rpn.params"]
    head = _RecordedHead([T(ts[f"rpn.cls{l}"], dev) for l in range(len(fsz))],
                         [T(ts[f"rpn.reg{l}"], dev) for l in range(len(fsz))])
    rpn = RPN(dict(base, rpn_head=head, num_pre_nms=500, num_post_nms=100, pos_threshold=float(pos),
                   neg_threshold=float(neg), num_sample=int(S), pos_fraction=float(f))).train()
    targets = [{"box": ts[f"rpn.b{b}.gt"].tolist()} for b in range(info[0])]
    torch.manual_seed(int(ts["rpn.seed"]))
    props, loss = rpn(info, [torch.zeros(info[0], 1, h, w, device=dev) for h, w in fsz], targets)
    for b in range(info[0]):
        want = g[f"rpn_train.props{b}"]
        assert tuple(props[b].shape) == want.shape and len(want) > 0
        np.testing.assert_allclose(props[b].detach().cpu().numpy(), want, rtol=RTOL, atol=1e-4)
    np.testing.assert_allclose(float(loss["cls_loss"]), float(g["rpn_train.cls_loss"]), rtol=1e-6)
    np.testing.assert_allclose(float(loss["reg_loss"]), float(g["rpn_train.reg_loss"]), rtol=1e-5)


def test_roi_head_eval_reference_file_golden(dev):
    """Row f1 against a fixture the reference's file produced: ROIHead.forward of the package on the recorded
    predictor outputs against the result dicts ROIHead/roi_head.py ITSELF returned when run (gen_reference_modules),
    and the RoI features its ROIPooling (roi_pooling.py:49-54) handed to the head on the two-level pyramid."""
    from objectdetection_amd.ROIHead import ROIHead, ROIPooling
    g = load_golden("ref_modules.npz")
    K = int(g["roi_eval.K"])
    info = tuple(int(v) for v in g["roi_eval.info"])
    sizes = [tuple(int(v) for v in s) for s in g["roi_eval.sizes"]]
    st, ms, thr, post = g["roi_eval.params"]
    seen = {}

    class Flat(torch.nn.Module):
        def forward(self, x):
            seen["roi_features"] = x
            return x.flatten(1)

    class Fixed(torch.nn.Module):
        def forward(self, x):
            return T(g["roi_eval.logits"], dev), T(g["roi_eval.regs"], dev)

    head = ROIHead(dict(pos_threshold=0.5, neg_threshold=0.5, num_sample=128, pos_fraction=0.25,
                        regression_weights=[10., 10., 5., 5.], score_thresh=float(st), proposal_min_size=float(ms),
                        nms_threshold=float(thr), num_post_nms=int(post), roi_pooling=ROIPooling(['1', '2'], [7, 7], 2),
                        mlp_head=Flat(), predictor=Fixed()))
    props = [T(g[f"roi_eval.props{i}"], dev) for i in range(len(sizes))]
    feats = [T(g["roi_eval.feat0"], dev), T(g["roi_eval.feat1"], dev)]
    result, loss = head(props, feats, info, None, sizes)
    assert loss == {} and len(result) == len(sizes)
    np.testing.assert_allclose(seen["roi_features"].cpu().numpy(), g["roi_eval.roi_features"], rtol=RTOL, atol=1e-6)
    for i, r in enumerate(result):
        assert r["label"].dtype == torch.int64
        assert np.array_equal(r["label"].cpu().numpy(), g[f"roi_eval.b{i}.label"])
        np.testing.assert_allclose(r["score"].cpu().numpy(), g[f"roi_eval.b{i}.score"], rtol=1e-6)
        np.testing.assert_allclose(r["proposal"].cpu().numpy(), g[f"roi_eval.b{i}.proposal"], rtol=RTOL, atol=1e-4)


@pytest.mark.parametrize("K,nprops,thresh", [(2, [3000, 1, 0, 2500], -1.0), (21, [400, 0, 37, 1], -1.0),
                                             (21, [2000, 1500], 0.3), (1024, [5, 2], -1.0), (1025, [4, 3], -1.0),
                                             (1, [2100, 5], -1.0)])
def test_roi_detect_filter_counts_and_order(dev, oracle, K, nprops, thresh):
    """The batched filter of roi_head.py:178-209, direct: every class-c>0 candidate kept (threshold below every
    score), so each image fills all the rows it can -- right up to the slots in which the first launch leaves
    its per-tile counts -- and a mid threshold; K = 1025 takes the one-launch kernel, K = 1 keeps nothing."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(K * 7 + len(nprops))
    N = sum(nprops)
    props = realistic_boxes(rng, N, 320, 256) if N else np.zeros((0, 4), np.float32)
    regs = (rng.standard_normal((N, K * 4)) * 0.2).astype(np.float32)
    scores = rng.random((N, K)).astype(np.float32)
    sizes = [(256 - 3 * i, 320 - 5 * i) for i in range(len(nprops))]
    row_off = np.concatenate([[0], np.cumsum(nprops)]).astype(np.int32)
    bx, sc, lb, cnt = ops.roi_detect_filter(T(props, dev), T(regs, dev), T(scores, dev), T(row_off, dev, torch.int32),
                                            T(np.asarray(sizes, np.float32), dev), max(nprops), thresh, 1e-3)
    cnt = cnt.cpu().numpy()
    for i, n in enumerate(nprops):
        lo = row_off[i]
        if n == 0:
            assert cnt[i] == 0
            continue
        box = oracle.box_clip(oracle.box_regression(props[lo:lo + n], regs[lo:lo + n]).reshape(-1, 4), sizes[i])
        s1 = scores[lo:lo + n].reshape(-1)
        lab = np.tile(np.arange(K), n)
        m = (s1 > np.float32(thresh)) & (lab > 0) & (box[:, 2] - box[:, 0] > np.float32(1e-3)) & \
            (box[:, 3] - box[:, 1] > np.float32(1e-3))
        assert cnt[i] == m.sum(), (i, cnt[i], m.sum())
        c = int(cnt[i])
        assert np.array_equal(lb[i, :c].cpu().numpy(), lab[m])
        assert np.array_equal(sc[i, :c].cpu().numpy(), s1[m])
        np.testing.assert_allclose(bx[i, :c].cpu().numpy(), box[m], rtol=RTOL, atol=1e-4)


def test_sampled_losses_golden(dev, oracle):
    """fused gather + loss kernels against what torch / the reference's smooth_l1_loss
    produced (losses.npz), gradients through autograd included, and against the CPU restatement."""
    from objectdetection_amd import ops
    g = load_golden("losses.npz")
    scores = T(g["rpn.scores"], dev).requires_grad_(True)
    regs = T(g["rpn.regs"], dev).requires_grad_(True)
    cls_loss, reg_loss = ops.RpnSampledLoss.apply(scores, regs, T(g["rpn.labels"], dev), T(g["rpn.reg_labels"], dev),
                                                  T(g["rpn.index"], dev, torch.int64))
    np.testing.assert_allclose(cls_loss.item(), g["rpn.cls_loss"], rtol=RTOL)
    np.testing.assert_allclose(reg_loss.item(), g["rpn.reg_loss"], rtol=RTOL)
    up = g["rpn.upstream"]
    (cls_loss * float(up[0]) + reg_loss * float(up[1])).backward()
    np.testing.assert_allclose(scores.grad.cpu().numpy(), g["rpn.grad_scores"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(regs.grad.cpu().numpy(), g["rpn.grad_regs"], rtol=RTOL, atol=1e-8)
    ref_loss, ref_gs, ref_gr = oracle.rpn_sampled_loss(g["rpn.scores"], g["rpn.regs"], g["rpn.labels"],
                                                       g["rpn.reg_labels"], g["rpn.index"])
    np.testing.assert_allclose([cls_loss.item(), reg_loss.item()], ref_loss, rtol=RTOL)
    np.testing.assert_allclose(scores.grad.cpu().numpy(), ref_gs * up[0], rtol=RTOL, atol=1e-8)

    logits = T(g["roi.logits"], dev).requires_grad_(True)
    pred = T(g["roi.pred"], dev).requires_grad_(True)
    ce, rl = ops.RoiLoss.apply(logits, pred, T(g["roi.cls_label"], dev), T(g["roi.reg_label"], dev))
    np.testing.assert_allclose(ce.item(), g["roi.cls_loss"], rtol=RTOL)
    np.testing.assert_allclose(rl.item(), g["roi.reg_loss"], rtol=RTOL)
    up = g["roi.upstream"]
    (ce * float(up[0]) + rl * float(up[1])).backward()
    np.testing.assert_allclose(logits.grad.cpu().numpy(), g["roi.grad_logits"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(pred.grad.cpu().numpy(), g["roi.grad_pred"], rtol=RTOL, atol=1e-8)
    # no positive proposal: the regression mean is over an empty set -> NaN, as in torch; CE stays finite
    loss, _, _ = ops.roi_loss(T(g["roi.logits"], dev), T(g["roi.pred"], dev),
                              torch.zeros(g["roi.cls_label"].shape[0], device=dev), T(g["roi.reg_label"], dev))
    assert torch.isfinite(loss[0]) and torch.isnan(loss[1])
    # an out-of-range sample index poisons the RPN losses instead of reading out of bounds
    bad = T(g["rpn.index"], dev, torch.int64).clone()
    bad[3] = g["rpn.scores"].size
    loss, _, _ = ops.rpn_sampled_loss(T(g["rpn.scores"], dev), T(g["rpn.regs"], dev), T(g["rpn.labels"], dev),
                                      T(g["rpn.reg_labels"], dev), bad)
    assert torch.isnan(loss).all()


def test_generalized_transform_golden(dev):
    """GeneralizedTransform / batched_images against the reference's outputs."""
    from objectdetection_amd.generalized_transform import GeneralizedTransform
    from objectdetection_amd import utils as U
    g = load_golden("transform.npz")
    for name in "abcd":
        mn, mx, n = (int(v) for v in g[f"{name}.sizes"])
        images = [T(g[f"{name}.image{i}"], dev) for i in range(n)]
        targets = [{"box": T(g[f"{name}.box{i}"], dev)} for i in range(n)]
        if int(g[f"{name}.raises"][0]):
            with pytest.raises(ValueError):
                GeneralizedTransform(mn, mx)(images, targets)
            continue
        batch, new_targets = GeneralizedTransform(mn, mx)(images, targets)
        assert batch.is_cuda and tuple(batch.shape) == g[f"{name}.batch"].shape
        np.testing.assert_allclose(batch.cpu().numpy(), g[f"{name}.batch"], rtol=RTOL, atol=5e-6)
        for i in range(n):
            np.testing.assert_allclose(new_targets[i]["box"].cpu().numpy(), g[f"{name}.new_box{i}"], rtol=1e-6)
    batch = U.batched_images([T(g[f"pad.image{i}"], dev) for i in range(3)], 32)
    assert_bits_equal(batch.cpu().numpy(), g["pad.batch"])
    # out=: the same result written into a caller's (dirty) buffer, which is what comes back; a wrong shape raises
    buf = torch.full(tuple(batch.shape), float("nan"), device=dev)
    again = U.batched_images([T(g[f"pad.image{i}"], dev) for i in range(3)], 32, out=buf)
    assert again.data_ptr() == buf.data_ptr()
    assert_bits_equal(buf.cpu().numpy(), g["pad.batch"])
    with pytest.raises(ValueError):
        U.batched_images([T(g[f"pad.image{i}"], dev) for i in range(3)], 32, out=buf[:, :, :-1])


def test_proposal_path_full_size_properties(dev, oracle):
    """BASELINE config 2 at full size (batch 16, A = 20,646, k = 12,000, keep 2,000) on the bench's inputs:
    top-k is sorted and is exactly the k largest; NMS keeps are in candidate order, mutually below the
    threshold, and every earlier candidate that was dropped has a kept suppressor before it (the defining
    properties of greedy NMS); two images are also checked against the oracle outright."""
    import bench
    from objectdetection_amd import ops
    B = 16
    cls, reg, feat = (t.to(dev) for t in bench.make_batch(100, B, "realistic"))
    path = bench.Path(dev)
    anchors = path.anchors(B, feat)
    scores, deltas = ops.rpn_flatten([cls], [reg])
    top_val, top_idx = ops.topk_desc(scores, bench.PRE_NMS)
    # top-k: sorted descending, values are the gathered scores, nothing outside beats the k-th
    assert bool((top_val[:, :-1] >= top_val[:, 1:]).all())
    assert torch.equal(torch.gather(scores, 1, top_idx.long()), top_val)
    kth = top_val[:, -1:]
    assert int((scores > kth).sum()) == int((top_val > kth).sum())
    boxes, fscores, src, count = ops.decode_clip_filter(anchors, deltas, top_idx, top_val, bench.IMG_H, bench.IMG_W,
                                                       bench.MIN_SIZE)
    keep, n_keep = ops.nms_sorted(boxes, count, bench.NMS_THR, bench.POST_NMS)
    for b in range(B):
        c, nk = int(count[b]), int(n_keep[b])
        kp = keep[b, :nk].long()
        assert bool((kp[1:] > kp[:-1]).all()) and int(kp[-1]) < c
        kept = boxes[b, kp]
        iou = ops.box_iou(kept, kept).double()
        iou.fill_diagonal_(0)
        assert not bool((iou > bench.NMS_THR).any())
        if nk < bench.POST_NMS:
            last = c
        else:
            last = int(kp[-1])            # the scan stopped here: candidates after it were never looked at
        dropped = torch.ones(last, dtype=torch.bool, device=dev)
        dropped[kp[kp < last]] = False
        cand = torch.nonzero(dropped)[:, 0]
        if cand.numel():
            ci = ops.box_iou(boxes[b, cand], kept).double()
            earlier = kp[None, :] < cand[:, None]
            assert bool(((ci > bench.NMS_THR) & earlier).any(dim=1).all())
    for b in (3, 11):
        c = int(count[b])
        ref = oracle.nms(boxes[b, :c].cpu().numpy(), fscores[b, :c].cpu().numpy(), bench.NMS_THR)[:bench.POST_NMS]
        assert np.array_equal(keep[b, :int(n_keep[b])].cpu().numpy(), ref)


def test_determinism(dev):
    """Same input twice -> identical bits (no atomics-order dependence anywhere on the path)."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(77)
    boxes = T(realistic_boxes(rng, 6000), dev)
    scores = T(tie_free_scores(rng, 6000), dev)
    a = ops.nms(boxes, scores, 0.6)
    b = ops.nms(boxes, scores, 0.6)
    assert torch.equal(a, b)
    s = T(rng.standard_normal((4, 30000)).astype(np.float32), dev)
    v1, i1 = ops.topk_desc(s, 5000)
    v2, i2 = ops.topk_desc(s, 5000)
    assert torch.equal(i1, i2) and torch.equal(v1, v2)


def test_config5_at_its_stated_batch_of_64(dev, oracle):
    """BASELINE configs[4] at the batch it names (64 images, A = 200,907, 30,000 -> 2,000, four-level pooling with
    C = 256: 6.4 GB of RoI features) -- too large for the CPU oracle as a whole, so: (a) the proposals of an image
    inside the batch of 64 are, bit for bit, what the same image gives alone (whose parity with the oracle
    test_rpn_forward_vs_oracle and test_rpn_proposals_one_call_matches_stages establish at this shape); (b) sampled
    rows of the pooled features against the oracle, for both kernel families; (c) every image fills its 2,000 keeps."""
    import bench
    from objectdetection_amd import ops
    B = 64
    x = bench.config5_inputs(B, dev)
    Himg, Wimg = x["Himg"], x["Wimg"]
    props, n_keep, _ = ops.rpn_proposals_nchw(x["cls"], x["reg"], x["anchors"], 30000, Himg, Wimg, 1e-3, 0.7, 2000)
    assert int(n_keep.min()) == 2000
    picks = (0, 37, 63)
    for b in picks:
        p1, n1, _ = ops.rpn_proposals_nchw([c[b:b + 1].contiguous() for c in x["cls"]],
                                           [r[b:b + 1].contiguous() for r in x["reg"]], x["anchors"], 30000, Himg, Wimg,
                                           1e-3, 0.7, 2000)
        assert int(n1[0]) == int(n_keep[b]) and torch.equal(p1[0], props[b])
    feats = x["feats"][::-1]                                            # fine to coarse
    pool = ops.MultiScaleRoIAlign(['1', '2', '3', '4'], (7, 7), 2)
    fl, scales, k_min, k_max = pool._levels({str(i + 1): f for i, f in enumerate(feats)}, [(Himg, Wimg)])
    cfg = ((7, 7), 2, tuple(float(s_) for s_ in scales), int(k_min), int(k_max))
    rng = np.random.default_rng(64)
    rows = {b: np.sort(rng.choice(2000, 24, replace=False)) for b in picks}
    refs = {}
    for b in picks:
        refs[b], _ = oracle.multiscale_roi_align([f[b:b + 1].cpu().numpy() for f in feats],
                                                 [props[b].cpu().numpy()[rows[b]]], [(Himg, Wimg)], (7, 7), 2)
    for path in ("tiles", "cl"):
        out, total = ops._roi_align_levels(cfg, props, n_keep, fl, path=path)
        assert int(total) == B * 2000
        for b in picks:
            got = out[torch.as_tensor(b * 2000 + rows[b], device=dev)].cpu().numpy()
            np.testing.assert_allclose(got, refs[b], rtol=RTOL, atol=2e-5, err_msg="%s image %d" % (path, b))
        del out


def test_rpn_proposals_from_two_streams_do_not_share_a_workspace(dev):
    """One RPN module called alternately from two streams with different head outputs and no synchronisation in
    between: each stream's calls keep their own intermediates (the one-call proposal half used to keep ONE
    workspace on the module, which calls on different streams overwrote for each other)."""
    from objectdetection_amd.RPN import RPN, AnchorGenerator
    rng = np.random.default_rng(91)
    info, fsz = (2, 600, 1000), [(37, 62)]
    heads = []
    for _ in range(2):
        list_cls, list_regs = synth_head_outputs(rng, info[0], fsz)
        heads.append(([T(c, dev) for c in list_cls], [T(r, dev) for r in list_regs]))
    head = _FixedHead(*heads[0])
    params = dict(anchor_generator=AnchorGenerator(), rpn_head=head, num_pre_nms=12000, num_post_nms=2000,
                  nms_threshold=0.7, proposal_min_size=1e-3, pos_threshold=0.7, neg_threshold=0.3, num_sample=256,
                  pos_fraction=0.5)
    rpn = RPN(params).eval()
    feats = [torch.zeros(info[0], 1, h, w, device=dev) for h, w in fsz]
    ref = []
    for c, r in heads:
        head.c, head.r = c, r
        p_, n_ = rpn.proposals_padded(info, feats)
        torch.cuda.synchronize()
        ref.append((p_.clone(), n_.clone()))
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    got = []
    for it in range(6):
        i = it & 1
        head.c, head.r = heads[i]
        with torch.cuda.stream(streams[i]):
            got.append((i,) + tuple(rpn.proposals_padded(info, feats)))
    torch.cuda.synchronize()
    assert len(rpn._path_ws) >= 3  # the default stream's and one per side stream
    for i, p_, n_ in got:
        assert torch.equal(n_, ref[i][1]) and torch.equal(p_, ref[i][0])


def test_two_stream_pipeline_matches_serial(dev):
    """ProposalRoIPipeline (two batches in flight on two streams) returns, bit for bit, what the same
    stages give when run one after the other on one stream -- over consecutive, different batches, so
    that a missing event or an early buffer reuse would show."""
    from objectdetection_amd import ops
    from objectdetection_amd.pipeline import ProposalRoIPipeline
    from objectdetection_amd.RPN.anchor_generator import AnchorGenerator
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    B, Himg, Wimg, fh, fw, C = 3, 256, 320, 16, 20, 24
    rng = np.random.default_rng(5)
    ag, pool = AnchorGenerator(), ROIPooling(['1'], [7, 7], 2)
    batches = []
    for _ in range(4):
        list_cls, list_regs = synth_head_outputs(rng, B, [(fh, fw)])
        feat = T(rng.standard_normal((B, C, fh, fw)).astype(np.float32), dev)
        batches.append((T(list_cls[0], dev), T(list_regs[0], dev), feat))
    anchors = torch.cat([a[0].reshape(-1, 4) for a in ag((B, Himg, Wimg), [batches[0][2]])], 0)

    def proposals(cls, reg):
        scores, deltas = ops.rpn_flatten([cls], [reg])
        top_val, top_idx = ops.topk_desc(scores, 1000)
        boxes, _, _, count = ops.decode_clip_filter(anchors, deltas, top_idx, top_val, Himg, Wimg, 1e-3)
        keep, n_keep = ops.nms_sorted(boxes, count, 0.7, 300)
        return ops.gather_boxes(boxes, keep, n_keep), n_keep

    serial = []
    for cls, reg, feat in batches:
        props, n_keep = proposals(cls, reg)
        out, total = pool.forward_padded({'1': feat}, props, n_keep, [(Himg, Wimg)])
        torch.cuda.synchronize()
        serial.append((props.cpu().numpy(), n_keep.cpu().numpy(), out[:int(total)].cpu().numpy()))

    pipe = ProposalRoIPipeline(dev)
    pipe.wait_for_current()
    got = []
    for cls, reg, feat in batches:  # no synchronisation between submits
        got.append(pipe.submit(lambda cls=cls, reg=reg: proposals(cls, reg),
                               lambda p, n, feat=feat: pool.forward_padded({'1': feat}, p, n, [(Himg, Wimg)])))
    pipe.synchronize()
    for (props, n_keep, (out, total)), (rp, rn, ro) in zip(got, serial):
        assert np.array_equal(n_keep.cpu().numpy(), rn)
        assert_bits_equal(props.cpu().numpy(), rp)
        assert int(total) == ro.shape[0] and ro.shape[0] > 0
        assert_bits_equal(out[:int(total)].cpu().numpy(), ro)


# ---- closed-form vectors for the torchvision-owned ops (tests/golden/make_closed_form.py: arithmetic only; the
# ---- only pin for them that does not go through this repository's own C restatement) ----------------------------
def test_closed_form_roi_align_single_level(dev):
    from objectdetection_amd import ops
    g = load_golden("closed_form.npz")
    pool = ops.MultiScaleRoIAlign(['0'], (7, 7), 2)
    boxes = [T(g["ra1.boxes0"], dev), T(g["ra1.boxes1"], dev)]
    out = pool({'0': T(g["ra1.feat"], dev)}, boxes, [tuple(int(v) for v in g["ra1.image_hw"])])
    np.testing.assert_allclose(out.cpu().numpy(), g["ra1.expect"], rtol=RTOL, atol=2e-5)
    # the (K,5) roi-list entry point as well
    rois = torch.cat([torch.cat([torch.full((b.shape[0], 1), float(i), device=dev), b], 1) for i, b in enumerate(boxes)])
    out2 = ops.roi_align(T(g["ra1.feat"], dev), rois, (7, 7), 1.0 / 16, 2)
    np.testing.assert_allclose(out2.cpu().numpy(), g["ra1.expect"], rtol=RTOL, atol=2e-5)


def test_closed_form_roi_align_levels(dev):
    from objectdetection_amd import ops
    g = load_golden("closed_form.npz")
    pool = ops.MultiScaleRoIAlign(['a', 'b', 'c', 'd'], (7, 7), 2)
    feats = {n: T(g[f"ra4.feat{l}"], dev) for l, n in enumerate("abcd")}
    boxes = [T(g["ra4.boxes0"], dev), T(g["ra4.boxes1"], dev)]
    out = pool(feats, boxes, [tuple(int(v) for v in g["ra4.image_hw"])])
    np.testing.assert_allclose(out.cpu().numpy(), g["ra4.expect"], rtol=RTOL, atol=2e-5)
    allb = torch.cat(boxes)
    rois = torch.cat([torch.zeros((allb.shape[0], 1), device=dev), allb], 1)
    lvl = ops.level_map(rois, 2, 5).cpu().numpy()
    assert np.array_equal(lvl, np.concatenate([g["ra4.levels0"], g["ra4.levels1"]]))


def test_closed_form_nms(dev):
    from objectdetection_amd import ops
    g = load_golden("closed_form.npz")
    for name in g["nms.names"]:
        keep = ops.nms(T(g[f"nms.{name}.boxes"], dev), T(g[f"nms.{name}.scores"], dev), float(g[f"nms.{name}.thr"]))
        assert keep.dtype == torch.int64
        assert np.array_equal(keep.cpu().numpy(), g[f"nms.{name}.keep"]), name
    for i, thr in enumerate(g["nms.lattice.thr"]):   # 9,000 boxes: the round path
        keep = ops.nms(T(g["nms.lattice.boxes"], dev), T(g["nms.lattice.scores"], dev), float(thr))
        assert np.array_equal(keep.cpu().numpy(), g[f"nms.lattice.keep{i}"])


def test_closed_form_roi_pool(dev):
    from objectdetection_amd import ops
    g = load_golden("closed_form.npz")
    out, arg = ops.roi_pool(T(g["rp.feat"], dev), T(g["rp.rois"], dev), (7, 7), float(g["rp.scale"]))
    assert np.array_equal(arg.cpu().numpy(), g["rp.argmax"])
    assert_bits_equal(out.cpu().numpy(), g["rp.expect"])
    # and the batched plane-scheme kernel (one image, all RoIs)
    out, arg, total = ops.roi_pool_batched(T(g["rp.feat"], dev), T(g["rp.rois"][None, :, 1:], dev), None, (7, 7),
                                           float(g["rp.scale"]))
    assert int(total) == g["rp.rois"].shape[0]
    assert np.array_equal(arg.cpu().numpy(), g["rp.argmax"])
    assert_bits_equal(out.cpu().numpy(), g["rp.expect"])


# ---- RoIAlign backward (ROIHead/roi_head.py:127-130 under autograd: the RoI losses train the backbone) -----------
@pytest.mark.parametrize("case", ["headline_map", "tail_channels_small_batch", "pyramid_with_oversize_plane",
                                  "one_workgroup_per_chunk"])
def test_roi_align_backward_vs_oracle(dev, oracle, case):
    """d(sum(out * G)) / d(features) through MultiScaleRoIAlign: the HIP backward (LDS-atomic plane kernel, the
    RoI-split form for small batches, the global-atomic fallback for planes beyond LDS) against the oracle's
    restatement of torchvision's roi_align_backward.  Sums are accumulated in a different order: 1e-5 relative
    to the plane's scale."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(abs(hash(case)) % 1000)
    if case == "headline_map":
        B, C, img, fsz, n = 3, 20, (600, 1000), [(37, 62)], [300, 1, 120]
    elif case == "tail_channels_small_batch":
        B, C, img, fsz, n = 1, 7, (256, 320), [(32, 40)], [64]
    elif case == "one_workgroup_per_chunk":
        # B * chunks >= 256: no RoI split, plain stores; hundreds of RoIs per workgroup = several table batches
        B, C, img, fsz, n = 8, 515, (384, 640), [(24, 40)], [260, 3, 150, 200, 64, 65, 129, 300]
    else:
        B, C, img, fsz, n = 2, 3, (800, 1344), [(200, 336), (100, 168), (50, 84), (25, 42)], [150, 90]
    names = [str(i) for i in range(len(fsz))]
    feats = [torch.randn(B, C, h, w, device=dev, requires_grad=True) for h, w in fsz]
    boxes = [realistic_boxes(rng, k, img[1], img[0]) for k in n]
    boxes[0][0] = [-50.0, -40.0, 90.0, 70.0]             # starts outside the map
    boxes[0][-1] = [img[1] - 30.0, img[0] - 20.0, img[1] + 60.0, img[0] + 80.0]   # runs off it
    pool = ops.MultiScaleRoIAlign(names, (7, 7), 2)
    out = pool(dict(zip(names, feats)), [T(b, dev) for b in boxes], [img])
    assert out.requires_grad and out.shape == (sum(n), C, 7, 7)
    G = torch.randn(out.shape, device=dev)
    out.backward(G)
    ref = oracle.multiscale_roi_align_backward(G.cpu().numpy(), [tuple(f.shape) for f in feats], boxes, [img])
    for f, r in zip(feats, ref):
        assert f.grad is not None and f.grad.shape == f.shape
        np.testing.assert_allclose(f.grad.cpu().numpy(), r, rtol=RTOL, atol=1e-5 * max(1.0, float(np.abs(r).max())))
    assert sum(float(np.abs(r).sum()) > 0 for r in ref) >= min(2, len(ref))


def test_roi_align_backward_adjoint_of_closed_form(dev):
    """<RoIAlign(f), G> = <f, RoIAlign^T(G)> with the affine planes of closed_form.npz, whose forward values are known
    by arithmetic: pins the HIP backward without the oracle."""
    from objectdetection_amd import ops
    g = load_golden("closed_form.npz")
    rng = np.random.default_rng(5)
    feats = {n: T(g[f"ra4.feat{l}"], dev).requires_grad_(True) for l, n in enumerate("abcd")}
    out = ops.MultiScaleRoIAlign(list("abcd"), (7, 7), 2)(feats, [T(g["ra4.boxes0"], dev), T(g["ra4.boxes1"], dev)],
                                                          [tuple(int(v) for v in g["ra4.image_hw"])])
    G = rng.standard_normal(g["ra4.expect"].shape).astype(np.float32)
    out.backward(T(G, dev))
    lvl = np.concatenate([g["ra4.levels0"], g["ra4.levels1"]])
    for l, n in enumerate("abcd"):
        lhs = (g["ra4.expect"].astype(np.float64) * G)[lvl == l].sum(axis=(0, 2, 3))
        rhs = (g[f"ra4.feat{l}"].astype(np.float64) * feats[n].grad.cpu().numpy()).sum(axis=(0, 2, 3))
        np.testing.assert_allclose(rhs, lhs, rtol=1e-5, atol=1e-3)


def test_roi_align_backward_full_batch_adjoint_and_determinism(dev):
    """Batch 16 x 2,000 RoIs on the headline map (256 channels = 256 workgroups: one per (image, chunk), no RoI split,
    ~90 table batches each).
    <RoIAlign(f), G> = <f, RoIAlign^T(G)> per (image, channel) against the forward plane kernel (whose values the
    oracle and the closed forms pin), and the same bits run to run (no atomics on this form)."""
    from objectdetection_amd import ops
    rng = np.random.default_rng(17)
    B, P, C, H, W = 16, 2000, 256, 37, 62
    props = T(np.stack([realistic_boxes(rng, P, 1000, 600) for _ in range(B)]), dev)
    cnt = T(rng.integers(1500, P + 1, B).astype(np.int32), dev, torch.int32)
    cfg = ((7, 7), 2, (1.0 / 16,), 4, 4)
    torch.manual_seed(3)
    f = torch.randn((B, C, H, W), device=dev)
    y, total = ops._roi_align_levels(cfg, props, cnt, [f])
    K = int(total)
    assert K == int(cnt.sum())
    G = torch.randn((K, C, 7, 7), device=dev)
    g1 = ops.roi_align_levels_backward(cfg, props, cnt, G, [(B, C, H, W)])[0]
    g2 = ops.roi_align_levels_backward(cfg, props, cnt, G, [(B, C, H, W)])[0]
    assert torch.equal(g1, g2)
    img = torch.repeat_interleave(torch.arange(B, device=dev), cnt.long())
    terms = (y[:K].double() * G.double()).sum(dim=(2, 3))
    lhs = torch.zeros((B, C), dtype=torch.float64, device=dev).index_add_(0, img, terms)
    scale = torch.zeros((B, C), dtype=torch.float64, device=dev).index_add_(
        0, img, (y[:K].double().abs() * G.double().abs()).sum(dim=(2, 3)))
    rhs = (f.double() * g1.double()).sum(dim=(2, 3))
    assert float(((lhs - rhs).abs() / scale).max()) < 1e-6


def test_roi_head_training_sends_gradients_to_the_feature_maps(dev):
    """ROIHead(force_eval=False).train(): the RoI losses reach the feature maps through the pooling, as they do in the
    reference (torchvision's op is differentiable); under no_grad / for maps that need none, nothing is recorded."""
    from objectdetection_amd.ROIHead import ROIHead, ROIPooling, TwoMLPHead, FasterRCNNPredictor
    rng = np.random.default_rng(3)
    info = (2, 256, 256)
    feats = [torch.rand(2, 3, 128, 128, device=dev, requires_grad=True), torch.rand(2, 3, 64, 64, device=dev)]
    props = [T(realistic_boxes(rng, 64, 256, 256), dev), T(realistic_boxes(rng, 35, 256, 256), dev)]
    targets = [{'class': [1.0, 2.0], 'box': [[104, 78, 175, 183], [33, 88, 197, 123]]},
               {'class': [2.0], 'box': [[104, 78, 205, 183]]}]
    torch.manual_seed(1)
    head = ROIHead(dict(pos_threshold=0.3, neg_threshold=0.3, num_sample=512, pos_fraction=0.25,
                        regression_weights=[10., 10., 5., 5.], score_thresh=0.05, proposal_min_size=1e-3,
                        nms_threshold=0.5, num_post_nms=1000, roi_pooling=ROIPooling(['1', '2'], [7, 7], 2),
                        mlp_head=TwoMLPHead(3 * 49, 64).to(dev), predictor=FasterRCNNPredictor(64, 3).to(dev),
                        force_eval=False)).train()
    _, loss = head(props, feats, info, targets, [(250, 226), (256, 240)])
    (loss["cls_loss"] + loss["reg_loss"]).backward()
    assert feats[0].grad is not None and float(feats[0].grad.abs().sum()) > 0
    assert feats[1].grad is None
    with torch.no_grad():
        out = head.roi_pooling({'1': feats[0], '2': feats[1]}, props, [(256, 256)])
    assert not out.requires_grad


def test_randomised_shapes_against_the_oracle(dev, oracle):
    """A fixed slice of tools/fuzz_parity.py's draws (shapes, counts, data kinds and bounds drawn at random rather
    than chosen): every op of the path, 20 draws each.  The tool itself runs the same draws for minutes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    for name, fn in fz.OPS.items():
        for i in range(20):
            seed = 424243 + 1000 * i + len(name)
            try:
                fn(np.random.default_rng(seed))
            except AssertionError as e:
                raise AssertionError(f"{name}, seed {seed}: {e}") from e


@pytest.mark.parametrize("B", [8, 11, 17])
def test_roi_align_window_kernel_whole_images_per_xcd(dev, oracle, B):
    """Batches of 8 images or more: the window kernel hands whole images to an XCD (below that whole window rows).
    Image counts that are and are not multiples of 8, ragged RoI counts, an image without RoIs."""
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    rng = np.random.default_rng(800 + B)
    C, H, W, stride = 12, 100, 168, 8
    Himg, Wimg = H * stride, W * stride
    feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
    counts = [int(rng.integers(1, 60)) for _ in range(B)]
    counts[B // 2] = 0
    props = []
    for n in counts:
        s = rng.uniform(8, 240, (n, 2))
        c = rng.uniform(0, 1, (n, 2)) * [Wimg, Himg]
        bx = np.concatenate([c - s / 2, c + s / 2], 1)
        bx[:, 0::2] = bx[:, 0::2].clip(0, Wimg)
        bx[:, 1::2] = bx[:, 1::2].clip(0, Himg)
        props.append(bx.astype(np.float32))
    ref, _ = oracle.multiscale_roi_align([feat], props, [(Himg, Wimg)], (7, 7), 2)
    P = max(counts)
    padded = np.zeros((B, P, 4), np.float32)
    for b, p in enumerate(props):
        padded[b, :len(p)] = p
    pool = ROIPooling(['1'], [7, 7], 2)
    out, total = pool.forward_padded({'1': T(feat, dev)}, T(padded, dev), T(np.asarray(counts, np.int32), dev, torch.int32),
                                     [(Himg, Wimg)])
    assert int(total) == sum(counts)
    np.testing.assert_allclose(out[:sum(counts)].cpu().numpy(), ref, rtol=RTOL, atol=1e-5)
