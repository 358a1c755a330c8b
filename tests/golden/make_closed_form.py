"""Closed-form vectors for the operations the reference takes from torchvision (nms, MultiScaleRoIAlign /
roi_align / the FPN level mapper, and the north-star max roi_pool).

torchvision is not installed here and the reference holds no fixture for these, so the C oracle's restatement
of them has nothing of the reference's to be pinned against.  This script is the independent pin that IS
available: inputs whose correct outputs follow from the published definitions by ARITHMETIC ALONE --
it imports neither the oracle, nor the product, nor the reference, only numpy (float64 / integers):

  * RoIAlign of an affine plane f(x, y) = a x + b y + c.  Bilinear interpolation reproduces an affine function
    exactly, whichever cell a sample falls into, so a sample at (x, y) is worth f(clamp(x, 0, W-1),
    clamp(y, 0, H-1)), or 0 when x < -1, x > W, y < -1 or y > H (roi_align's border rules), and a bin is the
    mean of its 2 x 2 samples at start + p * bin + (i + 0.5) * bin / 2, bin = max(extent * scale, 1) / 7.
  * Level mapper: floor(4 + log2(sqrt(area) / 224) + 1e-6), clamped to [k_min, k_max], on square boxes whose
    side is exactly 112 / 224 / 448 (the level boundaries), one fp32 ulp either side of them (the 1e-6 decides:
    same level as the boundary) and 1 % either side.
  * NMS: a handful of hand-worked cases -- every expected keep list below is written out by hand, with the
    IoU arithmetic in the comment beside it.
  * max roi_pool: planes on which the maximum of a bin is known without searching (one pixel per bin; a
    constant plane: the first element wins the tie; f = y * W + x: the last element of the bin), with the
    bin edges floor(p * bin) / ceil((p + 1) * bin) computed in integers.

Run:  python tests/golden/make_closed_form.py     (writes closed_form.npz next to this file)
"""
import os
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GUARD = 1e-3  # no sample of a generated case lies this close to a discontinuity of roi_align's border rules


# ------------------------------------------------------------------------------------------ RoIAlign
def affine_plane(coef, H, W):
    """(C,3) coefficients (a, b, c) -> (C,H,W) fp32 planes; coefficients are multiples of 1/8 and the maps are
    small, so every pixel value is exact in fp32."""
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    p = coef[:, 0, None, None] * x + coef[:, 1, None, None] * y + coef[:, 2, None, None]
    assert np.array_equal(p.astype(np.float32).astype(np.float64), p)
    return p.astype(np.float32)


def roi_align_affine(coef, H, W, box, scale):
    """Expected (C,7,7) of one RoI on planes with coefficients coef (float64 arithmetic of the definition)."""
    box = np.asarray(box, np.float32).astype(np.float64)  # what the kernels are given
    x1, y1, x2, y2 = box * scale
    roi_w, roi_h = max(x2 - x1, 1.0), max(y2 - y1, 1.0)
    bw, bh = roi_w / 7.0, roi_h / 7.0
    out = np.zeros((coef.shape[0], 7, 7))
    for ph in range(7):
        for pw in range(7):
            acc = np.zeros(coef.shape[0])
            for iy in range(2):
                for ix in range(2):
                    y = y1 + ph * bh + (iy + 0.5) * bh / 2.0
                    x = x1 + pw * bw + (ix + 0.5) * bw / 2.0
                    for v, n in ((x, W), (y, H)):
                        assert abs(v + 1.0) > GUARD and abs(v - n) > GUARD, "sample on a border rule's edge"
                    if y < -1.0 or y > H or x < -1.0 or x > W:
                        continue
                    xc, yc = min(max(x, 0.0), W - 1.0), min(max(y, 0.0), H - 1.0)
                    acc += coef[:, 0] * xc + coef[:, 1] * yc + coef[:, 2]
            out[:, ph, pw] = acc / 4.0
    return out


def level_of(box, k_min, k_max):
    """LevelMapper in float64 on the fp32 box; asserts that fp32 evaluation cannot land on another integer
    unless the 1e-6 epsilon alone decides (boundary and +-1 ulp cases: the generator names those)."""
    b = np.asarray(box, np.float32).astype(np.float64)
    s = np.sqrt((b[2] - b[0]) * (b[3] - b[1]))
    t = 4.0 + np.log2(s / 224.0) + 1e-6
    return int(min(max(np.floor(t), k_min), k_max)) - k_min, t - np.floor(t)


def gen_roi_align(out):
    # ---- single level: the headline geometry (37 x 62 map of a 600 x 1000 image, scale 1/16), 2 images
    H, W, scale = 37, 62, 1.0 / 16
    coef = np.array([[0.5, 0.25, 1.0], [-0.125, 0.75, -3.0], [0.0, 0.0, 2.5], [1.0, -1.0, 0.0], [0.375, 0.0, 8.0]])
    feat = np.stack([affine_plane(coef, H, W), affine_plane(coef * np.array([1, 1, -1.0]), H, W)])  # image 1: c -> -c
    boxes = [
        [[100.0, 80.0, 420.0, 300.0],          # interior
         [0.0, 0.0, 1000.0, 600.0],            # the whole image: last samples beyond W-1 / H-1 are clamped
         [-40.0, -30.0, 200.0, 180.0],         # starts outside (samples in [-1, 0) clamp to 0, beyond -1 give 0)
         [900.0, 500.0, 1100.0, 700.0],        # ends outside the map: samples beyond W, H contribute zeros
         [1200.0, 700.0, 1300.0, 800.0],       # entirely outside: all zeros
         [333.0, 222.0, 335.0, 223.0],         # smaller than a feature pixel: roi size clamps to 1
         [500.0, 100.0, 500.0, 100.0]],        # empty box: the same clamp
        [[16.0, 16.0, 240.0, 130.0],
         [700.5, 33.25, 977.75, 590.0]],
    ]
    expect = []
    for b, bl in enumerate(boxes):
        cf = coef * (np.array([1, 1, -1.0]) if b == 1 else 1.0)
        expect += [roi_align_affine(cf, H, W, bx, scale) for bx in bl]
    out["ra1.feat"] = feat
    out["ra1.image_hw"] = np.asarray([600, 1000])
    for b, bl in enumerate(boxes):
        out[f"ra1.boxes{b}"] = np.asarray(bl, np.float32)
    out["ra1.expect"] = np.stack(expect).astype(np.float32)
    assert np.abs(out["ra1.expect"][4]).max() == 0 and np.abs(out["ra1.expect"][3]).max() > 0

    # ---- four levels (448 x 448 image; maps 112, 56, 28, 14 = scales 1/4 .. 1/32, k_min 2, k_max 5); every level
    # has its own coefficients, so a RoI pooled from the wrong level cannot give the expected values
    img = 448
    sizes = [112, 56, 28, 14]
    lcoef = [np.array([[0.25, 0.5, 1.0], [1.0, 0.0, -2.0]]) * (l + 1) for l in range(4)]
    feats = [np.stack([affine_plane(lcoef[l], n, n), affine_plane(lcoef[l] + np.array([0, 0, 0.5]), n, n)])
             for l, n in enumerate(sizes)]
    up, dn = lambda v: float(np.nextafter(np.float32(v), np.float32(1e9))), lambda v: float(np.nextafter(np.float32(v), np.float32(0)))
    sides = [(40.0, "clear"), (111.0, "clear"), (dn(112), "eps"), (112.0, "eps"), (up(112), "eps"), (113.0, "clear"),
             (221.76, "clear"), (dn(224), "eps"), (224.0, "eps"), (up(224), "eps"), (226.24, "clear"),
             (443.5, "clear"), (dn(448), "eps"), (448.0, "eps"), (up(448), "eps"), (452.5, "clear"),
             (3.0, "clear"), (1790.0, "clear")]   # far below k_min / above k_max: clamped
    boxes, levels = [[], []], [[], []]
    for i, (side, kind) in enumerate(sides):
        b = i % 2
        x0, y0 = (8.0, 4.0) if side < 400 else (-2.0, -1.0)
        box = [x0, y0, x0 + side, y0 + side]
        side32 = float(np.float32(box[2])) - float(np.float32(box[0]))
        lvl, frac = level_of(box, 2, 5)
        if kind == "clear":   # fp32 evaluation errs by ~1e-6 at most: nowhere near an integer
            assert 1e-3 < frac < 1 - 1e-3, (side, frac)
        else:                 # within an ulp of a boundary s0 * 2^j: the epsilon puts it in the boundary's level
            assert abs(frac - 1e-6) < 3e-7 and abs(side32 / side - 1) < 1e-6, (side, frac)
        boxes[b].append(box)
        levels[b].append(lvl)
    expect = []
    for b in range(2):
        for box, lvl in zip(boxes[b], levels[b]):
            expect.append(roi_align_affine(lcoef[lvl] + (np.array([0, 0, 0.5]) if b else 0.0), sizes[lvl], sizes[lvl],
                                           box, 1.0 / (4 << lvl)))
    for l in range(4):
        out[f"ra4.feat{l}"] = feats[l]
    out["ra4.image_hw"] = np.asarray([img, img])
    for b in range(2):
        out[f"ra4.boxes{b}"] = np.asarray(boxes[b], np.float32)
        out[f"ra4.levels{b}"] = np.asarray(levels[b], np.int32)
    out["ra4.expect"] = np.stack(expect).astype(np.float32)
    assert sorted(set(levels[0] + levels[1])) == [0, 1, 2, 3]


# ----------------------------------------------------------------------------------------------- NMS
def gen_nms(out):
    nan = float("nan")
    cases = {
        # IoU([0,0,10,10], [0,0,10,5]) = 50 / (100 + 50 - 50) = 0.5 exactly: suppression needs iou > thr (strict)
        "exact_thr_keeps": ([[0, 0, 10, 10], [0, 0, 10, 5]], [0.9, 0.8], 0.5, [0, 1]),
        "just_below_thr_suppresses": ([[0, 0, 10, 10], [0, 0, 10, 5]], [0.9, 0.8], 0.4999, [0]),
        # candidates are visited by descending score: box 1 (score .9) first; IoU = 0.5 > 0.3
        "score_order": ([[0, 0, 10, 5], [0, 0, 10, 10]], [0.8, 0.9], 0.3, [1]),
        # three identical boxes: IoU = 1 > 0.7, only the best-scored survives
        "duplicates": ([[1, 1, 5, 5], [1, 1, 5, 5], [1, 1, 5, 5]], [0.3, 0.5, 0.4], 0.7, [1]),
        # equal scores: the adopted rule is a stable sort, the lower index goes first (and suppresses: IoU = 1)
        "tied_scores_lower_index_first": ([[1, 1, 5, 5], [1, 1, 5, 5], [20, 20, 30, 30]], [0.5, 0.5, 0.5], 0.7, [0, 2]),
        # greedy chain, thr 0.4: A=[0,0,10,10], B=[4,0,14,10], C=[8,0,18,10].  IoU(A,B) = 60/140 = 0.4286 > 0.4:
        # B goes.  IoU(A,C) = 20/180 = 0.11.  IoU(B,C) = 0.4286 but B is not kept, so C stays.
        "greedy_chain": ([[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10]], [0.9, 0.8, 0.7], 0.4, [0, 2]),
        # a NaN coordinate makes area and IoU NaN; NaN > thr is false: never suppressed, suppresses nothing
        "nan_box": ([[0, 0, 10, 10], [nan, 0, 10, 10], [0, 0, 10, 9]], [0.9, 0.8, 0.7], 0.5, [0, 1]),
        # two zero-area boxes at the same point: inter 0, union 0, 0/0 = NaN: both stay
        "degenerate_pair": ([[5, 5, 5, 5], [5, 5, 5, 5]], [0.6, 0.5], 0.1, [0, 1]),
        # disjoint boxes: everything is kept, in score order
        "disjoint": ([[0, 0, 4, 4], [10, 10, 14, 14], [20, 0, 24, 4], [0, 20, 4, 24]], [0.1, 0.4, 0.3, 0.2], 0.7,
                     [1, 2, 3, 0]),
        # thr 0: any positive overlap suppresses; touching boxes (inter 0 -> IoU 0, not > 0) do not
        "thr_zero": ([[0, 0, 10, 10], [10, 0, 20, 10], [9, 0, 19, 10]], [0.9, 0.8, 0.7], 0.0, [0, 1]),
        # a box inside a larger one: IoU = 16/100 = 0.16 <= 0.7, both stay; negative scores order like any other
        "nested": ([[0, 0, 10, 10], [3, 3, 7, 7]], [-2.0, -1.0], 0.7, [1, 0]),
        "single": ([[0, 0, 1, 1]], [0.0], 0.7, [0]),
    }
    out["nms.names"] = np.asarray(sorted(cases))
    for name, (boxes, scores, thr, keep) in cases.items():
        out[f"nms.{name}.boxes"] = np.asarray(boxes, np.float32).reshape(-1, 4)
        out[f"nms.{name}.scores"] = np.asarray(scores, np.float32)
        out[f"nms.{name}.thr"] = np.asarray(thr, np.float64)
        out[f"nms.{name}.keep"] = np.asarray(keep, np.int64)
    # a larger, still closed-form case: a lattice of 60 x 50 disjoint 4 x 4 boxes, each with two shifted copies
    # (shift 1 px: IoU = 12/20 = 0.6; shift 2 px: IoU = 8/24 = 0.333; the two copies: 0.6).  Scores: base >
    # copy 1 > copy 2 within a cell, cells in a scrambled order.  thr 0.5: base kills copy 1 (0.6), copy 2
    # survives the base (0.333) and copy 1 is gone, so keeps = base and copy 2 of every cell = 6,000 of 9,000.
    # thr 0.3: base kills both = 3,000 keeps.  Keep order = descending score.
    nx, ny = 60, 50
    cell = np.arange(nx * ny)
    order = (cell * 7919) % (nx * ny)  # a permutation (7919 is prime to 3000)
    assert len(set(order.tolist())) == nx * ny
    x0, y0 = (cell % nx) * 8.0, (cell // nx) * 8.0
    base = np.stack([x0, y0, x0 + 4, y0 + 4], 1)
    boxes = np.concatenate([base, base + [1, 0, 1, 0], base + [2, 0, 2, 0]]).astype(np.float32)
    score = np.concatenate([3.0 - order / 4096.0 * 0.25, 2.0 - order / 4096.0 * 0.25, 1.0 - order / 4096.0 * 0.25])
    assert len(set(score.astype(np.float32).tolist())) == 3 * nx * ny
    by_score = np.argsort(order)  # cells from best to worst
    n = nx * ny
    out["nms.lattice.boxes"] = boxes
    out["nms.lattice.scores"] = score.astype(np.float32)
    out["nms.lattice.thr"] = np.asarray([0.5, 0.3])
    out["nms.lattice.keep0"] = np.concatenate([by_score, by_score + 2 * n]).astype(np.int64)
    out["nms.lattice.keep1"] = by_score.astype(np.int64)


# ------------------------------------------------------------------------------------------ max roi_pool
def c_round(v):
    """C round(): halves away from zero (what roi_pool applies to coord * scale)."""
    f = Fraction(v)
    return int(f + Fraction(1, 2)) if f >= 0 else -int(-f + Fraction(1, 2))


def roi_pool_bins(box, scale, H, W):
    """Integer bin edges of torchvision roi_pool: start = round(coord * scale), extent = max(end - start + 1, 1),
    bin = extent / 7, [floor(p * bin), ceil((p + 1) * bin)) + start, clipped to the map."""
    x1, y1, x2, y2 = [c_round(Fraction(float(np.float32(c))) * Fraction(scale)) for c in box]
    rw, rh = max(x2 - x1 + 1, 1), max(y2 - y1 + 1, 1)
    edges = []
    for start, ext, n in ((y1, rh, H), (x1, rw, W)):
        e = []
        for p in range(7):
            lo = (Fraction(p * ext, 7)).__floor__() + start
            hi = (Fraction((p + 1) * ext, 7)).__ceil__() + start
            e.append((min(max(lo, 0), n), min(max(hi, 0), n)))
        edges.append(e)
    return edges


def gen_roi_pool(out):
    H, W = 12, 16
    y, x = np.mgrid[0:H, 0:W]
    feat = np.stack([(y * W + x).astype(np.float32),            # strictly increasing in scan order: last element wins
                     np.full((H, W), 2.5, np.float32),          # constant: the FIRST element wins the tie
                     (-(y * W + x)).astype(np.float32)])[None]  # strictly decreasing: the first element wins
    boxes = [[0, 0, 6, 6],        # extent 7: one pixel per bin
             [2, 1, 15, 11],      # extent 14 x 11: bins of 2 columns, 1-2 rows
             [3, 2, 4, 3],        # extent 2 x 2: bins overlap (floor / ceil), each covers one pixel
             [10, 8, 30, 20],     # runs off the map: the bins beyond it are empty -> 0, argmax -1
             [2.5, 1.5, 8.5, 9.5],  # halves round away from zero: 3, 2, 9, 10
             [40, 40, 50, 50]]    # entirely outside: every bin empty
    scale = Fraction(1, 2)        # boxes are given in image pixels (x 2) below
    rois = np.asarray([[0] + [c * 2 for c in b] for b in boxes], np.float32)
    K = len(boxes)
    val = np.zeros((K, 3, 7, 7), np.float32)
    arg = np.full((K, 3, 7, 7), -1, np.int32)
    for k in range(K):
        ey, ex = roi_pool_bins(rois[k, 1:], scale, H, W)
        for ph, (hs, he) in enumerate(ey):
            for pw, (ws, we) in enumerate(ex):
                if he <= hs or we <= ws:
                    continue  # empty bin: value 0, argmax -1
                first, last = hs * W + ws, (he - 1) * W + (we - 1)
                val[k, 0, ph, pw], arg[k, 0, ph, pw] = last, last
                val[k, 1, ph, pw], arg[k, 1, ph, pw] = 2.5, first
                val[k, 2, ph, pw], arg[k, 2, ph, pw] = -first, first
    assert (arg[3] == -1).any() and (arg[3] >= 0).any() and (arg[5] == -1).all()
    assert np.array_equal(arg[0, 0], (y * W + x)[:7, :7])
    out["rp.feat"] = feat
    out["rp.rois"] = rois
    out["rp.scale"] = np.asarray(float(scale))
    out["rp.expect"] = val
    out["rp.argmax"] = arg


if __name__ == "__main__":
    out = {}
    gen_roi_align(out)
    gen_nms(out)
    gen_roi_pool(out)
    path = os.path.join(HERE, "closed_form.npz")
    np.savez_compressed(path, **out)
    print("closed_form.npz", os.path.getsize(path), "bytes,", len(out), "arrays")
