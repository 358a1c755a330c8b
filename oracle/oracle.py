"""CPU oracle for the Faster R-CNN proposal / RoI hot path -- TEST INFRASTRUCTURE ONLY.

Thin numpy/ctypes front end over ``oracle/frcnn_oracle.c`` plus numpy restatements of the
reference's orchestration (``RPN/rpn.py:54-85`` proposal half, ``RPN/rpn.py:105-122`` and
``ROIHead/roi_head.py:98-125`` target halves, ``torchvision.ops.MultiScaleRoIAlign``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product package ``objectdetection_amd`` never does.

Pinning: see the header of ``frcnn_oracle.c``.  Functions restating torchvision ops and the
top-k tie rule are PARITY UNPINNED (no reference fixture exists and torchvision is absent).
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfrcnn_oracle.so")


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "frcnn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None
_F = ctypes.POINTER(ctypes.c_float)
_I32 = ctypes.POINTER(ctypes.c_int32)
_I64 = ctypes.POINTER(ctypes.c_int64)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_remove_small_box.restype = ctypes.c_int
        _lib.orc_nms.restype = ctypes.c_int
        _lib.orc_max_threads.restype = ctypes.c_int
        _lib.orc_get_threads.restype = ctypes.c_int
    return _lib


def set_threads(n):
    lib().orc_set_threads(ctypes.c_int(int(n)))


def max_threads():
    return int(lib().orc_max_threads())


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _fp(a):
    return a.ctypes.data_as(_F)


def _null(t):
    return ctypes.cast(None, t)


# ----------------------------------------------------------------------------- anchors
def anchor_pairs(anchor_sizes=(128, 256, 512), anchor_ratios=(0.5, 1.0, 2.0)):
    """(size, ratio) per base anchor, following AnchorGenerator.__init__/get_base_anchor
    (RPN/anchor_generator.py:31-64)."""
    if not isinstance(anchor_sizes[0], (list, tuple)):
        anchor_sizes = tuple((s,) for s in anchor_sizes)
    if not isinstance(anchor_ratios[0], (list, tuple)):
        anchor_ratios = (anchor_ratios,) * len(anchor_sizes)
    sizes, ratios = [], []
    for size, ratio in zip(anchor_sizes, anchor_ratios):
        if len(size) not in (1, len(ratio)):
            raise ValueError("anchor sizes per group must be 1 or len(ratios)")
        for j, r in enumerate(ratio):
            sizes.append(size[0] if len(size) == 1 else size[j])
            ratios.append(r)
    return np.asarray(sizes, np.float32), np.asarray(ratios, np.float32)


def base_anchors(anchor_sizes=(128, 256, 512), anchor_ratios=(0.5, 1.0, 2.0)):
    s, r = anchor_pairs(anchor_sizes, anchor_ratios)
    out = np.empty((len(s), 4), np.float32)
    lib().orc_base_anchors(_fp(s), _fp(r), ctypes.c_int(len(s)), _fp(out))
    return out


def anchor_grid(image_h, image_w, fh, fw, base):
    """One level of AnchorGenerator.generate_anchor -> (fh*fw, na, 4).  Strides are the
    Python-double quotient rounded to fp32 (RPN/anchor_generator.py:106-107)."""
    base = _f32(base)
    sy = np.float32(image_h / fh)
    sx = np.float32(image_w / fw)
    out = np.empty((fh * fw, base.shape[0], 4), np.float32)
    lib().orc_anchor_grid(ctypes.c_int(fh), ctypes.c_int(fw), ctypes.c_float(sy), ctypes.c_float(sx),
                          _fp(base), ctypes.c_int(base.shape[0]), _fp(out))
    return out


def anchors_all_levels(batch_image_info, feature_sizes, anchor_sizes=(128, 256, 512),
                       anchor_ratios=(0.5, 1.0, 2.0)):
    """cat over levels of anchor.flatten(0,1) -> (A, 4)  (RPN/rpn.py:55, one image)."""
    base = base_anchors(anchor_sizes, anchor_ratios)
    _, H, W = batch_image_info
    return np.concatenate([anchor_grid(H, W, fh, fw, base).reshape(-1, 4) for fh, fw in feature_sizes], 0)


# ---------------------------------------------------------------------------- box math
def box_iou(box1, box2):
    box1, box2 = _f32(box1).reshape(-1, 4), _f32(box2).reshape(-1, 4)
    out = np.empty((box1.shape[0], box2.shape[0]), np.float32)
    lib().orc_box_iou(_fp(box1), ctypes.c_int(box1.shape[0]), _fp(box2), ctypes.c_int(box2.shape[0]), _fp(out))
    return out


def iou_label_encode(box1, box2, neg_thr, pos_thr, labels=None, weights=None):
    """-> (label fp32 (n), argmax int64 (n), max_iou (n), deviation (n,4))."""
    box1, box2 = _f32(box1).reshape(-1, 4), _f32(box2).reshape(-1, 4)
    n, m = box1.shape[0], box2.shape[0]
    if m == 0:
        raise ValueError("box_2 is empty: the reference's iou.max(dim=1) raises on (N,0)")
    label = np.empty(n, np.float32)
    arg = np.empty(n, np.int64)
    mx = np.empty(n, np.float32)
    dev = np.empty((n, 4), np.float32)
    lab = _f32(labels) if labels is not None else None
    wt = _f32(weights) if weights is not None else None
    lib().orc_iou_label_encode(
        _fp(box1), ctypes.c_int(n), _fp(box2), ctypes.c_int(m),
        ctypes.c_float(np.float32(neg_thr)), ctypes.c_float(np.float32(pos_thr)),
        _fp(lab) if lab is not None else _null(_F), _fp(wt) if wt is not None else _null(_F),
        _fp(label), arg.ctypes.data_as(_I64), _fp(mx), _fp(dev))
    return label, arg, mx, dev


def tagging_box(box1, box2, neg_thr, pos_thr, labels=None):
    return iou_label_encode(box1, box2, neg_thr, pos_thr, labels)[0]


def box_deviation(box1, box2, weights=None):
    return iou_label_encode(box1, box2, 0.0, 0.0, None, weights)[3]


def box_regression(box, regs):
    box, regs = _f32(box).reshape(-1, 4), _f32(regs)
    n = box.shape[0]
    regs = regs.reshape(n, -1)
    K = regs.shape[1] // 4
    out = np.empty((n, 4 * K), np.float32)
    lib().orc_box_regression(_fp(box), _fp(regs), ctypes.c_int(n), ctypes.c_int(K), _fp(out))
    return out


def box_regression_backward(box, regs, grad_out):
    """-> (grad_regs, grad_box): the reference's autograd through utils.py:173-208."""
    box, regs, grad_out = _f32(box).reshape(-1, 4), _f32(regs), _f32(grad_out)
    n = box.shape[0]
    regs, grad_out = regs.reshape(n, -1), grad_out.reshape(n, -1)
    K = regs.shape[1] // 4
    grad_regs = np.empty((n, 4 * K), np.float32)
    grad_box = np.empty((n, 4), np.float32)
    lib().orc_box_regression_backward(_fp(box), _fp(regs), _fp(grad_out), ctypes.c_int(n), ctypes.c_int(K),
                                      _fp(grad_regs), _fp(grad_box))
    return grad_regs, grad_box


def transform_sizes(shapes, min_size, max_size):
    """Host logic of generalized_transform.py:46-56 + utils.py:351-360, bugs included.
    shapes: [(C,h,w)] -> ([(out_h,out_w)], canvas_h, canvas_w).  The `if` at :54 is always true up to
    rounding, so the scale is max_size / max(h,w); the `elif` at utils.py:356 skips the width of any
    image that raised the height."""
    import math
    outs = []
    for _, h, w in shapes:
        mn, mx = float(min(h, w)), float(max(h, w))
        scale = min_size / mn
        if mn * scale >= min_size:
            scale = max_size / mx
        outs.append((int(math.floor(float(h) * scale)), int(math.floor(float(w) * scale))))
    return (outs,) + canvas_size(outs, 32)


def canvas_size(sizes, divide_size):
    """utils.py:351-360: max height / width scan with its `elif`, rounded up to divide_size."""
    import math
    max_h = max_w = 0
    for h, w in sizes:
        if h > max_h:
            max_h = h
        elif w > max_w:
            max_w = w
    return math.ceil(max_h / divide_size) * divide_size, math.ceil(max_w / divide_size) * divide_size


def resize_pad(image, out_h, out_w, canvas_h, canvas_w):
    image = _f32(image)
    C, h, w = image.shape
    out = np.empty((C, canvas_h, canvas_w), np.float32)
    lib().orc_resize_pad(_fp(image), ctypes.c_int(C), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(out_h),
                         ctypes.c_int(out_w), _fp(out), ctypes.c_int(canvas_h), ctypes.c_int(canvas_w))
    return out


def generalized_transform(images, boxes, min_size, max_size):
    """GeneralizedTransform.forward -> (batch (B,C,H,W), [flattened rescaled boxes]); raises ValueError where
    the reference's canvas is too small for an image (it raises a RuntimeError there)."""
    outs, ch, cw = transform_sizes([im.shape for im in images], min_size, max_size)
    batch, new_boxes = [], []
    for im, box, (oh, ow) in zip(images, boxes, outs):
        if oh > ch or ow > cw:
            raise ValueError("image does not fit the canvas")
        batch.append(resize_pad(im, oh, ow, ch, cw))
        rh = np.float32(im.shape[1] / oh)      # generalized_transform.py:69-74: old / new (inverse, as written)
        rw = np.float32(im.shape[2] / ow)
        box = _f32(box)
        new_boxes.append(np.concatenate([box[:, 0] * rw, box[:, 1] * rh, box[:, 2] * rw, box[:, 3] * rh]))
    return np.stack(batch), new_boxes


def rpn_sampled_loss(scores, regs, labels, reg_labels, index, beta=1.0 / 9):
    """rpn.py:122-137 -> (loss (2), grad_scores, grad_regs) with the inputs' shapes."""
    scores, regs, labels, reg_labels = _f32(scores), _f32(regs), _f32(labels), _f32(reg_labels)
    index = np.ascontiguousarray(index, np.int64)
    loss = np.empty(2, np.float32)
    gs, gr = np.empty_like(scores), np.empty_like(regs)
    lib().orc_rpn_sampled_loss(_fp(scores), _fp(regs), _fp(labels), _fp(reg_labels),
                               index.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(index.size),
                               ctypes.c_int64(scores.size), ctypes.c_float(beta), _fp(loss), _fp(gs), _fp(gr))
    return loss, gs, gr


def roi_loss(logits, pred, cls_label, reg_label, beta=1.0 / 9):
    """roi_head.py:141-152 -> (loss (2), grad_logits, grad_pred)."""
    logits, pred, cls_label, reg_label = _f32(logits), _f32(pred), _f32(cls_label), _f32(reg_label)
    N, Kc = logits.shape
    loss = np.empty(2, np.float32)
    gl, gp = np.empty_like(logits), np.empty_like(pred)
    lib().orc_roi_loss(_fp(logits), _fp(pred), _fp(cls_label), _fp(reg_label), ctypes.c_int(N), ctypes.c_int(Kc),
                       ctypes.c_float(beta), _fp(loss), _fp(gl), _fp(gp))
    return loss, gl, gp


def box_clip(box, image_size):
    box = _f32(box).reshape(-1, 4)
    h, w = image_size
    out = np.empty_like(box)
    lib().orc_box_clip(_fp(box), ctypes.c_int(box.shape[0]), ctypes.c_float(np.float32(h)),
                       ctypes.c_float(np.float32(w)), _fp(out))
    return out


def remove_small_box(box, size_thresh):
    box = _f32(box).reshape(-1, 4)
    idx = np.empty(box.shape[0], np.int64)
    n = lib().orc_remove_small_box(_fp(box), ctypes.c_int(box.shape[0]),
                                   ctypes.c_float(np.float32(size_thresh)), idx.ctypes.data_as(_I64))
    idx = idx[:n].copy()
    return box[idx], idx


# ------------------------------------------------------------------- ordering, top-k, NMS
def argsort_desc(scores):
    scores = _f32(scores).reshape(-1)
    out = np.empty(scores.shape[0], np.int64)
    lib().orc_argsort_desc(_fp(scores), ctypes.c_int(scores.shape[0]), out.ctypes.data_as(_I64))
    return out


def topk(scores, k):
    scores = _f32(scores).reshape(-1)
    k = min(int(k), scores.shape[0])
    idx = np.empty(k, np.int64)
    val = np.empty(k, np.float32)
    lib().orc_topk(_fp(scores), ctypes.c_int(scores.shape[0]), ctypes.c_int(k), idx.ctypes.data_as(_I64), _fp(val))
    return val, idx


def nms(boxes, scores, iou_threshold, max_keep=-1):
    boxes, scores = _f32(boxes).reshape(-1, 4), _f32(scores).reshape(-1)
    n = boxes.shape[0]
    keep = np.empty(max(n, 1), np.int64)
    nk = lib().orc_nms(_fp(boxes), _fp(scores), ctypes.c_int(n), ctypes.c_double(float(iou_threshold)),
                       ctypes.c_int(int(max_keep)), keep.ctypes.data_as(_I64))
    return keep[:nk].copy()


# ----------------------------------------------------------------------- RoI pooling ops
def roi_align(feat, rois, output_size, spatial_scale, sampling_ratio, sel=None, sel_value=0, out=None):
    feat = _f32(feat)
    rois = _f32(rois).reshape(-1, 5)
    B, C, H, W = feat.shape
    K = rois.shape[0]
    ph, pw = output_size
    if out is None:
        out = np.zeros((K, C, ph, pw), np.float32)
    selp = sel.ctypes.data_as(_I32) if sel is not None else _null(_I32)
    lib().orc_roi_align(_fp(feat), ctypes.c_int(B), ctypes.c_int(C), ctypes.c_int(H), ctypes.c_int(W),
                        _fp(rois), ctypes.c_int(K), ctypes.c_int(ph), ctypes.c_int(pw),
                        ctypes.c_float(np.float32(spatial_scale)), ctypes.c_int(int(sampling_ratio)),
                        selp, ctypes.c_int(int(sel_value)), _fp(out))
    return out


def roi_align_backward(grad_out, feat_shape, rois, output_size, spatial_scale, sampling_ratio, sel=None,
                       sel_value=0, grad_feat=None):
    """d(sum(out * grad_out)) / d feat of roi_align: (B,C,H,W), accumulated into grad_feat when given."""
    grad_out = _f32(grad_out)
    rois = _f32(rois).reshape(-1, 5)
    B, C, H, W = feat_shape
    K = rois.shape[0]
    ph, pw = output_size
    if grad_feat is None:
        grad_feat = np.zeros((B, C, H, W), np.float32)
    selp = sel.ctypes.data_as(_I32) if sel is not None else _null(_I32)
    lib().orc_roi_align_backward(_fp(grad_out), ctypes.c_int(B), ctypes.c_int(C), ctypes.c_int(H), ctypes.c_int(W),
                                 _fp(rois), ctypes.c_int(K), ctypes.c_int(ph), ctypes.c_int(pw),
                                 ctypes.c_float(np.float32(spatial_scale)), ctypes.c_int(int(sampling_ratio)),
                                 selp, ctypes.c_int(int(sel_value)), _fp(grad_feat))
    return grad_feat


def multiscale_roi_align_backward(grad_out, feature_shapes, list_proposals, image_shapes, output_size=(7, 7),
                                  sampling_ratio=2):
    """Backward of multiscale_roi_align with respect to every feature map -> list of (B,C,h,w)."""
    rois = np.concatenate(
        [np.concatenate([np.full((len(p), 1), i, np.float32), _f32(p).reshape(-1, 4)], 1)
         for i, p in enumerate(list_proposals)], 0)
    max_h = max(s[0] for s in image_shapes)
    max_w = max(s[1] for s in image_shapes)
    scales = [infer_scale(fs[-2:], (max_h, max_w)) for fs in feature_shapes]
    if len(feature_shapes) == 1:
        return [roi_align_backward(grad_out, feature_shapes[0], rois, output_size, scales[0], sampling_ratio)]
    lvl = level_map(rois[:, 1:5], int(-math.log2(scales[0])), int(-math.log2(scales[-1])))
    return [roi_align_backward(grad_out, fs, rois, output_size, s, sampling_ratio, sel=lvl, sel_value=l)
            for l, (fs, s) in enumerate(zip(feature_shapes, scales))]


def roi_pool(feat, rois, output_size, spatial_scale):
    feat = _f32(feat)
    rois = _f32(rois).reshape(-1, 5)
    B, C, H, W = feat.shape
    K = rois.shape[0]
    ph, pw = output_size
    out = np.zeros((K, C, ph, pw), np.float32)
    arg = np.full((K, C, ph, pw), -1, np.int32)
    lib().orc_roi_pool(_fp(feat), ctypes.c_int(B), ctypes.c_int(C), ctypes.c_int(H), ctypes.c_int(W),
                       _fp(rois), ctypes.c_int(K), ctypes.c_int(ph), ctypes.c_int(pw),
                       ctypes.c_float(np.float32(spatial_scale)), _fp(out), arg.ctypes.data_as(_I32))
    return out, arg


def infer_scale(feature_hw, image_hw):
    """torchvision poolers._infer_scale: 2 ** round(log2(fp32(s_feat / s_img))), both axes agree."""
    scales = []
    for s1, s2 in zip(feature_hw, image_hw):
        approx = np.float32(float(s1) / float(s2))
        scales.append(2.0 ** float(np.round(np.log2(approx))))
    if scales[0] != scales[1]:
        raise AssertionError("feature/image scale differs between axes: %r" % (scales,))
    return scales[0]


def level_map(boxes, k_min, k_max):
    boxes = _f32(boxes).reshape(-1, 4)
    out = np.empty(boxes.shape[0], np.int32)
    lib().orc_level_map(_fp(boxes), ctypes.c_int(boxes.shape[0]), ctypes.c_int(int(k_min)),
                        ctypes.c_int(int(k_max)), out.ctypes.data_as(_I32))
    return out


def multiscale_roi_align(features, list_proposals, image_shapes, output_size=(7, 7), sampling_ratio=2):
    """MultiScaleRoIAlign.forward.  features: list of (B,C,h,w) arrays in
    feature-name order; list_proposals: list[B] of (P_i,4); image_shapes: list of (h,w)."""
    rois = np.concatenate(
        [np.concatenate([np.full((len(p), 1), i, np.float32), _f32(p).reshape(-1, 4)], 1)
         for i, p in enumerate(list_proposals)], 0) if len(list_proposals) else np.zeros((0, 5), np.float32)
    max_h = max(s[0] for s in image_shapes)
    max_w = max(s[1] for s in image_shapes)
    scales = [infer_scale(f.shape[-2:], (max_h, max_w)) for f in features]
    k_min = int(-math.log2(scales[0]))
    k_max = int(-math.log2(scales[-1]))
    C = features[0].shape[1]
    if len(features) == 1:
        return roi_align(features[0], rois, output_size, scales[0], sampling_ratio), np.zeros(len(rois), np.int32)
    lvl = level_map(rois[:, 1:5], k_min, k_max)
    out = np.zeros((rois.shape[0], C, output_size[0], output_size[1]), np.float32)
    for l, (f, s) in enumerate(zip(features, scales)):
        roi_align(f, rois, output_size, s, sampling_ratio, sel=lvl, sel_value=l, out=out)
    return out, lvl


# ------------------------------------------------------------------------- orchestration
def flatten_rpn_outputs(list_cls, list_regs):
    """RPN/rpn.py:56-57 for a batch: cls (B,na,h,w)->(B,A); reg (B,4na,h,w)->(B,A,4)."""
    cls = np.concatenate([c.transpose(0, 2, 3, 1).reshape(c.shape[0], -1) for c in list_cls], 1)
    reg = np.concatenate([r.transpose(0, 2, 3, 1).reshape(r.shape[0], -1, 4) for r in list_regs], 1)
    return np.ascontiguousarray(cls, np.float32), np.ascontiguousarray(reg, np.float32)


def rpn_proposals_image(anchors, scores, regs, image_hw, num_pre_nms, num_post_nms, nms_threshold,
                        proposal_min_size, stages=None):
    """RPN/rpn.py:63-85 for ONE image.  anchors (A,4), scores (A,), regs (A,4).
    Returns proposals (P,4); if `stages` is a dict it receives the intermediate results."""
    k = min(int(num_pre_nms), scores.shape[0])
    top_val, top_idx = topk(scores, k)
    top_reg = regs[top_idx]
    top_anchor = anchors[top_idx]
    proposal = box_regression(top_anchor, top_reg)
    proposal = box_clip(proposal, image_hw)
    proposal, index = remove_small_box(proposal, proposal_min_size)
    score = top_val[index]
    keep = nms(proposal, score, nms_threshold)[:num_post_nms]
    if stages is not None:
        stages.update(top_idx=top_idx, top_val=top_val, small_index=index, filtered=proposal,
                      filtered_score=score, keep=keep)
    return proposal[keep]


def rpn_proposals(batch_image_info, list_cls, list_regs, params, anchor_sizes=(128, 256, 512),
                  anchor_ratios=(0.5, 1.0, 2.0), stages=None):
    """Proposal half of RPN.forward for a batch (RPN/rpn.py:54-85)."""
    B, H, W = batch_image_info
    feature_sizes = [c.shape[-2:] for c in list_cls]
    anchors = anchors_all_levels(batch_image_info, feature_sizes, anchor_sizes, anchor_ratios)
    cls, reg = flatten_rpn_outputs(list_cls, list_regs)
    out = []
    for b in range(B):
        st = {} if stages is not None else None
        out.append(rpn_proposals_image(anchors, cls[b], reg[b], (H, W), params["num_pre_nms"],
                                       params["num_post_nms"], params["nms_threshold"],
                                       params["proposal_min_size"], st))
        if stages is not None:
            stages.setdefault("per_image", []).append(st)
    return out


def sampling_counts(sampling_num, pos_fraction, pos_num, neg_num):
    """utils.get_sampling_factor's two counts (utils.py:306-308); the permutations themselves
    come from torch.randperm on the host in both the reference and the product."""
    neg_fraction = 1 - pos_fraction
    return int(min(sampling_num * pos_fraction, pos_num)), int(min(sampling_num * neg_fraction, neg_num))
