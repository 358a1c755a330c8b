"""Box math helpers with the signatures of the reference's code/FasterRCNN/utils.py, backed by
the gfx950 kernels.  Each function cites the reference lines it replaces.

Divergences from the reference, on purpose (SURVEY.md 8a, quirk register):
  * results live on the inputs' HIP device (the reference leaks CPU tensors, quirk Q6:
    utils.py:275,168,310-311 -- its target path only works on CPU);
  * outputs are detached (the kernels are not differentiable), except box_regression, which has a
    HIP backward;
  * CPU tensors are rejected: there is no CPU fallback.
Everything else is bug-compatible, including quirk Q1 (box_regression ignores the box origin).
"""
import math  # noqa: F401  (kept: the reference module exposes it)

import torch

from . import ops

# Thin aliases in BASELINE.json's vocabulary (decision 1)
__all__ = ["box_iou", "box_deviation", "box_regression", "remove_small_box", "box_clip", "tagging_box",
           "get_sampling_factor", "smooth_l1_loss", "batched_images", "loc2bbox", "bbox2loc", "bbox_iou"]


def batched_images(images, divide_size, out=None):
    """utils.py:339-372 -- zero-pad a list of (C,h,w) images into one (B,C,H,W) batch (on the device);
    `out` (an extension): a batch buffer of that shape to reuse."""
    from .generalized_transform import batched_images as impl
    return impl(images, divide_size, out)


def box_iou(box_1, box_2):
    """utils.py:93-120 -- (N,4),(M,4) -> (N,M) IoU, inter / ((a1 + a2) - inter), no epsilon."""
    return ops.box_iou(box_1.reshape(-1, 4), box_2.reshape(-1, 4))


def box_deviation(box_1, box_2, weights=None):
    """utils.py:123-171 -- regression targets of each box_1 row w.r.t. its arg-max-IoU box_2 row."""
    if isinstance(weights, torch.Tensor):
        weights = weights.detach().cpu().tolist()
    out = ops.iou_label_encode(box_1.reshape(-1, 4), box_2.reshape(1, -1, 4), None, 0.0, 0.0, None, weights,
                               want=("dev",))
    return out["dev"][0]


def box_regression(box, regression_params):
    """utils.py:173-208 -- decode (n,4K) deltas against (n,4) boxes; centre = (w/2, h/2) (Q1).
    Differentiable like the reference's (which keeps proposals on the graph, rpn.py:75,85): when an
    input requires grad the result carries a HIP backward (ops.BoxRegression)."""
    if torch.is_grad_enabled() and (box.requires_grad or regression_params.requires_grad):
        return ops.BoxRegression.apply(box, regression_params)
    return ops.box_regression(box, regression_params)


def remove_small_box(box, size_thresh, box_score=None, score_thresh=None):
    """utils.py:211-239 -- optional score prefilter, then keep rows with w > t and h > t.
    Returns (box[index], index) with int64 ascending indices (relative to the score-filtered
    rows when the prefilter is used, exactly like the reference)."""
    if box_score is not None and score_thresh is not None:
        index, count = ops.select_gt(box_score, score_thresh)
        box = box[index[: int(count.item())]]
    box_out, index, count = ops.remove_small_box(box, size_thresh)
    n = int(count.item())
    return box_out[:n], index[:n]


def box_clip(box, image_size):
    """utils.py:242-259 -- clamp x to [0, w] and y to [0, h]; image_size = (h, w)."""
    image_h, image_w = image_size
    return ops.box_clip(box, image_h, image_w)


def tagging_box(box_1, box_2, neg_threshold, pos_threshold, labels=None):
    """utils.py:261-291 -- IoU-threshold labels: > pos -> 1 (or labels[argmax]); < neg -> 0;
    in between -> -1 (assigned in that order); fp32 tensor (N,)."""
    lab = None
    if labels is not None:
        lab = torch.as_tensor(labels, dtype=torch.float32, device=box_1.device).reshape(1, -1)
    out = ops.iou_label_encode(box_1.reshape(-1, 4), box_2.reshape(1, -1, 4), None, neg_threshold, pos_threshold,
                               lab, None, want=("label",))
    return out["label"][0]


def get_sampling_factor(sampling_num, pos_fraction, pos_num, neg_num):
    """utils.py:293-312 -- host-side on purpose: torch.randperm on the CPU global RNG, positives
    first, is the only way to stay RNG-identical to the reference under torch.manual_seed.
    No negative top-up when positives are short (quirk Q9)."""
    neg_fraction = 1 - pos_fraction
    pos_sampling_num = int(min(sampling_num * pos_fraction, pos_num))
    neg_sampling_num = int(min(sampling_num * neg_fraction, neg_num))
    pos_factor = torch.randperm(pos_num)[:pos_sampling_num]
    neg_factor = torch.randperm(neg_num)[:neg_sampling_num]
    return pos_factor, neg_factor


def smooth_l1_loss(y_hat, y, beta=1. / 9, size_average=True):
    """utils.py:315-336 -- stock torch (the loss is outside the kernel scope)."""
    x = torch.abs(y_hat - y)
    loss = torch.where(torch.lt(x, beta), 0.5 * x ** 2 / beta, x - 0.5 * beta)
    return loss.mean() if size_average else loss.sum()


loc2bbox = box_regression
bbox2loc = box_deviation
bbox_iou = box_iou
