"""GeneralizedTransform of the reference (code/FasterRCNN/generalized_transform.py:21-83) on the GPU.

Same constructor and forward signature; the bilinear resize and the zero-padded batch assembly
(`utils.batched_images`, utils.py:339-372) are one kernel launch per image that writes the image's slice
of the batch tensor directly.  Bug-compatible by default, quirk for quirk:
  * the `if` at :54 is always taken, so every image is scaled by max_size / max(h, w);
  * boxes are multiplied by old/new size (the inverse ratio, :69-79) and come back flattened to 1-D
    in [all x1, all y1, all x2, all y2] order by the `torch.cat` at :81;
  * the canvas scan's `elif` (utils.py:356) ignores the width of an image that raised the height; where
    the reference then crashes inside `batched_images` this raises a ValueError.
Divergence on purpose: the batch lives on the images' HIP device (the reference builds it on the CPU).
"""
import math

import torch
from torch.nn import Module

from . import ops


def canvas_size(sizes, divide_size):
    """utils.py:351-360 on a list of (h, w): the max scan with its `elif`, rounded up to divide_size."""
    max_height = max_width = 0
    for h, w in sizes:
        if h > max_height:
            max_height = h
        elif w > max_width:
            max_width = w
    return math.ceil(max_height / divide_size) * divide_size, math.ceil(max_width / divide_size) * divide_size


def batched_images(images, divide_size, out=None):
    """utils.py:339-372 -- zero-pad a list of (C,h,w) images to one (B,C,H,W) batch.  `out` (not in the
    reference): a contiguous fp32 batch of exactly that shape to write into instead of a new tensor."""
    sizes = [(int(im.size(1)), int(im.size(2))) for im in images]
    return _assemble(images, sizes, divide_size, out)


def _assemble(images, out_sizes, divide_size, out=None):
    canvas_h, canvas_w = canvas_size(out_sizes, divide_size)
    channels = int(images[0].size(0))
    for (h, w), im in zip(out_sizes, images):
        if h > canvas_h or w > canvas_w or int(im.size(0)) != channels:
            raise ValueError("a (%d, %d, %d) image does not fit the (%d, %d, %d) canvas the reference computes "
                             "(utils.py:353-357: the elif skips widths)" % (im.size(0), h, w, channels, canvas_h, canvas_w))
    shape = (len(images), channels, canvas_h, canvas_w)
    if out is not None:
        # a caller that assembles same-shaped batches step after step keeps one buffer: a fresh ~100 MB tensor per
        # call can send torch's caching allocator to hipMalloc (milliseconds; the kernel itself is ~0.06 ms)
        if tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() \
                or out.device != images[0].device:
            raise ValueError("out must be a contiguous fp32 %s tensor on %s" % (shape, images[0].device))
        batch = out
    else:
        batch = torch.empty(shape, dtype=torch.float32, device=images[0].device)
    ops.resize_pad_batch(images, out_sizes, batch)   # one launch for the list (per 32 images)
    return batch


class GeneralizedTransform(Module):
    def __init__(self, min_size, max_size):
        super().__init__()
        self.min_size = min_size
        self.max_size = max_size

    def forward(self, images, targets, out=None):
        """`out` (not in the reference): batch buffer to reuse, see batched_images."""
        out_sizes = []
        for i in range(len(images)):
            image, target = images[i], targets[i]
            height, width = int(image.size(1)), int(image.size(2))
            min_size, max_size = float(min(height, width)), float(max(height, width))
            scale_factor = self.min_size / min_size
            if min_size * scale_factor >= self.min_size:              # generalized_transform.py:54 (always true)
                scale_factor = self.max_size / max_size
            # F.interpolate(recompute_scale_factor=True): output size = floor(size * scale) in double
            new_h, new_w = int(math.floor(float(height) * scale_factor)), int(math.floor(float(width) * scale_factor))
            out_sizes.append((new_h, new_w))
            if target is not None:
                box = target['box']
                ratio_height = torch.tensor([height / new_h], dtype=box.dtype, device=box.device)   # :69-74
                ratio_width = torch.tensor([width / new_w], dtype=box.dtype, device=box.device)
                target['box'] = torch.cat([box[:, 0] * ratio_width, box[:, 1] * ratio_height,
                                           box[:, 2] * ratio_width, box[:, 3] * ratio_height])       # :76-81
        return _assemble(images, out_sizes, 32, out), targets
