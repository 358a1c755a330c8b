"""AnchorGenerator with the reference's interface (RPN/anchor_generator.py:20-120), anchors
produced by the K1 kernel.

Bug-compatible with the reference: base anchors are w = size*ratio, h = size (Q2); strides are the
float quotients H_img/h_feat, W_img/w_feat with no half-cell offset (Q3); every base anchor is laid
on every level (Q5); the batch dimension is a stride-0 expand and the anchor count per cell is the
literal 9 of anchor_generator.py:119 (Q4: any other count raises, as it does there).
"""
import torch
from torch.nn import Module

from .. import ops


class AnchorGenerator(Module):
    def __init__(self, anchor_sizes=(128, 256, 512), anchor_ratios=(0.5, 1.0, 2.0)):
        super().__init__()
        if not isinstance(anchor_sizes[0], (list, tuple)):
            anchor_sizes = tuple((size,) for size in anchor_sizes)
        if not isinstance(anchor_ratios[0], (list, tuple)):
            anchor_ratios = (anchor_ratios,) * len(anchor_sizes)
        self.anchor_sizes = anchor_sizes
        self.anchor_ratios = anchor_ratios
        self.base_anchor = None
        self._cache = {}

    def get_base_anchor(self, data_device, data_type):
        """anchor_generator.py:43-64; cached after the first call like the reference."""
        if self.base_anchor is not None:
            return
        sizes, ratios = [], []
        for size, ratio in zip(self.anchor_sizes, self.anchor_ratios):
            if len(size) not in (1, len(ratio)):
                raise RuntimeError("anchor sizes per group must broadcast against the ratios")
            for j, r in enumerate(ratio):
                sizes.append(size[0] if len(size) == 1 else size[j])
                ratios.append(r)
        self.base_anchor = ops.base_anchors(sizes, ratios).to(device=data_device, dtype=data_type)

    def generate_anchor(self, feature_sizes, strides, device, data_type):
        """anchor_generator.py:66-95 -> list over levels of (h*w, na, 4)."""
        anchor_list = []
        for size, stride in zip(feature_sizes, strides):
            anchor_list.append(ops.anchor_grid(self.base_anchor, int(size[0]), int(size[1]), float(stride[0]),
                                               float(stride[1])))
        return anchor_list

    def level_anchors(self, batch_image_info, list_features):
        """Per-level (h*w, na, 4) anchors, cached per (image size, feature sizes, device)."""
        batch_height, batch_width = batch_image_info[1], batch_image_info[2]
        feature_sizes = tuple(tuple(int(v) for v in f.shape[2:]) for f in list_features)
        data_device, data_type = list_features[0].device, list_features[0].dtype
        if data_type != torch.float32:
            raise TypeError("AnchorGenerator kernels are fp32 (got %s)" % data_type)
        self.get_base_anchor(data_device, data_type)
        key = (float(batch_height), float(batch_width), feature_sizes, str(data_device))
        if key not in self._cache:
            # Python-double quotient, rounded to fp32 at the C boundary (anchor_generator.py:106-107)
            strides = [(batch_height / s[0], batch_width / s[1]) for s in feature_sizes]
            self._cache = {key: self.generate_anchor(feature_sizes, strides, data_device, data_type)}
        return self._cache[key]

    def forward(self, batch_image_info, list_features):
        batch_size = batch_image_info[0]
        list_anchors = []
        for anchor in self.level_anchors(batch_image_info, list_features):
            list_anchors.append(anchor[None, :].expand(batch_size, anchor.size(0), 9, 4))
        return list_anchors
