"""ROIPooling with the reference's interface (ROIHead/roi_pooling.py:21-54).

Default mode "align" = torchvision MultiScaleRoIAlign semantics (what the reference calls):
FPN level mapper + bilinear-average RoIAlign, sampling_ratio samples per bin axis, aligned=False.
Mode "max" is the north-star extra (torchvision roi_pool semantics with argmax, single level);
the reference never calls it.
"""
import torch
from torch.nn import Module

from .. import ops


class ROIPooling(Module):
    def __init__(self, list_feature_names, output_size, sampling_ratio, mode="align"):
        super().__init__()
        if mode not in ("align", "max"):
            raise ValueError("mode must be 'align' or 'max'")
        self.list_feature_names = list_feature_names
        self.output_size = output_size
        self.sampling_ratio = sampling_ratio
        self.mode = mode
        self.roi_pooling = ops.MultiScaleRoIAlign(self.list_feature_names, self.output_size, self.sampling_ratio)
        self.last_argmax = None

    def forward(self, dict_feature_maps, list_proposals, image_size):
        if self.mode == "align":
            return self.roi_pooling(dict_feature_maps, list_proposals, image_size)
        feats, scales, _, _ = self.roi_pooling._levels(dict_feature_maps, image_size)
        if len(feats) != 1:
            raise ValueError("max RoIPool mode is single-level")
        dev = feats[0].device
        rois = torch.cat([torch.cat([torch.full((p.shape[0], 1), float(i), device=dev),
                                     p.to(device=dev, dtype=torch.float32).reshape(-1, 4)], dim=1)
                          for i, p in enumerate(list_proposals)], dim=0)
        out, self.last_argmax = ops.roi_pool(feats[0], rois, tuple(self.output_size), scales[0])
        return out

    def forward_padded(self, dict_feature_maps, props, count, image_size):
        """Sync-free batched form: props (B,P,4) zero-padded, count (B) int32 on device.
        -> (features (B*P,C,oh,ow) with the first sum(count) rows valid, total (1) int32 on device)."""
        if self.mode == "align":
            return self.roi_pooling.forward_padded(dict_feature_maps, props, count, image_size)
        feats, scales, _, _ = self.roi_pooling._levels(dict_feature_maps, image_size)
        if len(feats) != 1:
            raise ValueError("max RoIPool mode is single-level")
        out, self.last_argmax, total = ops.roi_pool_batched(feats[0], props, count, tuple(self.output_size), scales[0])
        return out, total
