"""objectdetection_amd -- MI355X-native drop-in for the Faster R-CNN region-proposal / RoI hot
path of DingLuBJUT/ObjectDetection (code/FasterRCNN).

Same module layout and signatures as the reference:

    objectdetection_amd.utils                  <- code/FasterRCNN/utils.py
    objectdetection_amd.RPN.anchor_generator   <- code/FasterRCNN/RPN/anchor_generator.py
    objectdetection_amd.RPN.rpn                <- code/FasterRCNN/RPN/rpn.py
    objectdetection_amd.ROIHead.roi_pooling    <- code/FasterRCNN/ROIHead/roi_pooling.py
    objectdetection_amd.ROIHead.roi_head       <- code/FasterRCNN/ROIHead/roi_head.py

All arithmetic on the path runs in hand-written gfx950 kernels behind the C ABI of
include/frcnn_hip.h (objectdetection_amd/lib/libfrcnn_hip.so).  There is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["utils", "ops", "RPN", "ROIHead"]
