"""Class / box predictor -- stock PyTorch-ROCm (out of kernel scope; mirrors
ROIHead/predictor.py:19-49)."""
from torch.nn import Linear, Module


class FasterRCNNPredictor(Module):
    def __init__(self, in_channels, num_classes):
        super().__init__()
        self.cls_score = Linear(in_channels, num_classes)
        self.bbox_pred = Linear(in_channels, num_classes * 4)

    def forward(self, x):
        x = x.flatten(start_dim=1)
        return self.cls_score(x), self.bbox_pred(x)
