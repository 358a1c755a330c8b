"""RPN conv head -- stock PyTorch-ROCm (out of kernel scope; mirrors RPN/rpn_head.py:24-69 so the
reference's constructors and __main__ blocks run unchanged)."""
from torch.nn import Conv2d, Module, functional, init


class RPNHead(Module):
    def __init__(self, in_channels, num_pixel_anchors):
        super().__init__()
        self.cov = Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.cls = Conv2d(in_channels, num_pixel_anchors, kernel_size=1, stride=1)
        self.reg = Conv2d(in_channels, num_pixel_anchors * 4, kernel_size=1, stride=1)
        for layer in self.children():
            if isinstance(layer, Conv2d):
                init.normal_(layer.weight, std=0.01)
                init.constant_(layer.bias, 0)

    def forward(self, list_features):
        list_cls, list_regs = [], []
        for feature in list_features:
            t = functional.relu(self.cov(feature))
            list_cls.append(self.cls(t))
            list_regs.append(self.reg(t))
        return list_cls, list_regs
