"""Two-FC head -- stock PyTorch-ROCm (out of kernel scope; mirrors ROIHead/mlp_head.py:19-49)."""
from torch.nn import Linear, Module, functional


class TwoMLPHead(Module):
    def __init__(self, in_channels, representation_size):
        super().__init__()
        self.fc1 = Linear(in_channels, representation_size)
        self.fc2 = Linear(representation_size, representation_size)

    def forward(self, x):
        x = x.flatten(start_dim=1)
        x = functional.relu(self.fc1(x))
        x = functional.relu(self.fc2(x))
        return x
