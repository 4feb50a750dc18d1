"""SNet: the DnCNN variance encoder (reference ``networks/DnCNN.py:8-52``) on the HIP path."""
from __future__ import annotations

import torch
from torch import nn

from .SubBlocks import conv3x3
from .. import engine


class DnCNN(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, dep: int = 5, num_filters: int = 64, noise_avg: bool = False):
        super().__init__()
        self.dep, self.noise_avg = dep, noise_avg
        self.in_channels, self.out_channels = in_channels, out_channels
        self.conv1 = conv3x3(in_channels, num_filters, bias=True)
        # the reference interleaves LeakyReLU modules in an nn.Sequential, so convs sit at even indices (DnCNN.py:24-28)
        self.mid_layer = nn.ModuleDict({str(2 * (ii - 1)): conv3x3(num_filters, num_filters, bias=True)
                                        for ii in range(1, dep - 1)})
        self.conv_last = conv3x3(num_filters, out_channels, bias=True)
        self._initialize()

    def _initialize(self) -> None:
        # DnCNN.py:46-52: orthogonal weights with the leaky-relu(0.25) gain, zero bias
        gain = nn.init.calculate_gain("leaky_relu", 0.25)
        for m in (self.conv1, *self.mid_layer.values(), self.conv_last):
            nn.init.orthogonal_(m.weight, gain=gain)
            nn.init.constant_(m.bias, 0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Raw log-variance map [N,C,H,W], or [N,C,1,1] with ``noise_avg`` (DnCNN.py:37-44)."""
        return engine.snet_forward(self, x, mode="raw")
