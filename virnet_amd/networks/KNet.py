"""KNet: the blur-kernel encoder (reference ``networks/KNet.py``) on the HIP path."""
from __future__ import annotations

import torch
from torch import nn

from .params import ConvParam
from .. import engine


class CALayer(nn.Module):
    """Squeeze-excite channel attention (KNet.py:12-26); the 1x1 convs are body.0 and body.2."""

    def __init__(self, nf: int, reduction: int = 16):
        super().__init__()
        self.body = nn.ModuleDict({"0": ConvParam(nf, nf // reduction, 1), "2": ConvParam(nf // reduction, nf, 1)})


class RB_Layer(nn.Module):
    """x + CA(conv(lrelu(conv(x)))) (KNet.py:28-39); convs at body.0 / body.2, CALayer at body.3."""

    def __init__(self, nf: int):
        super().__init__()
        self.body = nn.ModuleDict({"0": ConvParam(nf, nf, 3), "2": ConvParam(nf, nf, 3), "3": CALayer(nf)})


class KernelNet(nn.Module):
    def __init__(self, in_nc: int = 3, out_chn: int = 3, nf: int = 64, num_blocks: int = 8, scale: int = 4):
        super().__init__()
        self.in_nc, self.out_chn, self.nf, self.num_blocks = in_nc, out_chn, nf, num_blocks
        self.head = ConvParam(in_nc, nf, 9, bias=False, stride=4)
        self.body = nn.ModuleList([RB_Layer(nf) for _ in range(num_blocks)])
        self.tail = nn.ModuleDict({"0": ConvParam(nf, out_chn, 3)})

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[N,3,h,w] -> kernel descriptor [N,3,1,1] = (lam1, lam2, rho) (KNet.py:52-59)."""
        return engine.knet_forward(self, x)
