"""RNet: the attention residual U-Net (reference ``networks/AttResUNet.py``) on the HIP path.

The module tree, attribute names and parameter shapes follow the reference so that its checkpoints load with
``strict=True``; ``forward`` hands the work to :mod:`virnet_amd.engine`.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
from torch import nn

from .params import ConvParam
from .. import engine


class AttLayer(nn.Module):
    """SFT scale/shift generator: 1x1 convs on the extra maps (AttResUNet.py:11-32)."""

    def __init__(self, out_chn: int = 64, extra_chn: int = 4):
        super().__init__()
        nf1, nf2 = out_chn // 8, out_chn // 4
        self.out_chn, self.extra_chn = out_chn, extra_chn
        self.conv1 = ConvParam(extra_chn, nf1, 1)
        self.conv2 = ConvParam(nf1, nf2, 1)
        self.mul_conv = ConvParam(nf2, out_chn, 1)
        self.add_conv = ConvParam(nf2, out_chn, 1)


class AttResBlock(nn.Module):
    """Pre-activation residual block with optional SFT modulation (AttResUNet.py:34-60)."""

    def __init__(self, nf: int = 64, extra_chn: int = 4):
        super().__init__()
        self.nf, self.extra_chn = nf, extra_chn
        if extra_chn > 0:
            self.sft1 = AttLayer(nf, extra_chn)
            self.sft2 = AttLayer(nf, extra_chn)
        self.conv1 = ConvParam(nf, nf, 3)
        self.conv2 = ConvParam(nf, nf, 3)


class DownBlock(nn.Module):
    """n_resblocks x AttResBlock, then a 3x3 stride-2 conv (AttResUNet.py:62-75)."""

    def __init__(self, in_chn: int = 64, out_chn: int = 128, extra_chn: int = 4, n_resblocks: int = 1,
                 downsample: bool = True):
        super().__init__()
        self.body = nn.ModuleList([AttResBlock(in_chn, extra_chn) for _ in range(n_resblocks)])
        self.downsampler = ConvParam(in_chn, out_chn, 3, stride=2) if downsample else nn.Identity()


class UpBlock(nn.Module):
    """ConvTranspose2d(k2,s2) + bridge, then n_resblocks x AttResBlock without SFT (AttResUNet.py:77-90)."""

    def __init__(self, in_chn: int = 128, out_chn: int = 64, n_resblocks: int = 1):
        super().__init__()
        self.upsampler = ConvParam(in_chn, out_chn, 2, transposed=True, stride=2)
        self.body = nn.ModuleList([AttResBlock(nf=out_chn, extra_chn=0) for _ in range(n_resblocks)])


class AttResUNet(nn.Module):
    def __init__(self, in_chn: int = 3, extra_chn: int = 4, out_chn: int = 3, n_resblocks: int = 2,
                 n_feat: Sequence[int] = (64, 128, 196, 256), extra_mode: str = "Input"):
        super().__init__()
        assert isinstance(n_feat, (tuple, list))                      # AttResUNet.py:110
        self.depth = len(n_feat)
        self.extra_mode = extra_mode.lower()
        assert self.extra_mode in ["null", "input", "down", "both"]   # AttResUNet.py:113-114
        self.in_chn, self.extra_chn, self.out_chn = in_chn, extra_chn, out_chn
        self.n_feat, self.n_resblocks = list(n_feat), n_resblocks
        for c in n_feat:
            if c % 32:
                raise ValueError(f"n_feat={list(n_feat)}: the MFMA kernels tile output channels in blocks of 32 "
                                 f"(all reference configurations use multiples of 32)")
        head_in = in_chn if self.extra_mode in ("down", "null") else in_chn + extra_chn
        if head_in > 16:
            raise ValueError("image + conditioning channels must fit one 16-channel pixel record")
        self.head = ConvParam(head_in, n_feat[0], 3)
        extra_down = extra_chn if self.extra_mode in ("down", "both") else 0
        self.down_path = nn.ModuleList()
        for ii in range(self.depth):
            last = ii + 1 == self.depth
            self.down_path.append(DownBlock(n_feat[ii], n_feat[ii] if last else n_feat[ii + 1], extra_chn=extra_down,
                                            n_resblocks=n_resblocks, downsample=not last))
        self.up_path = nn.ModuleList()
        for jj in reversed(range(self.depth - 1)):
            self.up_path.append(UpBlock(n_feat[jj + 1], n_feat[jj], n_resblocks))
        self.tail = ConvParam(n_feat[0], out_chn, 3)

    def forward(self, x_in: torch.Tensor, extra_maps_in: Optional[torch.Tensor]) -> torch.Tensor:
        """x_in [N,C,h,w], extra maps [N,E,h,w] (or None for extra_mode='null') -> [N,out_chn,h,w] (AttResUNet.py:141-175)."""
        return engine.rnet_forward(self, x_in, extra_map=extra_maps_in)
