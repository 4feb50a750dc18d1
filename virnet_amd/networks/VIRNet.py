"""Drop-in counterparts of the reference's ``networks/VIRNet.py`` boundary classes.

Same constructor keywords, same ``forward`` signatures and return values, same ``state_dict`` keys; the
arithmetic is the hand-written gfx950 path in ``libvirnet_hip`` (see ``include/virnet_hip.h``).
"""
from __future__ import annotations

from math import log
from typing import Sequence, Tuple

import torch
from torch import nn

from .AttResUNet import AttResUNet
from .DnCNN import DnCNN
from .KNet import KernelNet as KNet
from .. import engine
from ..graph import GraphedForward, auto_forward

log_max = log(1e2)    # VIRNet.py:15
log_min = log(1e-10)  # VIRNet.py:16


class VIRAttResUNet(nn.Module):
    """Denoising: SNet (variance map) + RNet (restorer).  Reference: VIRNet.py:18-46."""

    def __init__(self, im_chn: int, sigma_chn: int = 3, n_feat: Sequence[int] = (64, 128, 192), dep_S: int = 5,
                 n_resblocks: int = 2, noise_cond: bool = True, extra_mode: str = "Input", noise_avg: bool = False):
        super().__init__()
        self.SNet = DnCNN(im_chn, sigma_chn, dep=dep_S, noise_avg=noise_avg)
        self.noise_cond = noise_cond
        extra_chn = sigma_chn if noise_cond else 0
        self.RNet = AttResUNet(im_chn, extra_chn=extra_chn, out_chn=im_chn, n_feat=n_feat, n_resblocks=n_resblocks,
                               extra_mode=extra_mode)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """x [N,C,H,W] -> (mu [N,C,H,W], sigma [N,sigma_chn,H,W]); sigma is a variance map (VIRNet.py:42-46).

        With gradients enabled and trainable parameters the call is recorded for ``loss.backward()`` (train_denoising_syn.py:176-179):
        forward and backward both run on the HIP kernels (virnet_amd/train.py)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .. import train
            return train.denoise_forward_autograd(self, x)
        # inference: eager for the first calls of a shape, then replayed from a captured hipGraph (graph.auto_forward; VIRNET_AUTOGRAPH=0: always eager)
        return auto_forward(self, engine.denoise_forward, x)

    def graphed(self, check: str = "sync") -> GraphedForward:
        """hipGraph-replayed forward for the one-image-per-call script path; outputs are reused buffers (see GraphedForward: range guard
        around the replay, graphs dropped when a parameter changes)."""
        return GraphedForward(lambda x: engine.denoise_forward(self, x), params=self.parameters, check=check)


class VIRAttResUNetSR(nn.Module):
    """Super-resolution: SNet + KNet (kernel descriptor) + RNet on the nearest-upsampled image.  Reference: VIRNet.py:48-97."""

    def __init__(self, im_chn: int, sigma_chn: int = 1, kernel_chn: int = 3, n_feat: Sequence[int] = (64, 128, 192),
                 dep_S: int = 5, dep_K: int = 8, noise_cond: bool = True, kernel_cond: bool = True, n_resblocks: int = 1,
                 extra_mode: str = "Down", noise_avg: bool = True):
        super().__init__()
        self.noise_cond, self.noise_avg, self.kernel_cond = noise_cond, noise_avg, kernel_cond
        extra_chn = (kernel_chn if kernel_cond else 0) + (sigma_chn if noise_cond else 0)
        self.SNet = DnCNN(im_chn, sigma_chn, dep=dep_S, noise_avg=noise_avg)
        self.KNet = KNet(im_chn, kernel_chn, num_blocks=dep_K)
        self.RNet = AttResUNet(im_chn, extra_chn=extra_chn, out_chn=im_chn, n_feat=n_feat, n_resblocks=n_resblocks,
                               extra_mode=extra_mode)

    def forward(self, x: torch.Tensor, sf: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """x [N,C,h,w], sf -> (mu [N,C,h*sf,w*sf], kinfo [N,kernel_chn], sigma) (VIRNet.py:80-97).

        With gradients enabled and trainable parameters the call is recorded for ``loss.backward()`` (train_SISR.py:207-224): every
        convolution's forward and backward run on the HIP kernels (virnet_amd/train_sisr.py)."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .. import train_sisr
            return train_sisr.sisr_forward_train(self, x, sf)
        return auto_forward(self, engine.sisr_forward, x, sf, scale=int(sf) if isinstance(sf, (int, float)) and sf >= 1 else 1)

    def graphed(self, check: str = "sync") -> GraphedForward:
        """hipGraph-replayed forward: `g = net.graphed(); mu, kinfo, sigma = g(x, sf)`."""
        return GraphedForward(lambda x, sf: engine.sisr_forward(self, x, sf), params=self.parameters, check=check)
