"""CPU oracle for the VIRNet convolutional forward (RNet + SNet + KNet).

TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  Nothing under ``virnet_amd/`` imports it and the product path
raises when the HIP library is missing.

It is a plain-PyTorch (CPU, fp32) functional restatement of the reference's
algorithm, written from the behaviour catalogued in SURVEY.md section 8(a) and
citing, per function, the reference file:line it follows.  The arithmetic of
the reference lives in third-party PyTorch (pinned 1.13.0+cu11.6,
``/root/reference/README.md:23``); the reference repository ships no tests and
no golden vectors, so the pin is made by us:

PARITY PIN: ``tests/golden/*.npz`` were produced by importing the reference
itself from ``/root/reference`` in the build container
(``tests/golden/make_golden.py``, committed) on deterministic weights and
inputs; ``tests/test_oracle_golden.py`` checks this file against every one of
those vectors at <= 1e-6.  The reference's own tests pin nothing.

All functions take a flat ``state_dict``-style mapping ``sd`` (reference
parameter names, reference shapes: OIHW conv, IOHW transposed conv) and NCHW
fp32 tensors.
"""
from __future__ import annotations

import math
from typing import Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Mapping[str, Tensor]

# networks/VIRNet.py:15-16
LOG_MAX = math.log(1e2)
LOG_MIN = math.log(1e-10)
# networks/KNet.py:6-7
K_LOG_MAX = math.log(1e2)
K_LOG_MIN = math.log(1e-4)


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def _conv(sd: SD, name: str, x: Tensor, stride: int = 1, padding: int = 1) -> Tensor:
    """nn.Conv2d call site; bias key may be absent (KNet.head, networks/KNet.py:45)."""
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def pad_to_multiple(x: Tensor, mod: int) -> Tensor:
    """utils/util_net.py:20-25 -- reflect-pad bottom/right up to a multiple of ``mod``."""
    h, w = x.shape[-2:]
    bottom = -h % mod
    right = -w % mod
    if bottom == 0 and right == 0:
        return x
    return F.pad(x, (0, right, 0, bottom), mode="reflect")


# --------------------------------------------------------------------------
# SNet  (networks/DnCNN.py:8-52)
# --------------------------------------------------------------------------
def dncnn(sd: SD, prefix: str, x: Tensor, dep: int, noise_avg: bool) -> Tensor:
    """DnCNN.forward, networks/DnCNN.py:37-44.

    conv1 -> LeakyReLU(0.25) -> (dep-2) x [conv, LeakyReLU(0.25)] -> conv_last
    -> Identity | AdaptiveAvgPool2d((1,1)).  The mid convs sit at even indices
    of an nn.Sequential (``mid_layer.{0,2,4,...}``, DnCNN.py:24-28).
    """
    h = F.leaky_relu(_conv(sd, prefix + "conv1", x), 0.25)
    for ii in range(dep - 2):
        h = F.leaky_relu(_conv(sd, f"{prefix}mid_layer.{2 * ii}", h), 0.25)
    h = _conv(sd, prefix + "conv_last", h)
    if noise_avg:
        h = h.mean(dim=(2, 3), keepdim=True)
    return h


# --------------------------------------------------------------------------
# KNet  (networks/KNet.py:12-59)
# --------------------------------------------------------------------------
def ca_layer(sd: SD, prefix: str, x: Tensor) -> Tensor:
    """CALayer.forward, networks/KNet.py:23-26 (body indices 0 and 2 are the 1x1 convs)."""
    y = x.mean(dim=(2, 3), keepdim=True)
    y = F.leaky_relu(_conv(sd, prefix + "body.0", y, padding=0), 0.2)
    y = torch.sigmoid(_conv(sd, prefix + "body.2", y, padding=0))
    return x * y


def rb_layer(sd: SD, prefix: str, x: Tensor) -> Tensor:
    """RB_Layer.forward, networks/KNet.py:37-39: x + CA(conv(lrelu(conv(x))))."""
    h = F.leaky_relu(_conv(sd, prefix + "body.0", x), 0.2)
    h = _conv(sd, prefix + "body.2", h)
    h = ca_layer(sd, prefix + "body.3.", h)
    return h + x


def kernel_net(sd: SD, prefix: str, x: Tensor, num_blocks: int) -> Tensor:
    """KernelNet.forward, networks/KNet.py:52-59.  Returns [N,3,1,1]."""
    h = F.conv2d(x, sd[prefix + "head.weight"], None, stride=4, padding=4)
    for b in range(num_blocks):
        h = rb_layer(sd, f"{prefix}body.{b}.", h)
    out = _conv(sd, prefix + "tail.0", h).mean(dim=(2, 3), keepdim=True)
    lam12 = torch.exp(torch.clamp(out[:, :2], min=K_LOG_MIN, max=K_LOG_MAX))
    rho = torch.tanh(out[:, -1:])
    return torch.cat((lam12, rho), dim=1)


# --------------------------------------------------------------------------
# RNet  (networks/AttResUNet.py)
# --------------------------------------------------------------------------
def att_layer(sd: SD, prefix: str, extra: Tensor) -> Tuple[Tensor, Tensor]:
    """AttLayer.forward, networks/AttResUNet.py:27-32 (SFT scale/shift generator)."""
    f1 = F.leaky_relu(_conv(sd, prefix + "conv1", extra, padding=0), 0.2)
    f2 = F.leaky_relu(_conv(sd, prefix + "conv2", f1, padding=0), 0.2)
    mul = torch.sigmoid(_conv(sd, prefix + "mul_conv", f2, padding=0))
    add = _conv(sd, prefix + "add_conv", f2, padding=0)
    return mul, add


def att_res_block(sd: SD, prefix: str, x: Tensor, extra: Optional[Tensor]) -> Tensor:
    """AttResBlock.forward, networks/AttResUNet.py:48-60 (pre-activation, optional SFT)."""
    has_sft = (prefix + "sft1.conv1.weight") in sd
    if has_sft:
        mul1, add1 = att_layer(sd, prefix + "sft1.", extra)
        a1 = x * mul1 + add1
    else:
        a1 = x
    f1 = _conv(sd, prefix + "conv1", F.leaky_relu(a1, 0.2))
    if has_sft:
        mul2, add2 = att_layer(sd, prefix + "sft2.", extra)
        a2 = f1 * mul2 + add2
    else:
        a2 = f1
    f2 = _conv(sd, prefix + "conv2", F.leaky_relu(a2, 0.2))
    return x + f2


def att_res_unet(sd: SD, prefix: str, x_in: Tensor, extra_in: Optional[Tensor],
                 depth: int, n_resblocks: int, extra_mode: str) -> Tensor:
    """AttResUNet.forward, networks/AttResUNet.py:141-175."""
    mode = extra_mode.lower()
    assert mode in ("null", "input", "down", "both")  # AttResUNet.py:113-114
    h, w = x_in.shape[-2:]
    m = 2 ** (depth - 1)
    x = pad_to_multiple(x_in, m)
    extra = pad_to_multiple(extra_in, m) if mode != "null" else None
    if mode in ("input", "both"):
        x = _conv(sd, prefix + "head", torch.cat([x, extra], 1))
    else:
        x = _conv(sd, prefix + "head", x)
    use_down = mode in ("down", "both")
    bridges = []
    extra_lvl = extra if use_down else None
    for ii in range(depth):
        for jj in range(n_resblocks):
            x = att_res_block(sd, f"{prefix}down_path.{ii}.body.{jj}.", x, extra_lvl)
        if ii + 1 < depth:
            bridges.append(x)
            x = _conv(sd, f"{prefix}down_path.{ii}.downsampler", x, stride=2, padding=1)
            if use_down:
                # AttResUNet.py:168 -- always resized from the FULL-res padded extra maps
                extra_lvl = F.interpolate(extra, x.shape[-2:], mode="nearest")
    for jj in range(depth - 1):
        up = f"{prefix}up_path.{jj}."
        x = F.conv_transpose2d(x, sd[up + "upsampler.weight"], sd[up + "upsampler.bias"], stride=2)
        x = x + bridges[-jj - 1]                       # AttResUNet.py:85-89 (block 0 only)
        for kk in range(n_resblocks):
            x = att_res_block(sd, f"{up}body.{kk}.", x, None)
    out = _conv(sd, prefix + "tail", x)[..., :h, :w] + x_in
    return out


# --------------------------------------------------------------------------
# boundary forwards (networks/VIRNet.py)
# --------------------------------------------------------------------------
def virnet_denoise(sd: SD, x: Tensor, *, n_feat: Sequence[int], dep_S: int, n_resblocks: int,
                   noise_cond: bool = True, extra_mode: str = "Input",
                   noise_avg: bool = False) -> Tuple[Tensor, Tensor]:
    """VIRAttResUNet.forward, networks/VIRNet.py:42-46.  Returns (mu, sigma)."""
    sigma = torch.exp(torch.clamp(dncnn(sd, "SNet.", x, dep_S, noise_avg), min=LOG_MIN, max=LOG_MAX))
    extra = sigma.sqrt() if noise_cond else None
    mu = att_res_unet(sd, "RNet.", x, extra, len(n_feat), n_resblocks, extra_mode)
    return mu, sigma


def virnet_sisr(sd: SD, x: Tensor, sf: int, *, n_feat: Sequence[int], dep_S: int, dep_K: int,
                n_resblocks: int, noise_cond: bool = True, kernel_cond: bool = True,
                extra_mode: str = "Down", noise_avg: bool = True) -> Tuple[Tensor, Tensor, Tensor]:
    """VIRAttResUNetSR.forward, networks/VIRNet.py:80-97.  Returns (mu, kinfo[N,3], sigma)."""
    sigma = torch.exp(torch.clamp(dncnn(sd, "SNet.", x, dep_S, noise_avg), min=LOG_MIN, max=LOG_MAX))
    kinfo = kernel_net(sd, "KNet.", x, dep_K)
    x_up = F.interpolate(x, scale_factor=sf, mode="nearest")
    h_up, w_up = x_up.shape[-2:]
    parts = []
    if kernel_cond:
        parts.append(kinfo.repeat(1, 1, h_up, w_up))
    if noise_cond:
        if noise_avg:
            parts.append(sigma.sqrt().repeat(1, 1, h_up, w_up))
        else:
            parts.append(F.interpolate(sigma.sqrt(), scale_factor=sf, mode="nearest"))
    extra = torch.cat(parts, 1) if parts else None
    mu = att_res_unet(sd, "RNet.", x_up, extra, len(n_feat), n_resblocks, extra_mode)
    return mu, kinfo.squeeze(-1).squeeze(-1), sigma


# --------------------------------------------------------------------------
# per-kernel oracles: the semantics of each HIP entry point in NCHW torch ops
# (used by tests/test_ops_gpu.py).  They restate the same reference call sites.
# --------------------------------------------------------------------------
def conv_fused(x: Tensor, w: Tensor, b: Optional[Tensor], *, stride: int = 1,
               residual: Optional[Tensor] = None, mul: Optional[Tensor] = None,
               add: Optional[Tensor] = None, slope: float = 0.2) -> Tuple[Tensor, Tensor]:
    """3x3 conv + bias (+residual) -> raw ; act = lrelu(raw*mul+add, slope).

    raw is what AttResUNet.py:59 / :67 produce; act is the operand the NEXT
    pre-activation conv consumes (AttResUNet.py:55,58).  mul/add are [N,C,1,1] or None.
    """
    raw = F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2)
    if residual is not None:
        raw = raw + residual
    a = raw
    if mul is not None:
        a = a * mul
    if add is not None:
        a = a + add
    return raw, F.leaky_relu(a, slope)


def conv_transpose_fused(x: Tensor, w: Tensor, b: Tensor, bridge: Tensor,
                         slope: float = 0.2) -> Tuple[Tensor, Tensor]:
    """UpBlock head, networks/AttResUNet.py:84-87: ConvT(k2,s2)(x) + bridge -> raw, lrelu(raw)."""
    raw = F.conv_transpose2d(x, w, b, stride=2) + bridge
    return raw, F.leaky_relu(raw, slope)
