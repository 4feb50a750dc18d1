"""Parameter holders.

The reference keeps its weights in ``nn.Conv2d`` / ``nn.ConvTranspose2d`` leaves; here a leaf only OWNS the
parameters (same names, shapes, dtype, default initialisation, so ``state_dict`` round-trips with the
reference's checkpoints) and caches their MFMA-stage packing.  The arithmetic runs in ``libvirnet_hip``.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from .. import ops


class ConvParam(nn.Module):
    """``weight`` (+ ``bias``) of an nn.Conv2d (OIHW) or, with ``transposed``, an nn.ConvTranspose2d (IOHW)."""

    def __init__(self, cin: int, cout: int, ks: int, bias: bool = True, transposed: bool = False, stride: int = 1):
        super().__init__()
        self.cin, self.cout, self.ks, self.transposed, self.stride = cin, cout, ks, transposed, stride
        shape = (cin, cout, ks, ks) if transposed else (cout, cin, ks, ks)
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        self.reset_parameters()
        # cached packings per conv form (the range guard's fp32 re-run must not evict the split-fp16 image): form -> (key, packing)
        self._packs: dict = {}
        self._dgrads: dict = {}

    def reset_parameters(self) -> None:
        # torch's default for _ConvNd.reset_parameters (what the reference's un-initialised convs get)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def packed(self) -> ops.PackedWeight:
        """Packed weight for the MFMA kernel, rebuilt when the parameter storage or version changed."""
        form = ops.conv_form()
        w, b = self.weight, self.bias
        key = (w.data_ptr(), w._version, w.device, None if b is None else (b.data_ptr(), b._version))
        hit = self._packs.get(form)
        if hit is None or hit[0] != key:
            hit = (key, ops.pack_weight(self.weight, self.bias, transposed=self.transposed, stride=self.stride))
            self._packs = {f: h for f, h in self._packs.items() if h[0] == key}      # (images of an older parameter version go)
            self._packs[form] = hit
        return hit[1]

    def packed_dgrad(self) -> ops.PackedWeight:
        """Packing of this layer's input-gradient GEMM (training step), cached like ``packed``."""
        form = ops.conv_form()
        key = (self.weight.data_ptr(), self.weight._version, str(self.weight.device))
        hit = self._dgrads.get(form)
        if hit is None or hit[0] != key:
            hit = (key, ops.pack_weight(self.weight, None, transposed=self.transposed, dgrad=True))
            self._dgrads = {f: h for f, h in self._dgrads.items() if h[0] == key}
            self._dgrads[form] = hit
        return hit[1]

    def packed_thin(self) -> ops.PackedWeight:
        """Packing for the bandwidth-bound few-output-channel kernel (3x3, cout <= 4), cached like ``packed``."""
        key = (self.weight.data_ptr(), self.weight._version, str(self.weight.device),
               None if self.bias is None else (self.bias.data_ptr(), self.bias._version))
        if getattr(self, "_thin", None) is None or self._thin_key != key:
            self._thin = ops.pack_thin_weight(self.weight, self.bias)
            self._thin_key = key
        return self._thin

    def _apply(self, fn, *args, **kwargs):
        # Module.to() / .cuda() / .cpu() replace ``param.data`` without touching ``_version``: tell the captured graphs (graph._EPOCH)
        from .. import graph
        graph.bump_epoch()
        return super()._apply(fn, *args, **kwargs)

    def invalidate(self) -> None:
        """Drop the cached packings.  They follow the parameter's storage and ``_version``; a write through ``.data`` (EMA helpers,
        weight clipping) changes neither, so call this (or ``net.apply(lambda m: getattr(m, 'invalidate', lambda: None)())``) after one."""
        self._packs = {}
        self._dgrads = {}
        self._thin = None
        self._cond_dgrad = None

    def forward(self, *args, **kwargs):  # pragma: no cover - guard
        raise RuntimeError("ConvParam holds parameters only; the convolution runs inside libvirnet_hip "
                           "(call the enclosing network's forward)")

    def extra_repr(self) -> str:
        kind = "convT" if self.transposed else "conv"
        return f"{kind} {self.cin}->{self.cout}, k={self.ks}, s={self.stride}, bias={self.bias is not None}"
