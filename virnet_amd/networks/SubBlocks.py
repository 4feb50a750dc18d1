"""Counterpart of the reference's ``networks/SubBlocks.py:8-10`` (``conv3x3`` factory)."""
from .params import ConvParam


def conv3x3(in_chn: int, out_chn: int, bias: bool = True) -> ConvParam:
    """3x3, stride 1, pad 1 convolution parameters (reference: nn.Conv2d(k=3, s=1, p=1))."""
    return ConvParam(in_chn, out_chn, 3, bias=bias)
