"""Forward orchestration: which kernel runs when, with which fused epilogue.

Every pre-activation conv of the reference (``conv(lrelu(x*mul+add))``, networks/AttResUNet.py:55,58) reads a tensor that
its PRODUCER already stored activated: each MFMA conv can store ``raw`` (what residual adds and the stride-2 / transposed
convs consume) and/or ``act = lrelu(raw*mul+add)`` (what the next 3x3 consumes).  So the conv kernel itself sees plain
zero-padded input -- "pad after activation" holds by construction -- and no elementwise kernel runs between convs.

Python here only sequences launches on torch's current stream and owns the buffers; there is no CPU path.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import _native as nat
from . import ops

Tensor = torch.Tensor

# networks/VIRNet.py:15-16 and networks/KNet.py:6-7
LOG_MAX = 4.605170185988092
LOG_MIN = -23.025850929940457
K_LOG_MIN = -9.210340371976182


def _prep(x: Tensor, chn: int, name: str = "x") -> Tensor:
    if x.dim() != 4:
        raise ValueError(f"{name} must be [N,C,H,W], got {tuple(x.shape)}")
    if x.shape[1] != chn:
        raise ValueError(f"{name} has {x.shape[1]} channels, the network was built for {chn}")
    if not x.is_cuda:
        raise RuntimeError(f"{name} is on {x.device}: the VIRNet HIP path runs on a ROCm device only (no CPU fallback)")
    if x.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (the reference is fp32 end to end), got {x.dtype}")
    return x.detach().contiguous()


def _ceil_to(v: int, m: int) -> int:
    return (v + m - 1) // m * m


# ----------------------------------------------------------------------------------------------------------------
# SNet (reference networks/DnCNN.py:37-44)
# ----------------------------------------------------------------------------------------------------------------
def snet_forward(snet, x: Tensor, mode: str = "raw") -> Tensor:
    """mode 'raw': DnCNN.forward output; mode 'sigma': exp(clamp(.)) of it (VIRNet.py:43), fused where possible."""
    x = _prep(x, snet.in_channels)
    n, _, h, w = x.shape
    rec = ops.pack_input(x, h, w)
    _, cur = ops.conv_mfma(rec, snet.conv1.packed(), want_raw=False, want_act=True, slope=0.25)
    for key in sorted(snet.mid_layer.keys(), key=int):
        _, cur = ops.conv_mfma(cur, snet.mid_layer[key].packed(), want_raw=False, want_act=True, slope=0.25)
    last = snet.conv_last.packed()
    if snet.noise_avg:
        raw = ops.conv_mfma_nchw(cur, last, (h, w))
        fin = ops.GAP_EXPCLAMP if mode == "sigma" else ops.GAP_MEAN
        return ops.gap_nchw(raw, fin, (LOG_MIN, LOG_MAX)).view(n, -1, 1, 1)
    if mode == "sigma":
        return ops.conv_mfma_nchw(cur, last, (h, w), op=nat.NCHW_EXPCLAMP, clamp=(LOG_MIN, LOG_MAX))
    return ops.conv_mfma_nchw(cur, last, (h, w))


# ----------------------------------------------------------------------------------------------------------------
# KNet (reference networks/KNet.py:52-59)
# ----------------------------------------------------------------------------------------------------------------
def knet_forward(knet, x: Tensor) -> Tensor:
    x = _prep(x, knet.in_nc)
    n, _, h, w = x.shape
    cur = ops.conv_head_s4(x, knet.head.weight)                                  # KNet.py:45,53 -> NHWC raw
    for rb in knet.body:
        _, a = ops.conv_mfma(cur, rb.body["0"].packed(), want_raw=False, want_act=True, slope=0.2)   # KNet.py:32-33
        hcv, _ = ops.conv_mfma(a, rb.body["2"].packed(), want_raw=True)                               # KNet.py:34
        ca = rb.body["3"].body
        gate = ops.ca_gate(hcv, ca["0"].weight, ca["0"].bias, ca["2"].weight, ca["2"].bias)          # KNet.py:15-25
        cur = ops.scale_add(hcv, gate, cur)                                                           # KNet.py:26,38
    oh, ow = cur.shape[1:3]
    raw = ops.conv_mfma_nchw(cur, knet.tail["0"].packed(), (oh, ow))                                  # KNet.py:49
    return ops.gap_nchw(raw, ops.GAP_KINFO, (K_LOG_MIN, LOG_MAX)).view(n, -1, 1, 1)                  # KNet.py:50,56-59


# ----------------------------------------------------------------------------------------------------------------
# RNet (reference networks/AttResUNet.py:141-175)
# ----------------------------------------------------------------------------------------------------------------
class _Cond:
    """Conditioning seen by the SFT layers of the down path: per-image vector or per-pixel records."""

    def __init__(self, vec: Optional[Tensor], rec: Optional[Tensor], chan0: int, nchan: int):
        self.vec, self.rec, self.chan0, self.nchan = vec, rec, chan0, nchan

    def params(self, att) -> Tuple[Optional[Tensor], Optional[Tensor]]:
        if self.vec is None:
            return None, None
        return ops.sft_vec(self.vec, att)


def _produce(x: Tensor, conv, cond: Optional[_Cond], att, level: int, *, stride: int = 1, res: Optional[Tensor] = None,
             want_raw: bool, want_act: bool = True) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """Run one MFMA conv and deliver (raw, act) where act is what the next pre-activation conv reads.

    ``att`` is the AttLayer of the CONSUMER of ``act`` (None -> plain LeakyReLU(0.2))."""
    pw = conv.packed()
    if not want_act:
        return ops.conv_mfma(x, pw, stride=stride, res=res, want_raw=True, want_act=False)
    if att is None or cond is None:
        return ops.conv_mfma(x, pw, stride=stride, res=res, want_raw=want_raw, want_act=True, slope=0.2)
    if cond.vec is not None:   # spatially constant conditioning: SFT collapses to per-(image, channel) scale/shift
        mul, add = cond.params(att)
        return ops.conv_mfma(x, pw, stride=stride, res=res, mul=mul, add=add, want_raw=want_raw, want_act=True, slope=0.2)
    raw, _ = ops.conv_mfma(x, pw, stride=stride, res=res, want_raw=True, want_act=False)
    act = ops.sft_apply(raw, cond.rec, cond.chan0, cond.nchan, 1 << level, att)
    return (raw if want_raw else None), act


def rnet_forward(rnet, x_in: Tensor, *, extra_map: Optional[Tensor] = None, extra_vec: Optional[Tensor] = None,
                 sf: int = 1, map_sf: int = 1, map_sqrt: bool = False) -> Tensor:
    """AttResUNet.forward.  ``x_in`` may be the low-resolution image with ``sf`` > 1: the nearest up-sampling of
    VIRNet.py:83 is then fused into the entry kernel and into the final ``+ x_in``."""
    x_in = _prep(x_in, rnet.in_chn)
    mode = rnet.extra_mode
    n, _, h0, w0 = x_in.shape
    H, W = h0 * sf, w0 * sf
    m = 1 << (rnet.depth - 1)
    Hp, Wp = _ceil_to(H, m), _ceil_to(W, m)
    ev = 0 if extra_vec is None else extra_vec.shape[1]
    em = 0 if extra_map is None else extra_map.shape[1]
    if mode != "null":
        if ev + em != rnet.extra_chn:
            raise ValueError(f"conditioning has {ev + em} channels, the network was built for {rnet.extra_chn}")
        if extra_map is not None:
            extra_map = _prep(extra_map, em, "extra maps")
        if extra_vec is not None:
            extra_vec = extra_vec.detach().contiguous()
    feed_head = mode in ("input", "both")
    feed_down = mode in ("down", "both")
    rec = ops.pack_input(x_in, Hp, Wp, sf=sf, vec=extra_vec if feed_head else None,
                         map_=extra_map if feed_head else None, map_sf=map_sf, map_sqrt=map_sqrt)
    cond = None
    if feed_down:
        if extra_map is None:
            cond = _Cond(extra_vec, None, 0, ev)
        else:  # per-pixel conditioning: keep full-resolution padded records of the extra channels (AttResUNet.py:158,168)
            crec = rec if feed_head else ops.pack_input(x_in, Hp, Wp, sf=sf, vec=extra_vec, map_=extra_map,
                                                        map_sf=map_sf, map_sqrt=map_sqrt)
            cond = _Cond(None, crec, rnet.in_chn, ev + em)

    def sft(block, which):
        return getattr(block, which) if (cond is not None and block.extra_chn > 0) else None

    down = rnet.down_path
    first = down[0].body[0] if len(down[0].body) else None
    x_raw, x_act = _produce(rec, rnet.head, cond, sft(first, "sft1") if first is not None else None, 0,
                            want_raw=True, want_act=first is not None)
    bridges: List[Tensor] = []
    for ii, lvl in enumerate(down):
        nb = len(lvl.body)
        for jj, blk in enumerate(lvl.body):
            _, f1a = _produce(x_act, blk.conv1, cond, sft(blk, "sft2"), ii, want_raw=False)
            nxt = lvl.body[jj + 1] if jj + 1 < nb else None
            x_raw, x_act = _produce(f1a, blk.conv2, cond, sft(nxt, "sft1") if nxt is not None else None, ii, res=x_raw,
                                    want_raw=True, want_act=nxt is not None)
        if ii + 1 < len(down):
            bridges.append(x_raw)
            nlvl = down[ii + 1]
            nfirst = nlvl.body[0] if len(nlvl.body) else None
            x_raw, x_act = _produce(x_raw, lvl.downsampler, cond, sft(nfirst, "sft1") if nfirst is not None else None,
                                    ii + 1, stride=2, want_raw=True, want_act=nfirst is not None)
    for jj, up in enumerate(rnet.up_path):
        nb = len(up.body)
        x_raw, x_act = ops.conv_mfma(x_raw, up.upsampler.packed(), res=bridges[-jj - 1], want_raw=True, want_act=nb > 0,
                                     slope=0.2)
        for kk, blk in enumerate(up.body):
            _, f1a = ops.conv_mfma(x_act, blk.conv1.packed(), want_raw=False, want_act=True, slope=0.2)
            x_raw, x_act = ops.conv_mfma(f1a, blk.conv2.packed(), res=x_raw, want_raw=True, want_act=kk + 1 < nb, slope=0.2)
    return ops.conv_mfma_nchw(x_raw, rnet.tail.packed(), (H, W), op=nat.NCHW_ADD, res=x_in, res_sf=sf)


# ----------------------------------------------------------------------------------------------------------------
# boundary forwards (reference networks/VIRNet.py:42-46 and :80-97)
# ----------------------------------------------------------------------------------------------------------------
def denoise_forward(net, x: Tensor) -> Tuple[Tensor, Tensor]:
    x = _prep(x, net.SNet.in_channels)
    sigma = snet_forward(net.SNet, x, mode="sigma")
    if net.noise_cond:
        if net.SNet.noise_avg:
            # the reference fails here too: a [N,C,1,1] map cannot be reflect-padded / concatenated (AttResUNet.py:150,153)
            raise RuntimeError("VIRAttResUNet(noise_avg=True, noise_cond=True): the [N,C,1,1] variance cannot be "
                               "padded or concatenated with the image (same failure as the reference)")
        mu = rnet_forward(net.RNet, x, extra_map=sigma, map_sqrt=True)        # sqrt fused into the entry kernel (VIRNet.py:44)
    else:
        mu = rnet_forward(net.RNet, x)
    return mu, sigma


def sisr_forward(net, x: Tensor, sf) -> Tuple[Tensor, Tensor, Tensor]:
    x = _prep(x, net.SNet.in_channels)
    if int(sf) != sf or sf < 1:
        raise ValueError(f"sf must be a positive integer, got {sf}")
    sf = int(sf)
    n = x.shape[0]
    sigma = snet_forward(net.SNet, x, mode="sigma")            # [N,s,1,1] with noise_avg else [N,s,h,w]
    kinfo = knet_forward(net.KNet, x)                          # [N,k,1,1]
    vec_parts, emap = [], None
    if net.kernel_cond:
        vec_parts.append(kinfo.view(n, -1))
    if net.noise_cond:
        if net.noise_avg:
            vec_parts.append(sigma.view(n, -1).sqrt())         # [N,s] glue on a few floats (VIRNet.py:92)
        else:
            emap = sigma                                       # nearest x sf + sqrt fused in the entry kernel (VIRNet.py:94)
    vec = torch.cat(vec_parts, 1).contiguous() if vec_parts else None
    mu = rnet_forward(net.RNet, x, extra_map=emap, extra_vec=vec, sf=sf, map_sf=sf, map_sqrt=True)
    return mu, kinfo.view(n, -1), sigma
