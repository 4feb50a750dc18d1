"""Training step of the super-resolution model on the HIP path (reference train_SISR.py:207-224).

``VIRAttResUNetSR.forward`` routes here when gradients are enabled.  Every convolution -- forward, input gradient and weight gradient
-- runs on the C-ABI kernels through small ``torch.autograd.Function`` wrappers (the same kernels the denoiser's training step uses:
``virnet_conv_mfma`` / ``virnet_conv_f16`` with the dgrad packings, ``virnet_conv_wgrad``, ``virnet_colsum``, plus
``virnet_conv_head_s4`` / ``virnet_conv_head_s4_wgrad`` for KNet's 9x9 stride-4 entry).  A residual block is one node: the SFT
modulation ``lrelu(x*mul+add)`` with per-image vectors and the plain LeakyReLUs ride in the convs' input staging, their backward in the
dgrad convs' mask epilogue or one ``virnet_sft_backward`` pass.  What is left BETWEEN the nodes -- the AttLayer / CALayer MLPs on [N, C]
vectors, global average pools, the CALayer gate, ``exp(clamp)`` / ``tanh`` -- is PyTorch device ops differentiated by autograd:
host-side tensor plumbing in BASELINE.json's sense.
So the numbers match the inference forward to fp32 noise, the step is complete (every one of the 225 parameters receives its gradient),
and torch's DistributedDataParallel hooks fire sub-module group by sub-module group as the backward proceeds (the unscale gates of
``_Gates`` are per group).

Per-pixel conditioning (``noise_avg=False``: the variance MAP feeds the head and the SFT layers, VIRNet.py:94, the JPEG variant of
train_SISR.py:87; and the denoiser with ``extra_mode`` Down / Both) takes the unfused spelling of a residual block: the AttLayer runs as
``F.linear`` over the NHWC conditioning map, the modulation ``lrelu(x*mul+add)`` as device elementwise ops under autograd, the 3x3
convolutions on the same HIP nodes -- a rare configuration (no shipped config uses it), complete rather than fast.
``denoise_forward_nodes`` runs the DENOISER through the same nodes for the configurations its fused step (train.py) does not cover.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import math

import torch
import torch.nn.functional as F

from . import _native as nat
from . import ops
from .engine import K_LOG_MIN, LOG_MAX, LOG_MIN, _ceil_to, _prep

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------------------------------
# convolutions: HIP forward / dgrad / wgrad behind autograd Functions (NHWC fp32 activations, reference-layout parameters)
# ----------------------------------------------------------------------------------------------------------------------
class _Conv3x3(torch.autograd.Function):
    """3x3 conv, stride 1 or 2, NHWC -> NHWC (AttResBlock convs, DnCNN / RB_Layer convs, DownBlock.downsampler, entry convs)."""

    @staticmethod
    def forward(ctx, x, weight, bias, conv, stride):
        x = x.contiguous()
        y, _ = ops.conv_mfma(x, conv.packed(), stride=stride, want_raw=True)
        ctx.save_for_backward(x)
        ctx.conv, ctx.stride = conv, stride
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        conv, stride = ctx.conv, ctx.stride
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            if conv.cin < 16 and x.shape[-1] == 16:          # entry conv on 16-channel records: gradient of the whole record
                dx = ops.conv_mfma(dy, conv.packed_dgrad(), want_raw=True, out_channels=32)[0][..., :16]
            elif stride == 2:
                dx = ops.conv_mfma(ops.zero_stuff2(dy), conv.packed_dgrad(), want_raw=True)[0]
            else:
                dx = ops.conv_mfma(dy, conv.packed_dgrad(), want_raw=True)[0]
        if conv.bias is not None:
            dw, db = ops.conv_wgrad(x, dy, tuple(conv.weight.shape), stride=stride, bias_channels=dy.shape[-1])
        else:
            dw, db = ops.conv_wgrad(x, dy, tuple(conv.weight.shape), stride=stride), None
        return dx, dw, db, None, None


class _ConvT2x2(torch.autograd.Function):
    """ConvTranspose2d(k=2, s=2) (UpBlock.upsampler, AttResUNet.py:80): 1x1 GEMM to 4*Cout + depth-to-space."""

    @staticmethod
    def forward(ctx, x, weight, bias, conv):
        x = x.contiguous()
        y, _ = ops.conv_mfma(x, conv.packed(), want_raw=True)
        ctx.save_for_backward(x)
        ctx.conv = conv
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        conv = ctx.conv
        dy = dy.contiguous()
        dx = ops.convt_dgrad(dy, conv.packed_dgrad()) if ctx.needs_input_grad[0] else None
        dw, db = ops.convt_wgrad(x, dy, tuple(conv.weight.shape))
        return dx, dw, db, None


class _ConvExit(torch.autograd.Function):
    """3x3 conv to a few channels with planar store and crop (AttResUNet.tail, DnCNN.conv_last, KernelNet.tail): NHWC -> NCHW."""

    @staticmethod
    def forward(ctx, x, weight, bias, conv, crop_hw):
        from .train import _thin
        x = x.contiguous()
        y = _thin(conv, x, crop_hw)
        ctx.save_for_backward(x)
        ctx.conv = conv
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        conv = ctx.conv
        n, hp, wp, _ = x.shape
        g16 = ops.pack_input(dy.contiguous(), hp, wp, zero_pad=True)           # gradient records, zero beyond the crop
        dx = ops.conv_mfma(g16, conv.packed_dgrad(), want_raw=True)[0] if ctx.needs_input_grad[0] else None
        if conv.bias is not None:
            dw, db = ops.conv_wgrad(x, g16, tuple(conv.weight.shape), bias_channels=conv.cout)
        else:
            dw, db = ops.conv_wgrad(x, g16, tuple(conv.weight.shape)), None
        return dx, dw, db, None, None


class _HeadS4(torch.autograd.Function):
    """KernelNet.head: 9x9 stride-4 conv without bias on the NCHW image (KNet.py:45,53); the image needs no gradient."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x)
        ctx.cout = weight.shape[0]
        return ops.conv_head_s4(x, weight)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return None, ops.conv_head_s4_wgrad(x, dy.contiguous(), ctx.cout)


class _PackRecords(torch.autograd.Function):
    """Entry records [N, hp, wp, 16] = [nearest-upsampled image | per-image vector repeated | 0] (VIRNet.py:83,89-92; util_net.py:20-25).
    Only the vector is differentiable: every (padded) position reads the same value, so its gradient is the sum over positions."""

    @staticmethod
    def forward(ctx, x, vec, hp, wp, sf):
        ctx.c0, ctx.ev = x.shape[1], vec.shape[1]
        return ops.pack_input(x, hp, wp, sf=sf, vec=vec.contiguous())

    @staticmethod
    def backward(ctx, drec):
        dvec = drec[..., ctx.c0:ctx.c0 + ctx.ev].sum(dim=(1, 2))
        return None, dvec, None, None, None


def _wgrad(conv, x, dy, **kw):
    """(dW, db) of a 3x3 conv with bias; db rides on the weight gradient's pass over dy."""
    if conv.bias is None:
        return ops.conv_wgrad(x, dy, tuple(conv.weight.shape), **kw), None
    return ops.conv_wgrad(x, dy, tuple(conv.weight.shape), bias_channels=dy.shape[-1], **kw)


class _ResBlockFn(torch.autograd.Function):
    """AttResBlock (AttResUNet.py:48-60) as ONE autograd node, the way the inference path and the denoiser's training step run it:
        f1  = conv1(lrelu(x*mul1 + add1))        out = x + conv2(lrelu(f1*mul2 + add2))
    The SFT modulation and the LeakyReLU are applied while the conv stages its input (``in_mul`` / ``in_add`` / ``in_slope``), the skip
    is the conv epilogue's ``res``.  Backward: weight gradients with the same staging transform; without SFT the LeakyReLU masks are
    the dgrad convs' ``mask`` epilogue; with SFT one ``virnet_sft_backward`` pass per conv turns dL/d(activated input) into dL/dx and
    the per-image dmul / dadd that autograd carries on into the AttLayer MLPs."""

    @staticmethod
    def forward(ctx, x, mul1, add1, mul2, add2, w1, b1, w2, b2, blk):
        x = x.contiguous()
        ctx.blk, ctx.sft = blk, mul1 is not None
        if ctx.sft:
            mul1, add1, mul2, add2 = (t.contiguous() for t in (mul1, add1, mul2, add2))
            f1, _ = ops.conv_mfma(x, blk.conv1.packed(), in_slope=0.2, in_mul=mul1, in_add=add1, want_raw=True)
            out, _ = ops.conv_mfma(f1, blk.conv2.packed(), in_slope=0.2, in_mul=mul2, in_add=add2, res=x, want_raw=True)
            ctx.save_for_backward(x, f1, mul1, add1, mul2, add2)
        else:
            _, f1a = ops.conv_mfma(x, blk.conv1.packed(), in_slope=0.2, want_raw=False, want_act=True, slope=0.2)
            out, _ = ops.conv_mfma(f1a, blk.conv2.packed(), res=x, want_raw=True)
            ctx.save_for_backward(x, f1a)
        return out

    @staticmethod
    def backward(ctx, dout):
        blk = ctx.blk
        dout = dout.contiguous()
        if ctx.sft:
            x, f1, mul1, add1, mul2, add2 = ctx.saved_tensors
            dw2, db2 = _wgrad(blk.conv2, f1, dout, in_slope=0.2, in_mul=mul2, in_add=add2)
            da2, _ = ops.conv_mfma(dout, blk.conv2.packed_dgrad(), want_raw=True)
            df1, dmul2, dadd2 = ops.sft_backward(da2, f1, mul2, add2, slope=0.2)
            dw1, db1 = _wgrad(blk.conv1, x, df1, in_slope=0.2, in_mul=mul1, in_add=add1)
            da1, _ = ops.conv_mfma(df1, blk.conv1.packed_dgrad(), want_raw=True)
            dx, dmul1, dadd1 = ops.sft_backward(da1, x, mul1, add1, slope=0.2, res=dout)
            return dx, dmul1, dadd1, dmul2, dadd2, dw1, db1, dw2, db2, None
        x, f1a = ctx.saved_tensors
        dw2, db2 = _wgrad(blk.conv2, f1a, dout)
        d_f1, _ = ops.conv_mfma(dout, blk.conv2.packed_dgrad(), mask=f1a, mask_slope=0.2, want_raw=True)
        dw1, db1 = _wgrad(blk.conv1, x, d_f1, in_slope=0.2)
        dx, _ = ops.conv_mfma(d_f1, blk.conv1.packed_dgrad(), mask=x, mask_slope=0.2, res=dout, want_raw=True)
        return dx, None, None, None, None, dw1, db1, dw2, db2, None


class _ActConv(torch.autograd.Function):
    """y = conv3x3(lrelu(x, slope)) with the activation applied in the conv's input staging (DnCNN's and RB_Layer's conv -> LReLU ->
    conv chains, DnCNN.py:22-29 / KNet.py:32-34, re-associated so that each node owns the activation of its INPUT)."""

    @staticmethod
    def forward(ctx, x, weight, bias, conv, slope):
        x = x.contiguous()
        y, _ = ops.conv_mfma(x, conv.packed(), in_slope=slope, want_raw=True)
        ctx.save_for_backward(x)
        ctx.conv, ctx.slope = conv, slope
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        conv, slope = ctx.conv, ctx.slope
        dy = dy.contiguous()
        dx = ops.conv_mfma(dy, conv.packed_dgrad(), mask=x, mask_slope=slope, want_raw=True)[0] if ctx.needs_input_grad[0] else None
        dw, db = _wgrad(conv, x, dy, in_slope=slope)
        return dx, dw, db, None, None


def _conv(x: Tensor, conv, stride: int = 1) -> Tensor:
    return _Conv3x3.apply(x, conv.weight, conv.bias, conv, stride)


def _act_conv(x: Tensor, conv, slope: float) -> Tensor:
    return _ActConv.apply(x, conv.weight, conv.bias, conv, slope)


def _lin(v: Tensor, conv) -> Tensor:
    """1x1 conv on per-image vectors [N, C] (AttLayer, AttResUNet.py:18-25; CALayer body, KNet.py:17-19)."""
    return F.linear(v, conv.weight.view(conv.cout, conv.cin), conv.bias)


def _att_layer(vec: Tensor, att) -> Tuple[Tensor, Tensor]:
    """AttLayer.forward on spatially constant conditioning (AttResUNet.py:27-32): (mul, add), each [N, nf]."""
    f1 = F.leaky_relu(_lin(vec, att.conv1), 0.2)
    f2 = F.leaky_relu(_lin(f1, att.conv2), 0.2)
    return torch.sigmoid(_lin(f2, att.mul_conv)), _lin(f2, att.add_conv)


def _res_block(x: Tensor, blk, vec: Optional[Tensor]) -> Tensor:
    """AttResBlock.forward (AttResUNet.py:48-60) on NHWC tensors: one fused node (the AttLayer MLPs stay autograd on [N, C] vectors).
    ``vec`` [N, e]: spatially constant conditioning; [N, h, w, e]: a per-pixel conditioning map -> the unfused spelling."""
    c1, c2 = blk.conv1, blk.conv2
    if vec is not None and blk.extra_chn > 0 and vec.dim() == 4:
        mul1, add1 = _att_layer(vec, blk.sft1)                                  # [N, h, w, nf]: 1x1 convs = F.linear over the channel axis
        f1 = _conv(F.leaky_relu(x * mul1 + add1, 0.2), c1)                      # AttResUNet.py:54-55
        mul2, add2 = _att_layer(vec, blk.sft2)
        return x + _conv(F.leaky_relu(f1 * mul2 + add2, 0.2), c2)               # AttResUNet.py:57-59
    if vec is not None and blk.extra_chn > 0:
        mul1, add1 = _att_layer(vec, blk.sft1)
        mul2, add2 = _att_layer(vec, blk.sft2)
        return _ResBlockFn.apply(x, mul1, add1, mul2, add2, c1.weight, c1.bias, c2.weight, c2.bias, blk)
    return _ResBlockFn.apply(x, None, None, None, None, c1.weight, c1.bias, c2.weight, c2.bias, blk)


# ----------------------------------------------------------------------------------------------------------------------
# sub-networks
# ----------------------------------------------------------------------------------------------------------------------
def _snet(snet, x: Tensor, rescale=None) -> Tensor:
    """DnCNN.forward (DnCNN.py:37-44) -> raw log-variance, [N,C,h,w] or pooled [N,C,1,1]."""
    n, _, h, w = x.shape
    cur = _conv(ops.pack_input(x, h, w), snet.conv1)                # pre-activations; each following conv applies the LReLU on its input
    for key in sorted(snet.mid_layer.keys(), key=int):
        cur = _act_conv(cur, snet.mid_layer[key], 0.25)
    last = snet.conv_last
    v = _ConvExit.apply(F.leaky_relu(cur, 0.25), last.weight, last.bias, last, (h, w))
    if rescale is not None:
        v = rescale(v)                                              # (in front of the mean: its backward divides by h*w)
    return v.mean(dim=(2, 3), keepdim=True) if snet.noise_avg else v


def _knet(knet, x: Tensor, rescale=None) -> Tensor:
    """KernelNet.forward (KNet.py:52-59) -> [N, 3] = (lam1, lam2, rho)."""
    k = _HeadS4.apply(x, knet.head.weight)
    for rb in knet.body:
        hcv = _act_conv(_conv(k, rb.body["0"]), rb.body["2"], 0.2)              # KNet.py:32-34 (LReLU in the second conv's staging)
        ca = rb.body["3"].body
        y = F.leaky_relu(_lin(hcv.mean(dim=(1, 2)), ca["0"]), 0.2)              # KNet.py:15-19
        gate = torch.sigmoid(_lin(y, ca["2"]))
        k = hcv * gate[:, None, None, :] + k                                    # KNet.py:26,38
    oh, ow = k.shape[1:3]
    tail = knet.tail["0"]
    m = _ConvExit.apply(k, tail.weight, tail.bias, tail, (oh, ow))
    if rescale is not None:
        m = rescale(m)
    m = m.mean(dim=(2, 3))                                                      # KNet.py:49-50
    lam12 = torch.exp(torch.clamp(m[:, :-1], min=K_LOG_MIN, max=LOG_MAX))       # KNet.py:56
    return torch.cat((lam12, torch.tanh(m[:, -1:])), dim=1)                     # KNet.py:57-58


def _rnet(rnet, x_in: Tensor, vec: Optional[Tensor], sf: int, emap: Optional[Tensor] = None) -> Tensor:
    """AttResUNet.forward (AttResUNet.py:141-175) on the nearest-upsampled image with per-image conditioning vectors ``vec`` [N, ev]
    and / or a per-pixel conditioning map ``emap`` [N, em, H, W] at the up-sampled resolution (channel order: vector, then map --
    VIRNet.py:90-95)."""
    n, c0, h0, w0 = x_in.shape
    H, W = h0 * sf, w0 * sf
    m = 1 << (rnet.depth - 1)
    hp, wp = _ceil_to(H, m), _ceil_to(W, m)
    mode = rnet.extra_mode
    feed_head, feed_down = mode in ("input", "both"), mode in ("down", "both")
    ne = (0 if vec is None else vec.shape[1]) + (0 if emap is None else emap.shape[1])
    if mode != "null" and ne != rnet.extra_chn:
        raise ValueError(f"conditioning has {ne} channels, the network was built for {rnet.extra_chn}")
    emaps = None                                                                # padded NCHW extra maps (AttResUNet.py:150)
    if emap is not None and mode != "null":
        if tuple(emap.shape[-2:]) != (H, W):
            raise ValueError(f"conditioning map is {tuple(emap.shape[-2:])}, the up-sampled image {(H, W)}")
        parts = ([vec[:, :, None, None].expand(n, vec.shape[1], H, W)] if vec is not None else []) + [emap]
        emaps = F.pad(torch.cat(parts, 1), (0, wp - W, 0, hp - H), mode="reflect") if (hp, wp) != (H, W) else torch.cat(parts, 1)
    if not feed_head:
        rec = ops.pack_input(x_in, hp, wp, sf=sf)
    elif emaps is None:
        rec = _PackRecords.apply(x_in, vec, hp, wp, sf)
    else:                                                                       # records [image | extra maps | 0], the maps differentiable
        img = ops.pack_input(x_in, hp, wp, sf=sf)
        rec = torch.cat([img[..., :c0], emaps.permute(0, 2, 3, 1), img[..., c0 + ne:]], dim=-1)
    x = _conv(rec, rnet.head)                                                   # AttResUNet.py:153-155
    bridges: List[Tensor] = []
    for ii, lvl in enumerate(rnet.down_path):
        cond = None
        if feed_down:                                                           # AttResUNet.py:158,168: nearest-resized extra maps per level
            cond = vec if emaps is None else (emaps if ii == 0 else F.interpolate(emaps, x.shape[1:3], mode="nearest")).permute(0, 2, 3, 1)
        for blk in lvl.body:
            x = _res_block(x, blk, cond)
        if ii + 1 < len(rnet.down_path):
            bridges.append(x)
            x = _conv(x, lvl.downsampler, stride=2)                             # AttResUNet.py:67,74
    for jj, up in enumerate(rnet.up_path):
        us = up.upsampler
        x = _ConvT2x2.apply(x, us.weight, us.bias, us) + bridges[-jj - 1]      # AttResUNet.py:84-87
        for blk in up.body:
            x = _res_block(x, blk, None)
    tail = rnet.tail
    out = _ConvExit.apply(x, tail.weight, tail.bias, tail, (H, W))             # AttResUNet.py:139,173 (crop)
    x_up = x_in if sf == 1 else x_in.repeat_interleave(sf, dim=2).repeat_interleave(sf, dim=3)   # VIRNet.py:83 (nearest)
    return out + x_up


# ----------------------------------------------------------------------------------------------------------------------
# loss-scale independence of the backward.  Every GEMM of the backward splits its operands into fp16 pairs, exact between 6e-5 and
# 65504; a mean-reduced loss hands over ~1e-7 per entry, a sum-reduced one or a GradScaler 1e4 and more.  The backward is linear, so the
# three incoming gradients are multiplied by ONE power of two at the network's boundary (this node sees them together) and every
# parameter gradient is divided by it again where it leaves the graph (a _Gate node at the parameters' entry).  The factor is 1 while the
# largest incoming entry lies in [2^-10, 2^10]; it is computed and applied on the device (no host read).  Encoders whose output passes a
# spatial mean get a boundary of their own in front of that mean (_Gates).
# ----------------------------------------------------------------------------------------------------------------------
class _GradScaleState:
    """The factor of ONE boundary of ONE forward (a 0-d device tensor, or None = 1): written by that boundary's _Boundary node, read by
    its _Gate node.  Never shared between forwards, modules or copies of a module."""

    def __init__(self):
        self.scale: Optional[Tensor] = None


class _Boundary(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, *outs):
        ctx.state = state
        return tuple(t.view_as(t) for t in outs)

    @staticmethod
    def backward(ctx, *douts):
        grads = [g for g in douts if g is not None]
        if not grads:
            ctx.state.scale = None
            return (None,) + douts
        amax = grads[0].detach().abs().amax()
        for g in grads[1:]:
            amax = torch.maximum(amax, g.detach().abs().amax())
        # largest entry -> [0.5, 1) unless it already lies in [2^-10, 2^10]; everything stays on the device (no host read)
        scale = torch.exp2(torch.clamp(-torch.floor(torch.log2(amax.clamp_min(1e-37))) - 1.0, -100.0, 100.0))
        keep = (amax >= 2.0 ** -10) & (amax <= 2.0 ** 10) | ~torch.isfinite(amax) | (amax <= 0)
        scale = torch.where(keep, torch.ones_like(scale), scale)
        ctx.state.scale = scale
        return (None,) + tuple(None if g is None else g * scale for g in douts)


class _Gate(torch.autograd.Function):
    """Identity on the trainable parameters at the ENTRY of a forward; its backward divides their gradients by the factor the same
    forward's _Boundary node chose (it runs first: it sits behind every use).  The unscale therefore lives inside the autograd graph of
    the forward it belongs to -- no hooks on the leaves: nothing to lose in copy.deepcopy, nothing to trip over frozen parameters,
    parameters added later are simply gated by the next forward, and two forwards folded into one backward each use their own factor
    (train_sisr.py:319)."""

    @staticmethod
    def forward(ctx, state, *params):
        ctx.state = state
        return tuple(p.view_as(p) for p in params)

    @staticmethod
    def backward(ctx, *grads):
        s = ctx.state.scale
        if s is None:
            return (None,) + grads
        inv = 1.0 / s                                                   # (a power of two: exact)
        live = [g for g in grads if g is not None]
        scaled = iter(torch._foreach_mul(live, inv)) if live else iter(())      # one multi-tensor launch for the ~225 gradients
        return (None,) + tuple(None if g is None else next(scaled) for g in grads)


class _Gates:
    """The gated parameter views of one forward: one outer boundary (the network's outputs) and one inner boundary per encoder whose
    output passes a spatial MEAN (SNet with noise_avg, KNet: their gradient chains start 1/(h*w) below the outer scale, which would push
    the split-fp16 GEMMs' operands towards fp16's subnormals -- each such chain gets its own power of two, undone by its own gate)."""

    def __init__(self, net, inner_prefixes):
        self.outer = _GradScaleState()
        named = [(k, p) for k, p in net.named_parameters() if p.requires_grad]
        # one gate per sub-module group (first three name components: "RNet.body_down.0", "SNet.conv1", ...), all sharing the state: a
        # gate's backward runs as soon as ITS consumers are done, so gradients leave the graph group by group while the backward
        # proceeds and DDP's bucket hooks fire during it (one gate over all ~225 parameters -- round 4 -- held every gradient until the
        # very end of the backward: ADVICE r04)
        groups: dict = {}
        for k, p in named:
            groups.setdefault(".".join(k.split(".")[:3]), []).append((k, p))
        views = {}
        for members in groups.values():
            views.update(zip((k for k, _ in members), _Gate.apply(self.outer, *[p for _, p in members])))
        self.inner = {}
        for pre in inner_prefixes:
            keys = [k for k in views if k.startswith(pre)]
            if keys:
                st = _GradScaleState()
                views.update(zip(keys, _Gate.apply(st, *[views[k] for k in keys])))
                self.inner[pre] = st
        self.views = views

    def rescaler(self, prefix: str):
        """Identity for the last map of sub-network ``prefix`` whose backward rescales the incoming gradient for that chain (None: no
        inner boundary for it)."""
        st = self.inner.get(prefix)
        return None if st is None else (lambda t: _Boundary.apply(st, t)[0])


class _swapped_parameters:
    """`with _swapped_parameters(net, views):` -- the modules' parameter slots hold the gated VIEWS for the duration of the forward (the
    layer code reads `module.weight`), the Parameters come back on exit.  Same idea as torch.func.functional_call, spelled with public
    attributes only (`named_modules`, `Module._parameters`) because the body executed here is not `net.forward`.  Like functional_call it
    mutates the module while the forward runs: one training forward per module at a time (concurrent forwards need one module copy each)."""

    def __init__(self, net, views: dict):
        mods = dict(net.named_modules())
        self.slots = []
        for key, view in views.items():
            owner, _, leaf = key.rpartition(".")
            self.slots.append((mods[owner]._parameters, leaf, view))

    def __enter__(self):
        from . import graph
        self.saved = [(d, leaf, d[leaf]) for d, leaf, _ in self.slots]
        for d, leaf, view in self.slots:
            d[leaf] = view
        graph.bump_epoch()               # (writes into Module._parameters fire no registration hook: captured graphs re-validate)
        return self

    def __exit__(self, *exc):
        from . import graph
        for d, leaf, p in self.saved:
            d[leaf] = p
        graph.bump_epoch()
        return False


# ----------------------------------------------------------------------------------------------------------------------
# boundary forward (networks/VIRNet.py:80-97) with gradients
# ----------------------------------------------------------------------------------------------------------------------
def sisr_forward_train(net, x: Tensor, sf) -> Tuple[Tensor, Tensor, Tensor]:
    if x.requires_grad:
        raise RuntimeError("VIRAttResUNetSR: a gradient with respect to the input image is not implemented (train_SISR.py never asks "
                           "for one); pass x.detach()")
    if int(sf) != sf or sf < 1:
        raise ValueError(f"sf must be a positive integer, got {sf}")
    sf = int(sf)
    x = _prep(x, net.SNet.in_channels)
    n = x.shape[0]
    gates = _Gates(net, (["SNet."] if net.noise_avg else []) + ["KNet."])
    # the forward below reads the modules' attributes: swap the gated views in for its duration (_swapped_parameters)
    with _swapped_parameters(net, gates.views), torch.cuda.device(x.device):
        sigma = torch.exp(torch.clamp(_snet(net.SNet, x, gates.rescaler("SNet.")), min=LOG_MIN, max=LOG_MAX))   # VIRNet.py:81
        kinfo = _knet(net.KNet, x, gates.rescaler("KNet."))                                   # VIRNet.py:82
        parts = []
        if net.kernel_cond:
            parts.append(kinfo)
        emap = None
        if net.noise_cond:
            if net.noise_avg:
                parts.append(sigma.view(n, -1).sqrt())                                        # VIRNet.py:92
            else:                                                                             # VIRNet.py:94: per-pixel variance map
                emap = sigma.sqrt()
                if sf > 1:
                    emap = emap.repeat_interleave(sf, dim=2).repeat_interleave(sf, dim=3)     # F.interpolate(nearest, x sf)
        vec = torch.cat(parts, 1) if parts else None
        mu = _rnet(net.RNet, x, vec, sf, emap)
    return _Boundary.apply(gates.outer, mu, kinfo, sigma)


def denoise_forward_nodes(net, x: Tensor) -> Tuple[Tensor, Tensor]:
    """VIRAttResUNet.forward (VIRNet.py:42-46) with gradients through the per-layer nodes of this module: the configurations the
    denoiser's fused step (train.py: one autograd Function, conditioning through the head only) does not cover -- ``extra_mode`` Down /
    Both (SFT layers fed by the per-pixel sqrt-variance map) and ``noise_avg=True``."""
    if x.requires_grad:
        raise RuntimeError("VIRAttResUNet: a gradient with respect to the input image is not implemented (the reference's training "
                           "never asks for one, train_denoising_syn.py:171-184); pass x.detach()")
    x = _prep(x, net.SNet.in_channels)
    if net.noise_cond and net.SNet.noise_avg:
        # the reference fails here too: a [N,C,1,1] map cannot be reflect-padded / concatenated (AttResUNet.py:150,153)
        raise RuntimeError("VIRAttResUNet(noise_avg=True, noise_cond=True): the [N,C,1,1] variance cannot be "
                           "padded or concatenated with the image (same failure as the reference)")
    gates = _Gates(net, ["SNet."] if net.SNet.noise_avg else [])
    with _swapped_parameters(net, gates.views), torch.cuda.device(x.device):
        sigma = torch.exp(torch.clamp(_snet(net.SNet, x, gates.rescaler("SNet.")), min=LOG_MIN, max=LOG_MAX))   # VIRNet.py:43
        mu = _rnet(net.RNet, x, None, 1, sigma.sqrt() if net.noise_cond else None)            # VIRNet.py:44-45
    return _Boundary.apply(gates.outer, mu, sigma)
