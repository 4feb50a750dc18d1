"""Training step of the denoiser on the HIP path (reference train_denoising_syn.py:176-184).

``VIRAttResUNet.forward`` routes here when gradients are enabled and a parameter requires them: the whole network is ONE
``torch.autograd.Function`` whose forward is the same kernel sequence as inference (keeping the tensors the backward needs) and
whose backward walks the layers in reverse with hand-written kernels only:

  * input gradients  : ``virnet_conv_mfma`` with the layer's dgrad packing (flipped/transposed 3x3 taps; the stride-2 conv on the
    zero-stuffed gradient; the transposed conv as a pointwise GEMM over the space-to-depth gradient), the LeakyReLU derivative and
    the residual / bridge gradient fused into the epilogue (``mask``, ``res``);
  * weight gradients : ``virnet_conv_wgrad`` (MFMA, contraction over pixels) with the forward conv's staging transform;
  * bias gradients   : ``virnet_colsum``.

The ELBO itself (loss/ELBO_simple.py) stays host-side PyTorch on ``mu`` and ``sigma`` as BASELINE.json's north_star asks; autograd
hands its gradients to ``backward`` below.  The SISR step and the denoiser configurations with SFT conditioning (extra_mode Down / Both)
run through per-layer nodes instead: virnet_amd/train_sisr.py.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import _native as nat
from . import ops
from .engine import LOG_MAX, LOG_MIN, _ceil_to, _prep

Tensor = torch.Tensor


class _Tape:
    """What the backward needs, in forward order."""

    def __init__(self):
        self.snet: dict = {}
        self.blocks: List[dict] = []        # residual blocks in execution order
        self.misc: dict = {}


def _thin(conv, x, crop, **kw):
    if ops._f16_family() and conv.cout <= 32:
        return ops.conv_f16_nchw(x, conv.packed(), crop, **kw)
    return ops.conv3x3_thin(x, conv.packed_thin(), crop, **kw) if conv.cout <= 4 else ops.conv_mfma_nchw(x, conv.packed(), crop, **kw)


def _conv3(x, pw, emit, **kw):
    """ops.conv_mfma -> (raw, act, T image or None) whether or not an emission was asked for."""
    if emit is None:
        return ops.conv_mfma(x, pw, **kw) + (None,)
    return ops.conv_mfma(x, pw, emit=emit, **kw)


def denoise_forward_train(net, x: Tensor) -> Tuple[Tensor, Tensor, _Tape]:
    snet, rnet = net.SNet, net.RNet
    if snet.noise_avg:
        raise NotImplementedError("fused denoiser step: noise_avg=True runs through train_sisr.denoise_forward_nodes (denoise_forward_autograd routes it)")
    if rnet.extra_mode not in ("input", "null"):
        raise NotImplementedError("fused denoiser step: extra_mode Down / Both runs through train_sisr.denoise_forward_nodes (denoise_forward_autograd routes it)")
    x = _prep(x, snet.in_channels)
    n, _, h, w = x.shape
    tape = _Tape()
    # T emission: a conv whose output is the (staged) input of a stride-1 3x3 conv writes that conv's weight-gradient operand image
    # from its own epilogue (ops.conv_mfma(emit=...)); `None` in the tape = not emitted, the backward re-lays the tensor (virnet_chsplit)
    PLAIN, ACT02 = dict(act=None, colsum=None), dict(act=0.2, colsum=None)
    if _side_stream(x.device) is not None:                # (images would cross streams and pools: the second-stream mode re-lays its operands)
        PLAIN = ACT02 = None
    # ---- SNet (networks/DnCNN.py:37-44)
    rec_s = ops.pack_input(x, h, w)
    acts, acts_t = [], []
    _, cur, t = _conv3(rec_s, snet.conv1.packed(), PLAIN, want_raw=False, want_act=True, slope=0.25)
    acts.append(cur); acts_t.append(t)
    mids = [snet.mid_layer[k] for k in sorted(snet.mid_layer.keys(), key=int)]
    for conv in mids:
        _, cur, t = _conv3(cur, conv.packed(), PLAIN, want_raw=False, want_act=True, slope=0.25)
        acts.append(cur); acts_t.append(t)
    sigma = _thin(snet.conv_last, cur, (h, w), op=nat.NCHW_EXPCLAMP, clamp=(LOG_MIN, LOG_MAX))
    tape.snet = dict(rec=rec_s, acts=acts, acts_t=acts_t, mids=mids, sigma=sigma)
    # ---- RNet (networks/AttResUNet.py:141-175)
    m = 1 << (rnet.depth - 1)
    hp, wp = _ceil_to(h, m), _ceil_to(w, m)
    cond = net.noise_cond and rnet.extra_mode == "input"
    rec = ops.pack_input(x, hp, wp, map_=sigma if cond else None, map_sqrt=True)
    # what consumes a residual-stream tensor decides the image its producer emits: a block's conv1 stages lrelu(x, 0.2) (AttResUNet.py:55),
    # the tail the raw tensor (:173); the stride-2 / transposed convs' weight gradients take other layouts (no emission)
    down_levels = [list(lvl.body) for lvl in rnet.down_path]
    up_levels = [list(up.body) for up in rnet.up_path]
    xcur, _, xcur_t = _conv3(rec, rnet.head.packed(), ACT02 if down_levels[0] else None, want_raw=True)
    bridges = []
    order = []                 # ("block", blk, x_in, (f1a, T(lrelu x_in), T(f1a))) / ("down", conv, x_in, None) / ("up", conv, x_in, bridge index)

    def block(blk, xin, xin_t, next_spec):
        _, f1a, f1a_t = _conv3(xin, blk.conv1.packed(), PLAIN, in_slope=0.2, want_raw=False, want_act=True, slope=0.2)
        out, _, out_t = _conv3(f1a, blk.conv2.packed(), next_spec if PLAIN is not None else None, res=xin, want_raw=True)
        order.append(("block", blk, xin, (f1a, xin_t, f1a_t)))
        return out, out_t

    nlev = len(down_levels)
    for ii, body in enumerate(down_levels):
        for bi, blk in enumerate(body):
            # (the bottom level's last block feeds the first transposed conv: its weight gradient takes the plain image as low-res operand)
            xcur, xcur_t = block(blk, xcur, xcur_t, ACT02 if bi + 1 < len(body) else (PLAIN if ii + 1 == nlev and rnet.up_path else None))
        if ii + 1 < len(rnet.down_path):
            bridges.append(xcur)
            out, _ = ops.conv_mfma(xcur, rnet.down_path[ii].downsampler.packed(), stride=2, want_raw=True)
            order.append(("down", rnet.down_path[ii].downsampler, xcur, None))
            xcur, xcur_t = out, None
    for jj, up in enumerate(rnet.up_path):
        out, _ = ops.conv_mfma(xcur, up.upsampler.packed(), res=bridges[-jj - 1], want_raw=True)
        order.append(("up", up.upsampler, xcur, (len(bridges) - 1 - jj, xcur_t)))
        xcur, xcur_t = out, None
        body = up_levels[jj]
        for bi, blk in enumerate(body):
            last = bi + 1 == len(body)
            xcur, xcur_t = block(blk, xcur, xcur_t, PLAIN if last else ACT02)      # (last: the next transposed conv or the tail take the plain image)
    mu = _thin(rnet.tail, xcur, (h, w), op=nat.NCHW_ADD, res=x)
    tape.misc = dict(rec=rec, x_last=xcur, x_last_t=xcur_t, order=order, nbridges=len(bridges), hw=(h, w), hpwp=(hp, wp), cond=cond)
    return mu, sigma, tape


_SIDE: Dict[int, "torch.cuda.Stream"] = {}


def _side_stream(device: torch.device) -> Optional["torch.cuda.Stream"]:
    """Optional second HIP stream for the weight-gradient kernels (VIRNET_WGRAD_STREAM=1): they only depend on tensors the
    input-gradient chain has already produced, so they can run beside the next layers' dgrad kernels.  OFF by default since the weight
    gradients moved to the f16 pipe: every kernel of the step now fills the chip on its own, one stream measures 1 006-1 010 img/s run
    after run, two streams 1 016-1 030 at best and 635-870 when the host runs ahead of the device (tensors handed across streams
    pin their blocks in the caching allocator until the device catches up, and the step starts paying for fresh allocations)."""
    import os
    if os.environ.get("VIRNET_WGRAD_STREAM", "0") != "1":
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _SIDE:
        _SIDE[idx] = torch.cuda.Stream(device=idx)
    return _SIDE[idx]


def _wgrad_pair(conv, x_in: Tensor, dy: Tensor, stride: int, in_slope: Optional[float], cvalid: Optional[int], xt=None, yt=None) -> Dict:
    """{weight: dW[, bias: db]} of one conv; the bias gradient rides on the weight-gradient's pass over dy where it can.
    ``xt`` / ``yt``: operand images a conv's epilogue already emitted (ops.TImage), returned to the pool here."""
    kw = dict(xt=xt, yt=yt) if (stride == 1 and (xt is not None or yt is not None)) else (dict(yt=yt) if stride == 2 and yt is not None else {})
    try:
        if conv.bias is None:
            return {conv.weight: ops.conv_wgrad(x_in, dy, tuple(conv.weight.shape), stride=stride, in_slope=in_slope, **kw)}
        dw, db = ops.conv_wgrad(x_in, dy, tuple(conv.weight.shape), stride=stride, in_slope=in_slope,
                                bias_channels=conv.cout if cvalid is None else cvalid, **kw)
        return {conv.weight: dw, conv.bias: db}
    finally:
        ops.t_release(xt)
        ops.t_release(yt)


def _conv_grads(grads: Dict, conv, x_in: Tensor, dy: Tensor, *, stride: int = 1, in_slope: Optional[float] = None,
                cvalid: Optional[int] = None, reducer=None, xt=None, yt=None) -> None:
    """dW, db of one conv from its forward input and output gradient (NHWC); handed to the gradient reducer at once (DDP runs)."""
    side = _side_stream(dy.device)
    if side is None:
        new = _wgrad_pair(conv, x_in, dy, stride, in_slope, cvalid, xt, yt)
        if reducer is not None:
            reducer.push(new)
    else:
        main = torch.cuda.current_stream(dy.device)
        side.wait_stream(main)                              # dy (and x_in) are complete on the main stream up to here
        with torch.cuda.stream(side):
            new = _wgrad_pair(conv, x_in, dy, stride, in_slope, cvalid, xt, yt)
            if reducer is not None:
                reducer.push(new)                           # bucket copies + all-reduce start behind the wgrad kernels, on THEIR stream
        for t in (x_in, dy):
            t.record_stream(side)                           # the caching allocator must not recycle them under the side stream
        for g in new.values():
            g.record_stream(main)                           # allocated on the side stream, consumed (after the join) on the main one
    grads.update(new)


def _head_cond_dgrad(conv, in_chn: int, nc: int):
    """Packing of the input-gradient conv of the head restricted to its ``nc`` conditioning channels (AttResUNet.py:153: the record is
    [image | sqrt(sigma) | 0]; the image needs no gradient): W'[c][co][ky][kx] = W[co][in_chn + c][2-ky][2-kx], a features -> nc exit conv.
    None when the form has no exit kernel for it (fp32 forms, nc * 9 > 32).  Cached on the layer like its other packings."""
    if not ops._f16_family() or nc * 9 > 32 or conv.cout % 16:
        return None
    key = (conv.weight.data_ptr(), conv.weight._version, str(conv.weight.device), ops.conv_form(), in_chn, nc)
    hit = getattr(conv, "_cond_dgrad", None)
    if hit is None or hit[0] != key:
        wd = conv.weight.detach()[:, in_chn:in_chn + nc].flip(2, 3).permute(1, 0, 2, 3).contiguous()
        hit = (key, ops.pack_weight(wd, None))
        conv._cond_dgrad = hit
    pw = hit[1]
    return pw if pw.f16 is not None else None


def denoise_backward(net, tape: _Tape, dmu: Optional[Tensor], dsigma: Optional[Tensor], reducer=None) -> Dict:
    snet, rnet = net.SNet, net.RNet
    grads: Dict = {}
    h, w = tape.misc["hw"]
    hp, wp = tape.misc["hpwp"]
    sigma = tape.snet["sigma"]
    n = sigma.shape[0]
    d_sigma_total = torch.zeros_like(sigma) if dsigma is None else dsigma.detach().contiguous().clone()
    if dmu is not None:
        dmu = dmu.detach().contiguous()
        # ---- tail: mu = conv(x_last)[crop] + x_in  (AttResUNet.py:173)
        g16 = ops.pack_input(dmu, hp, wp, zero_pad=True)                       # gradient record, zero beyond the crop
        _conv_grads(grads, rnet.tail, tape.misc["x_last"], g16, reducer=reducer, xt=tape.misc.get("x_last_t"))
        # every gradient tensor that is the dY of a stride-1 conv's weight gradient leaves its producer with that operand image and
        # its channel sums (the bias gradient): dx_t travels with dx
        order = tape.misc["order"]

        emitting = _side_stream(dmu.device) is None

        def dy_spec(c):
            return dict(act=None, colsum=c) if emitting else None

        def emit_for(idx):          # the consumer of the gradient that flows INTO order[idx] (None: no stride-1 weight gradient takes it as dY)
            if idx < 0:
                return dy_spec(rnet.head.cout)                                  # the head's weight gradient
            if order[idx][0] == "block":
                return dy_spec(order[idx][1].conv2.cout)
            return dy_spec(order[idx][1].cout) if order[idx][0] == "down" else None     # (stride-2 conv: its low-res operand; transposed: column-phase layout)

        last = len(order) - 1
        dx, _, dx_t = _conv3(g16, rnet.tail.packed_dgrad(), emit_for(last), want_raw=True)
        nb = tape.misc["nbridges"]
        dbridge: List[Optional[Tensor]] = [None] * nb
        for oi in range(last, -1, -1):
            kind, mod, x_in, aux = order[oi]
            nxt = emit_for(oi - 1)
            if kind == "block":                                                 # AttResBlock, AttResUNet.py:48-60
                f1a, xin_t, f1a_t = aux
                _conv_grads(grads, mod.conv2, f1a, dx, reducer=reducer, xt=f1a_t, yt=dx_t)
                d_f1, _, d_f1_t = _conv3(dx, mod.conv2.packed_dgrad(), dy_spec(mod.conv1.cout), mask=f1a, mask_slope=0.2, want_raw=True)
                _conv_grads(grads, mod.conv1, x_in, d_f1, in_slope=0.2, reducer=reducer, xt=xin_t, yt=d_f1_t)
                dx, _, dx_t = _conv3(d_f1, mod.conv1.packed_dgrad(), nxt, mask=x_in, mask_slope=0.2, res=dx, want_raw=True)
            elif kind == "up":                                                  # UpBlock.upsampler + bridge, AttResUNet.py:84-87
                ops.t_release(dx_t); dx_t = None
                aux, xin_t = aux
                dbridge[aux] = dx
                side = _side_stream(dx.device)
                main = torch.cuda.current_stream(dx.device)
                if side is not None:
                    side.wait_stream(main)
                with torch.cuda.stream(side if side is not None else main):        # same stream as every other push (bucket order)
                    dw, db = ops.convt_wgrad(x_in, dx, tuple(mod.weight.shape), xt=xin_t)
                    ops.t_release(xin_t)
                    new = {mod.weight: dw, mod.bias: db}
                    if reducer is not None:
                        reducer.push(new)
                if side is not None:
                    for t in (x_in, dx):
                        t.record_stream(side)
                    for g in new.values():
                        g.record_stream(main)
                grads.update(new)
                dx = ops.convt_dgrad(dx, mod.packed_dgrad())                    # (the stride-2 kernel does not emit: the next block re-lays this one)
            else:                                                               # DownBlock.downsampler, AttResUNet.py:67,74
                _conv_grads(grads, mod, x_in, dx, stride=2, reducer=reducer, yt=dx_t)   # (dx_t: its low-resolution operand + bias sums)
                dx_t = None
                nb -= 1
                dx, _, dx_t = _conv3(ops.zero_stuff2(dx), mod.packed_dgrad(), nxt, res=dbridge[nb], want_raw=True)
        # ---- head (AttResUNet.py:153-155): weights, and the gradient flowing into sqrt(sigma) through the conditioning channel
        rec = tape.misc["rec"]
        _conv_grads(grads, rnet.head, rec, dx, reducer=reducer, yt=dx_t)
        if tape.misc["cond"]:
            nc = sigma.shape[1]
            pw = _head_cond_dgrad(rnet.head, rnet.in_chn, nc)
            if pw is not None:
                # only the conditioning channels of the record gradient are needed: a features -> nc-channel exit conv with planar store
                # (csrc/conv_exit.hip) instead of the 32-channel fp32 GEMM; a planar channel is an NHWC tensor of one channel
                dplanar = ops.conv_f16_nchw(dx, pw, (hp, wp))
                parts = [ops.pack_input_backward((dplanar if nc == 1 else dplanar[:, c].contiguous()).view(n, hp, wp, 1), 0, (h, w),
                                                 map_=sigma[:, c:c + 1].contiguous(), map_sqrt=True) for c in range(nc)]
            else:
                drec, _ = ops.conv_mfma(dx, rnet.head.packed_dgrad(), want_raw=True, out_channels=32)
                parts = [ops.pack_input_backward(drec, rnet.in_chn + c, (h, w), map_=sigma[:, c:c + 1].contiguous(), map_sqrt=True)
                         for c in range(nc)]
            d_sigma_total += torch.cat(parts, 1)
    else:
        for p in rnet.parameters():
            grads[p] = torch.zeros_like(p)
    # ---- SNet: sigma = exp(clamp(v))  (VIRNet.py:43) -> dv = dsigma * sigma inside the clamp range
    inside = (sigma > float(torch.tensor(LOG_MIN).exp())) & (sigma < float(torch.tensor(LOG_MAX).exp()))
    dv = (d_sigma_total * sigma * inside).contiguous()                           # few-channel map: host-side glue
    g16 = ops.pack_input(dv, h, w, zero_pad=True)
    acts, acts_t, mids = tape.snet["acts"], tape.snet["acts_t"], tape.snet["mids"]
    _conv_grads(grads, snet.conv_last, acts[-1], g16, reducer=reducer, xt=acts_t[-1])
    nxt_c = mids[-1].cout if mids else snet.conv1.cout
    emitting = _side_stream(g16.device) is None
    dpre, _, dpre_t = _conv3(g16, snet.conv_last.packed_dgrad(), dict(act=None, colsum=nxt_c) if emitting else None, mask=acts[-1], mask_slope=0.25, want_raw=True)
    for k in range(len(mids) - 1, -1, -1):                                       # post-activation stack, DnCNN.py:25-28
        _conv_grads(grads, mids[k], acts[k], dpre, reducer=reducer, xt=acts_t[k], yt=dpre_t)
        nxt_c = mids[k - 1].cout if k > 0 else snet.conv1.cout
        dpre, _, dpre_t = _conv3(dpre, mids[k].packed_dgrad(), dict(act=None, colsum=nxt_c) if emitting else None, mask=acts[k], mask_slope=0.25, want_raw=True)
    _conv_grads(grads, snet.conv1, tape.snet["rec"], dpre, reducer=reducer, yt=dpre_t)
    return grads


class DenoiseFunction(torch.autograd.Function):
    """mu, sigma = f(x; parameters): forward and backward both run on the C-ABI kernels."""

    @staticmethod
    def forward(ctx, x, net, *params):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("VIRAttResUNet: a gradient with respect to the input image is not implemented (the reference's training "
                               "never asks for one, train_denoising_syn.py:171-184); pass x.detach()")
        x = _prep(x, net.SNet.in_channels)          # raises on CPU / wrong dtype before anything touches a device
        with torch.no_grad(), torch.cuda.device(x.device):
            mu, sigma, tape = denoise_forward_train(net, x)
        ctx.net, ctx.tape, ctx.params = net, tape, params
        return mu, sigma

    @staticmethod
    def backward(ctx, dmu, dsigma):
        dev = (dmu if dmu is not None else dsigma).device
        reducer = getattr(ctx.net, "_grad_reducer", None)
        with torch.no_grad(), torch.cuda.device(dev):
            # The backward is linear in (dmu, dsigma) and its GEMMs split their operands into fp16 pairs, whose exact range is
            # 6e-5 .. 65504: the incoming gradients are scaled by a power of two (exact) so that their largest entry sits in [0.5, 1),
            # and the parameter gradients are scaled back at the end -- the result does not depend on how the caller scaled its loss
            # (a mean-reduced MSE hands over ~6e-8 per entry, a sum-reduced loss or a GradScaler 1e4 and more).  No host sync.
            amax = torch.zeros((), dtype=torch.float32, device=dev)
            for g in (dmu, dsigma):
                if g is not None:
                    amax = torch.maximum(amax, g.detach().abs().amax())
            if reducer is not None and reducer.world > 1:
                # every rank must use the SAME factor: the scaled gradients are summed across ranks inside the backward (reducer.push)
                # and unscaled afterwards -- with per-rank factors rank r would get (1/s_r) * mean_k(s_k g_k).  One 4-byte MAX
                # all-reduce on the device, ordered on the stream like every other collective: still no host sync.
                import torch.distributed as dist
                dist.all_reduce(amax, op=dist.ReduceOp.MAX, group=reducer.group)
            scale = torch.exp2(torch.clamp(-torch.floor(torch.log2(amax.clamp_min(1e-37))) - 1.0, -100.0, 100.0))
            scale = torch.where(amax > 0, scale, torch.ones_like(scale))
            dmu = None if dmu is None else dmu * scale
            dsigma = None if dsigma is None else dsigma * scale
            if reducer is not None:
                reducer.start()
            nbias = sum(p.numel() for p in ctx.params if p.dim() == 1)      # every bias gradient out of ONE zeroed buffer (45 fill launches less)
            with ops.zero_arena(nbias + 64, dev):
                grads = denoise_backward(ctx.net, ctx.tape, dmu, dsigma, reducer=reducer)
            side = _side_stream(dev)
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)      # gradients produced on the side stream are consumed after this
            if reducer is not None:
                grads = reducer.finish()                  # averaged over the ranks (copies of the flat buckets' slices)
            outs = [g for g in grads.values() if g is not None]
            if outs:
                torch._foreach_mul_(outs, 1.0 / scale)    # (a power of two: exact)
        ctx.tape = None
        return (None, None) + tuple(grads.get(p) if p.requires_grad else None for p in ctx.params)


def denoise_forward_autograd(net, x: Tensor) -> Tuple[Tensor, Tensor]:
    if net.SNet.noise_avg or net.RNet.extra_mode not in ("input", "null"):
        # SFT conditioning from the per-pixel variance map (extra_mode Down / Both) or a pooled variance: the per-layer autograd nodes
        # of the SISR step cover these (train_sisr.denoise_forward_nodes); the fused single-Function step below is the shipped
        # configurations' path (configs/denoising_*.json: extra_mode Input, noise_avg False)
        from . import train_sisr
        return train_sisr.denoise_forward_nodes(net, x)
    params = tuple(net.parameters())
    return DenoiseFunction.apply(x, net, *params)
