"""Forward orchestration: which kernel runs when, with which fused prologue/epilogue.

Every MFMA conv stores exactly ONE tensor.  The reference's pre-activation convs (``conv(lrelu(x*mul+add))``,
networks/AttResUNet.py:55,58) are served two ways, both inside the conv kernel:
  * a tensor that is also needed raw (the residual stream x: AttResUNet.py:59, the bridges, the stride-2 / transposed convs) is
    stored raw, and the consuming conv applies LeakyReLU (and the SFT scale/shift) while it stages the pixels into LDS;
  * a tensor only ever consumed activated (conv1's output, the DnCNN / KNet post-activation features) is stored activated by
    its producer's epilogue.
Either way the zero padding applies to the ACTIVATED tensor ("pad after activation") and no elementwise kernel runs between convs.

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


def _conv_planar(x: Tensor, conv, crop_hw, **kw) -> Tensor:
    """3x3 conv with planar (NCHW) store: the bandwidth-bound kernel for <= 4 output channels, the MFMA kernel otherwise."""
    if ops._f16_family() and conv.cout <= 32:
        return ops.conv_f16_nchw(x, conv.packed(), crop_hw, **kw)          # split-fp16 kernel, one slab, planar store
    if conv.cout <= 4:
        return ops.conv3x3_thin(x, conv.packed_thin(), crop_hw, **kw)
    return ops.conv_mfma_nchw(x, conv.packed(), crop_hw, **kw)


# ----------------------------------------------------------------------------------------------------------------
# SNet (reference networks/DnCNN.py:37-44)
# ----------------------------------------------------------------------------------------------------------------
def snet_forward(snet, x: Tensor, mode: str = "raw") -> Tensor:
    """mode 'raw': DnCNN.forward output; mode 'sigma': exp(clamp(.)) of it (VIRNet.py:43), fused where possible."""
    x = _prep(x, snet.in_channels)
    n, _, h, w = x.shape
    cur = ops.conv_entry(x, snet.conv1.packed(), h, w, want_act=True, slope=0.25)      # DnCNN.py:38: entry packing folded into the conv
    for key in sorted(snet.mid_layer.keys(), key=int):
        _, cur = ops.conv_mfma(cur, snet.mid_layer[key].packed(), want_raw=False, want_act=True, slope=0.25)
    last = snet.conv_last
    if snet.noise_avg:
        raw = _conv_planar(cur, last, (h, w))
        fin = ops.GAP_EXPCLAMP if mode == "sigma" else ops.GAP_MEAN
        return ops.gap_nchw(raw, fin, (LOG_MIN, LOG_MAX)).view(n, -1, 1, 1)
    if mode == "sigma":
        return _conv_planar(cur, last, (h, w), op=nat.NCHW_EXPCLAMP, clamp=(LOG_MIN, LOG_MAX))
    return _conv_planar(cur, last, (h, w))


# ----------------------------------------------------------------------------------------------------------------
# KNet (reference networks/KNet.py:52-59)
# ----------------------------------------------------------------------------------------------------------------
def knet_forward(knet, x: Tensor) -> Tensor:
    x = _prep(x, knet.in_nc)
    n, _, h, w = x.shape
    cur = ops.conv_head_s4(x, knet.head.weight)                                  # KNet.py:45,53 -> NHWC raw
    persistent = (ops._f16_family() and cur.shape[3] == 64 and max(cur.shape[1:3]) <= ops.KNET_BODY_MAX and len(knet.body) > 0
                  and ops._env("VIRNET_KNET_PERSISTENT", "1") != "0")
    if persistent:
        # the whole body in ONE launch, the map resident on one CU per image (csrc/knet_body.hip; KNet.py:28-39,46-48,54)
        cur = ops.knet_body(cur, [(rb.body["0"].packed(), rb.body["2"].packed(), rb.body["3"].body["0"].weight, rb.body["3"].body["0"].bias,
                                   rb.body["3"].body["2"].weight, rb.body["3"].body["2"].bias) for rb in knet.body])
    for rb in (() if persistent else knet.body):
        _, a = ops.conv_mfma(cur, rb.body["0"].packed(), want_raw=False, want_act=True, slope=0.2)   # KNet.py:32-33
        hcv, _ = ops.conv_mfma(a, rb.body["2"].packed(), want_raw=True)                               # KNet.py:34
        ca = rb.body["3"].body
        cur = ops.ca_scale_add(hcv, ca["0"].weight, ca["0"].bias, ca["2"].weight, ca["2"].bias, cur)  # KNet.py:15-26,38 (one launch)
    oh, ow = cur.shape[1:3]
    raw = _conv_planar(cur, knet.tail["0"], (oh, ow))                                                 # KNet.py:49
    return ops.gap_nchw(raw, ops.GAP_KINFO, (K_LOG_MIN, LOG_MAX)).view(n, -1, 1, 1)                  # KNet.py:50,56-59


# ----------------------------------------------------------------------------------------------------------------
# RNet (reference networks/AttResUNet.py:141-175)
# ----------------------------------------------------------------------------------------------------------------
class _Cond:
    """Conditioning seen by the SFT layers of the down path: per-image vector or per-pixel records."""

    def __init__(self, vec: Optional[Tensor], rec: Optional[Tensor], chan0: int, nchan: int):
        self.vec, self.rec, self.chan0, self.nchan = vec, rec, chan0, nchan
        self.pre = {}            # id(AttLayer) -> (mul, add) computed ahead for the whole down path (ops.sft_vec_multi)

    def sft(self, att):
        got = self.pre.get(id(att))
        return got if got is not None else ops.sft_vec(self.vec, att)


def _res_block(x_raw: Tensor, blk, cond: Optional[_Cond], level: int) -> Tensor:
    """AttResBlock.forward (AttResUNet.py:48-60): x + conv2(lrelu(sft2(conv1(lrelu(sft1(x)))))), two launches."""
    sft = cond is not None and blk.extra_chn > 0
    c1, c2 = blk.conv1.packed(), blk.conv2.packed()
    if not sft:
        _, f1a = ops.conv_mfma(x_raw, c1, in_slope=0.2, want_raw=False, want_act=True, slope=0.2)
    elif cond.vec is not None:   # spatially constant conditioning: SFT collapses to per-(image, channel) scale/shift
        mul1, add1 = cond.sft(blk.sft1)
        mul2, add2 = cond.sft(blk.sft2)
        _, f1a = ops.conv_mfma(x_raw, c1, in_slope=0.2, in_mul=mul1, in_add=add1, mul=mul2, add=add2, want_raw=False,
                               want_act=True, slope=0.2)
    else:                        # per-pixel conditioning: materialise the two modulated tensors (rare configuration)
        a1 = ops.sft_apply(x_raw, cond.rec, cond.chan0, cond.nchan, 1 << level, blk.sft1)
        f1, _ = ops.conv_mfma(a1, c1, want_raw=True)
        f1a = ops.sft_apply(f1, cond.rec, cond.chan0, cond.nchan, 1 << level, blk.sft2)
    out, _ = ops.conv_mfma(f1a, c2, res=x_raw, want_raw=True)
    return out


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
    cond = None
    rec = None
    if feed_down:
        if extra_map is None:
            cond = _Cond(extra_vec, None, 0, ev)
        else:  # per-pixel conditioning: keep full-resolution padded records of the extra channels (AttResUNet.py:158,168)
            rec = ops.pack_input(x_in, Hp, Wp, sf=sf, vec=extra_vec if feed_head else None,
                                 map_=extra_map if feed_head else None, map_sf=map_sf, map_sqrt=map_sqrt)
            crec = rec if feed_head else ops.pack_input(x_in, Hp, Wp, sf=sf, vec=extra_vec, map_=extra_map,
                                                        map_sf=map_sf, map_sqrt=map_sqrt)
            cond = _Cond(None, crec, rnet.in_chn, ev + em)
    if rec is not None:
        x, _ = ops.conv_mfma(rec, rnet.head.packed(), want_raw=True)                   # AttResUNet.py:153-155
    else:                                                                              # ... with the entry packing folded into the conv
        x = ops.conv_entry(x_in, rnet.head.packed(), Hp, Wp, sf=sf, vec=extra_vec if feed_head else None,
                           map_=extra_map if feed_head else None, map_sf=map_sf, map_sqrt=map_sqrt)
    if cond is not None and cond.vec is not None and ops._env("VIRNET_SFT_MULTI", "1") != "0":
        # every SFT layer of the down path sees the same vector: all their (mul, add) pairs in one launch, ahead of the first block
        atts = [a for lvl in rnet.down_path for blk in lvl.body if blk.extra_chn > 0 for a in (blk.sft1, blk.sft2)]
        if atts:
            cond.pre = {id(a): ma for a, ma in zip(atts, ops.sft_vec_multi(cond.vec, atts))}
    bridges: List[Tensor] = []
    for ii, lvl in enumerate(rnet.down_path):
        for blk in lvl.body:
            x = _res_block(x, blk, cond, ii)
        if ii + 1 < len(rnet.down_path):
            bridges.append(x)
            x, _ = ops.conv_mfma(x, lvl.downsampler.packed(), stride=2, want_raw=True)   # AttResUNet.py:67,74
    for jj, up in enumerate(rnet.up_path):
        x, _ = ops.conv_mfma(x, up.upsampler.packed(), res=bridges[-jj - 1], want_raw=True)   # AttResUNet.py:84-87
        for blk in up.body:
            x = _res_block(x, blk, None, 0)
    return _conv_planar(x, rnet.tail, (H, W), op=nat.NCHW_ADD, res=x_in, res_sf=sf)                  # AttResUNet.py:173


# ----------------------------------------------------------------------------------------------------------------
# boundary forwards (reference networks/VIRNet.py:42-46 and :80-97)
# ----------------------------------------------------------------------------------------------------------------
FP32_FORM = "wino"      # the form of the guard's re-run: Winograd / direct fp32 MFMA kernels, no range limit


class RangeOverflowRepaired(RuntimeWarning):
    """A forward of the deferred guard mode left fp16's range: its outputs were NaN on the device until the repair that this warning
    announces (the same input repeated with the fp32 kernels, written into the same output tensors)."""


_GUARD_WARNING = ("VIRNet HIP path: an activation left fp16's range in a split-fp16 convolution (|x| >= 65504, or >= ~6.5e3 in the "
                  "Winograd form); the forward was repeated with the fp32 kernels")
MAX_PENDING = 2         # deferred mode: forwards whose flag copy may still be in flight before the host waits for the oldest


def guard_check_mode() -> str:
    """``VIRNET_GUARD_CHECK``: how the eager forward learns that a split-fp16 kernel raised the range flag.

    * ``sync`` (default; the reference's semantics: what ``forward`` returns is final): one device -> host read at the end of the
      forward -- it waits for the forward, which a caller that consumes the outputs right away (every reference script) pays anyway.
    * ``deferred`` (pipelined callers: serving loops; bench.py times it beside the default): NO host wait on the launch path.  The forward ends with
      ``virnet_poison_on_flag`` on every output (an out-of-range forward is NaN on the device before the host has looked) and an
      asynchronous copy of the flag into pinned memory + an event; the NEXT guarded forward of the thread (or ``guard_poll()``) looks at
      the copies that have landed, and for an overflowed forward warns and repeats THAT input with the fp32 kernels into the SAME
      output tensors -- stream-ordered behind whatever was enqueued meanwhile: loud (NaN) in between, correct afterwards.
      At most MAX_PENDING forwards stay unchecked; ``guard_poll()`` settles all of them (call it before trusting outputs on the host).
      The repair re-runs the forward on the INPUT TENSOR OBJECT it was given: a caller that overwrites its input buffer in place between
      forwards must poll first (or use sync mode) -- the pending entry holds a reference, not a copy."""
    mode = ops._env("VIRNET_GUARD_CHECK", "sync")
    if mode not in ("sync", "deferred"):
        raise ValueError(f"VIRNET_GUARD_CHECK={mode!r}: expected sync or deferred")
    return mode


def _pending_list() -> list:
    lst = getattr(nat.tls, "guard_pending", None)
    if lst is None:
        lst = nat.tls.guard_pending = []
    return lst


def _settle(block: bool, keep: int = 0) -> None:
    """Look at the deferred forwards of this thread, oldest first: those whose flag copy has landed (all but `keep` of them when
    `block`) are checked and, if they overflowed, repaired in place -- on the STREAM the forward ran on (the repair's writes are then
    ordered behind that forward's own kernels and in front of whatever the caller enqueues on that stream next; a consumer on another
    stream has to order itself against that stream as it had to for the forward itself).  The repair comes FIRST and the warning after
    it: under ``-W error`` (or any filter that raises) the exception still escapes from this call, but the outputs it is about are
    already correct."""
    import warnings
    pend = _pending_list()
    while pend:
        pinned, ev, run, outs, dev, stream = pend[0]
        if not ev.query():
            if not block or len(pend) <= keep:
                return
            ev.synchronize()
        pend.pop(0)
        overflowed = bool(int(pinned[0]))
        _pinned_pool().append(pinned)
        if overflowed:
            _GUARD_STATS["reruns"] += 1
            with torch.no_grad(), torch.cuda.device(dev), torch.cuda.stream(stream), ops.forward_scope(form=FP32_FORM):
                res = run()
                for o, r in zip(outs, res if isinstance(res, tuple) else (res,)):
                    o.copy_(r)
            warnings.warn(_GUARD_WARNING + " (deferred check: its outputs were NaN until now)", RangeOverflowRepaired, stacklevel=4)


def _pinned_pool() -> list:
    """Pinned one-int buffers of the calling thread's deferred checks (thread-local like the pending list: a process-global pool was a
    check-then-pop race between two host threads)."""
    pool = getattr(nat.tls, "guard_pinned", None)
    if pool is None:
        pool = nat.tls.guard_pinned = []
    return pool


def guard_poll(block: bool = True) -> None:
    """Deferred guard mode: check (and repair) the calling thread's forwards that have not been looked at yet; with ``block`` the
    host waits for their flag copies first.  A no-op in sync mode."""
    _settle(block)


def _range_guarded(run, x: Tensor):
    """Run a forward in the default (split-fp16) forms; when a kernel reports an operand outside fp16's range (virnet_set_range_flag)
    run it again with the fp32 kernels and return that result.  ``guard_check_mode()`` says when the host looks at the flag: at the end
    of the forward (sync: one int read, i.e. a wait for the forward) or not on the launch path at all (deferred).  Skipped while a
    hipGraph is being captured -- graph.GraphedForward does its own check around the replay -- for the fp32 forms and with
    VIRNET_RANGE_GUARD=0.  The flag is per (device, host thread) and the re-run's form override lives in the thread's forward_scope:
    nothing process-global is touched."""
    import warnings
    guarded = ops._f16_family() and ops.range_guard_enabled() and not torch.cuda.is_current_stream_capturing()
    deferred = guarded and guard_check_mode() == "deferred"
    if guarded:
        _settle(block=False)
        flag = ops.range_flag(x.device)
        flag.zero_()                         # (the flag is sticky: whatever raised it before this forward is not this forward's business)
    with ops.forward_scope():                # (environment knobs and the stream handle: read once per forward, not per launch)
        out = run()
    if deferred:
        outs = out if isinstance(out, tuple) else (out,)
        for o in outs:
            ops.poison_on_flag(flag, o)
        pool = _pinned_pool()
        if pool:
            pinned = pool.pop()
        else:
            with nat.capture_lock:        # (a pinned allocation beside another thread's hipGraph capture invalidates that capture)
                pinned = torch.zeros(1, dtype=torch.int32).pin_memory()
        pinned.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _pending_list().append((pinned, ev, run, outs, x.device, torch.cuda.current_stream()))
        _settle(block=True, keep=MAX_PENDING)
    elif guarded and ops.range_overflowed(x.device):
        warnings.warn(_GUARD_WARNING, RuntimeWarning, stacklevel=3)
        _GUARD_STATS["reruns"] += 1
        with ops.forward_scope(form=FP32_FORM):
            out = run()
    if guarded:
        _GUARD_STATS["forwards"] += 1
    return out


_GUARD_STATS = {"forwards": 0, "reruns": 0}


def guard_stats(reset: bool = False) -> dict:
    """How many guarded forwards ran in this process and how many of them had to be repeated in fp32."""
    out = dict(_GUARD_STATS)
    if reset:
        _GUARD_STATS["forwards"] = _GUARD_STATS["reruns"] = 0
    return out


def denoise_forward(net, x: Tensor) -> Tuple[Tensor, Tensor]:
    x = _prep(x, net.SNet.in_channels)
    with torch.cuda.device(x.device):       # launches go to x's device even when it is not the process's current one
        return _range_guarded(lambda: _denoise_forward(net, x), x)


def _denoise_forward(net, x: Tensor) -> Tuple[Tensor, Tensor]:
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
    with torch.cuda.device(x.device):
        return _range_guarded(lambda: _sisr_forward(net, x, sf), x)


def _sisr_forward(net, x: Tensor, sf) -> Tuple[Tensor, Tensor, Tensor]:
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
