"""T emission (csrc/conv_f16.hip / conv_f16_wx4.hip, TE = 1): a training-step convolution writes, next to its NHWC tensor, the channel-major
image the weight-gradient GEMM contracts over and the tensor's channel sums.  Pinned here:
  * the NHWC result is bit for bit the non-emitting kernel's,
  * the emitted image is bit for bit what virnet_chsplit writes for the stored tensor (pads included), with and without the LeakyReLU
    of the consuming conv's staging, for split-fp16 and bf16 operands, all epilogue forms, ragged / narrow / multi-tile shapes,
  * the channel sums match a column sum of the stored tensor,
  * a weight gradient fed with emitted images equals the one that re-lays its operands itself (bitwise)."""
import ctypes as C

import pytest
import torch

from virnet_amd import _native as nat, ops
from test_ops_gpu import make_conv, nhwc, rnd

pytestmark = pytest.mark.gpu


def chsplit_ref(t, bf16, slope=None):
    n, h, w, c = t.shape
    lib = nat.load()
    out = torch.zeros(lib.virnet_chsplit_bytes(n, h, w, c), dtype=torch.uint8, device="cuda")
    nat.check(lib.virnet_chsplit(nat.ptr(t), n, h, w, c, int(slope is not None), 0.0 if slope is None else slope, None, None, int(bf16),
                                 nat.ptr(out), None, None, 0, nat.stream_handle()), "chsplit")
    return out


CASES = [  # form, n, h, w, c, epi (res, mask), act slope of T, want_act
    ("f16x3", 2, 19, 45, 96, (False, False), None, False),
    ("f16x3", 1, 8, 32, 64, (True, False), 0.2, False),          # conv2-type: raw + residual, T = lrelu(out)
    ("f16x3", 2, 17, 70, 160, (False, True), None, False),       # dgrad conv2-type: mask; 160 = 3 + 2 slabs
    ("f16x3", 1, 33, 31, 32, (True, True), None, False),         # dgrad conv1-type: mask + residual; one slab
    ("f16x3", 2, 12, 40, 96, (False, False), None, True),        # conv1-type: activated store
    ("bf16", 2, 19, 45, 96, (True, False), 0.2, False),
    ("bf16", 1, 24, 64, 192, (False, True), None, False),
    ("wx4", 16, 64, 64, 96, (False, False), None, True),         # 128 16-row tiles: the Winograd form emits
    ("wx4", 12, 56, 80, 96, (True, False), 0.2, False),          # ragged tiles
    ("wx4", 16, 32, 64, 192, (False, True), None, False),
    ("wx4", 64, 32, 32, 64, (True, True), None, False),          # two slabs, narrow T rows
]


CASES = [c + (16,) for c in CASES] + [c + (8,) for c in CASES if c[0] == "wx4"]     # the Winograd form emits from 16-row and from 8-row tiles


@pytest.mark.parametrize("form,n,h,w,c,epi,tslope,want_act,rows", CASES)
def test_emitted_image_is_the_chsplit_of_the_stored_tensor(form, n, h, w, c, epi, tslope, want_act, rows, monkeypatch):
    monkeypatch.setenv("VIRNET_CONV_FORM", form)
    monkeypatch.setenv("VIRNET_WX4_ROWS", str(rows))              # (pins the reference run's tile form; the emitting run follows it)
    monkeypatch.setenv("VIRNET_WX4_EMIT_ROWS", str(rows))
    cp = make_conv(c, c, seed=31).cuda()
    x = nhwc(rnd(n, c, h, w, seed=32))
    res = nhwc(rnd(n, c, h, w, seed=33)) if epi[0] else None
    mask = nhwc(rnd(n, c, h, w, seed=34)) if epi[1] else None
    kw = dict(res=res, mask=mask, want_raw=not want_act, want_act=want_act, slope=0.2, in_slope=0.2 if not epi[1] else None)
    raw0, act0 = ops.conv_mfma(x, cp.packed(), **kw)
    timer = ops.LaunchTimer()
    ops.set_launch_timer(timer)
    try:
        raw, act, timg = ops.conv_mfma(x, cp.packed(), emit=dict(act=tslope, colsum=c), **kw)
    finally:
        ops.set_launch_timer(None)
    assert [k[0] for k in timer.summary()] == [form]                                # the emitting launch ran in the form under test
    assert timg is not None and timg.bf16 == (form == "bf16") and (timg.n, timg.h, timg.w, timg.c) == (n, h, w, c)
    y0, y = (act0, act) if want_act else (raw0, raw)
    assert torch.equal(y0, y)                                                       # same arithmetic, different store order
    ref = chsplit_ref(y, form == "bf16", tslope)
    assert ref.numel() == timg.buf.numel()
    diff = int((ref != timg.buf).sum())
    assert diff == 0, f"{diff} of {ref.numel()} bytes differ"
    colsum = y.double().sum((0, 1, 2))
    assert float((timg.bias_sums().double() - colsum).abs().max()) <= 2e-5 * max(1.0, float(colsum.abs().max()))
    # pooled buffers: a released image comes back with clean pads even after a different tensor went through it
    ops.t_release(timg)
    _, _, t2 = ops.conv_mfma(x * 0.5, cp.packed(), emit=dict(act=tslope, colsum=None), **kw)
    y2 = ops.conv_mfma(x * 0.5, cp.packed(), **kw)[1 if want_act else 0]
    assert t2.bias_sums() is None and torch.equal(chsplit_ref(y2, form == "bf16", tslope), t2.buf)


def test_small_or_unsupported_launches_fall_back(monkeypatch):
    monkeypatch.setenv("VIRNET_CONV_FORM", "wx4")
    cp = make_conv(96, 96, seed=35).cuda()
    x = nhwc(rnd(1, 96, 32, 32, seed=36))
    # one small image: below the Winograd form's launch size -> the direct kernel emits
    raw, _, timg = ops.conv_mfma(x, cp.packed(), want_raw=True, emit=dict(act=None, colsum=None))
    assert timg is not None and torch.equal(chsplit_ref(raw, False), timg.buf)
    # two stored tensors: no emission, the caller re-lays
    raw, act, timg = ops.conv_mfma(x, cp.packed(), want_raw=True, want_act=True, emit=dict(act=None, colsum=None))
    assert timg is None and raw is not None and act is not None
    monkeypatch.setenv("VIRNET_T_EMIT", "0")
    assert ops.conv_mfma(x, cp.packed(), want_raw=True, emit=dict(act=None, colsum=None))[2] is None
    monkeypatch.delenv("VIRNET_T_EMIT")
    monkeypatch.setenv("VIRNET_CONV_FORM", "wino")                # fp32 forms: no T
    assert ops.conv_mfma(x, cp.packed(), want_raw=True, emit=dict(act=None, colsum=None))[2] is None


@pytest.mark.parametrize("form", ["wx4", "bf16"])
def test_weight_gradient_from_emitted_images_is_bitwise_the_relaid_one(form, monkeypatch):
    monkeypatch.setenv("VIRNET_CONV_FORM", form)
    n, h, w, c = 8, 64, 64, 96
    cp1, cp2 = make_conv(c, c, seed=41).cuda(), make_conv(c, c, seed=42).cuda()
    x = nhwc(rnd(n, c, h, w, seed=43))
    # forward pair: conv1 emits T(f1a); a stand-in gradient pair: dgrad conv emits T(dy) + its channel sums
    _, f1a, t_f1a = ops.conv_mfma(x, cp1.packed(), in_slope=0.2, want_raw=False, want_act=True, slope=0.2, emit=dict(act=None, colsum=None))
    g = nhwc(rnd(n, c, h, w, seed=44)) * 0.1
    dy, _, t_dy = ops.conv_mfma(g, cp2.packed_dgrad(), mask=f1a, mask_slope=0.2, want_raw=True, emit=dict(act=None, colsum=c))
    assert t_f1a is not None and t_dy is not None
    dw_a, db_a = ops.conv_wgrad(f1a, dy, (c, c, 3, 3), bias_channels=c, xt=t_f1a, yt=t_dy)
    dw_b, db_b = ops.conv_wgrad(f1a, dy, (c, c, 3, 3), bias_channels=c)
    assert torch.equal(dw_a, dw_b)
    assert float((db_a - db_b).abs().max()) <= 2e-5 * float(db_b.abs().max())
