"""GPU parity of the f16-pipe weight gradient (csrc/wgrad_f16.hip; SURVEY.md 8-f1: backward of networks/AttResUNet.py:43,46 and
DnCNN.py:22-29 under train_denoising_syn.py:176-179) against torch autograd on the CPU.

  * virnet_chsplit: the channel-major fp16 hi/lo re-layout is reproduced element for element in numpy (bit-exact),
  * virnet_conv_wgrad_f16 (+ the bias gradient fused into the dY pass): <= 2e-5 of the gradient's scale vs autograd (the split
    operands carry 22 bits; fp32 accumulation in a different order), on shapes that exercise every strip / ring / group edge,
  * bitwise reproducibility (no atomics), the bf16 single-product variant against autograd on bf16-rounded operands,
  * ABI rejections."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from virnet_amd import _native as nat, ops
from test_backward_gpu import autograd_conv, relerr
from test_ops_gpu import nhwc, rnd

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _nseg(w):
    return 6 if w <= 32 else 8 * ((w + 63) // 64) + 2


@pytest.mark.parametrize("bf16,w", [(0, 37), (1, 37), (0, 32), (0, 9)])
def test_chsplit_layout_is_exact(bf16, w):
    n, h, c = 2, 6, 40                                          # 40 channels: the second 32-block is a quarter full
    x = rnd(n, c, h, w, seed=300)
    mul, add = rnd(n, c, seed=301, lo=0.3, hi=1.2), rnd(n, c, seed=302)
    lib = nat.load()
    nbytes = lib.virnet_chsplit_bytes(n, h, w, c)
    nseg, cb = _nseg(w), (c + 31) // 32
    assert nbytes == n * (h + 2) * cb * 2 * nseg * 512
    out = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device="cuda")
    xd, muld, addd = nhwc(x), mul.cuda(), add.cuda()            # (named: the launch must not outlive its operands)
    nat.check(lib.virnet_chsplit(nat.ptr(xd), n, h, w, c, 1, 0.2, nat.ptr(muld), nat.ptr(addd), bf16, nat.ptr(out), None, None, 0,
                                 nat.stream_handle()), "chsplit")
    t = out.cpu().numpy().view(np.uint16).reshape(n, h + 2, cb, 2, nseg, 32, 8)
    # the staging transform is one fused multiply-add per element (exact product, one rounding), then lrelu in fp32
    a = F.leaky_relu((x.double() * mul.double().view(n, c, 1, 1) + add.double().view(n, c, 1, 1)).float(), 0.2)
    if bf16:
        hi = a.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
        planes = [hi]
    else:
        hi16 = a.to(torch.float16)
        lo16 = (a - hi16.float()).to(torch.float16)
        planes = [hi16.view(torch.int16).numpy().view(np.uint16), lo16.view(torch.int16).numpy().view(np.uint16)]
    for p, ref in enumerate(planes):
        exp = np.zeros((n, h + 2, cb * 32, nseg * 8), np.uint16)
        exp[:, 1:h + 1, :c, 8:8 + w] = ref.transpose(0, 2, 1, 3)           # [n][row][ch][x], pixel x at index x + 8
        got = t[:, :, :, p].transpose(0, 1, 2, 4, 3, 5).reshape(n, h + 2, cb * 32, nseg * 8)   # [n][row][cb][seg][32][8] -> [n][row][ch][x]
        # -0.0 halves can appear where lrelu gives -0: compare as values there
        same = (got == exp) | ((got & 0x7FFF) == 0) & ((exp & 0x7FFF) == 0)
        assert same.all(), f"plane {p}: {int((~same).sum())} elements differ"


SHAPES = [  # cin, cout, h, w, n
    (96, 96, 5, 64, 1),        # minimum height the row ring takes; exactly one 64-pixel strip
    (96, 96, 7, 65, 2),        # second strip one pixel wide
    (32, 288, 6, 130, 1),      # nine output blocks = three groups; three strips
    (288, 64, 9, 31, 2),       # two output blocks (8-wave workgroups), nine input blocks
    (64, 16, 11, 50, 3),       # one output block (4-wave workgroups)
    (160, 160, 6, 20, 1),      # five blocks: the second group is one short
]


@pytest.mark.parametrize("cin,cout,h,w,n", SHAPES)
def test_wgrad_f16_and_fused_bias_vs_autograd(cin, cout, h, w, n):
    x, dy = rnd(n, cin, h, w, seed=310), rnd(n, cout, h, w, seed=311)
    wt = rnd(cout, cin, 3, 3, seed=312) * 0.1
    _, dw_ref, db_ref = autograd_conv(x, wt, torch.zeros(cout), dy, in_slope=0.2)
    dw, db = ops.conv_wgrad(nhwc(x), nhwc(dy), (cout, cin, 3, 3), in_slope=0.2, bias_channels=cout)
    assert relerr(dw.cpu(), dw_ref) <= TOL, relerr(dw.cpu(), dw_ref)
    assert relerr(db.cpu(), db_ref) <= TOL, relerr(db.cpu(), db_ref)


def test_wgrad_f16_is_bitwise_reproducible_and_close_to_the_fp32_kernel(monkeypatch):
    cin = cout = 96
    x, dy = rnd(4, cin, 24, 70, seed=320), rnd(4, cout, 24, 70, seed=321)
    xd, dyd = nhwc(x), nhwc(dy)
    a = ops.conv_wgrad(xd, dyd, (cout, cin, 3, 3), in_slope=0.2)
    b = ops.conv_wgrad(xd, dyd, (cout, cin, 3, 3), in_slope=0.2)
    assert torch.equal(a, b)                                     # fixed-order reduction, no atomics
    monkeypatch.setenv("VIRNET_WGRAD_FORM", "f32")
    c = ops.conv_wgrad(xd, dyd, (cout, cin, 3, 3), in_slope=0.2)
    assert relerr(a.cpu(), c.cpu()) <= TOL


def test_wgrad_f16_wide_dynamic_range():
    """Gradients spanning five decades inside one contraction (the hi/lo split must carry the small ones next to the large)."""
    g = torch.Generator().manual_seed(330)
    cin, cout, h, w = 64, 96, 12, 48
    x = rnd(2, cin, h, w, seed=331)
    dy = rnd(2, cout, h, w, seed=332) * (10.0 ** torch.randint(-4, 1, (2, cout, h, w), generator=g).float())
    wt = rnd(cout, cin, 3, 3, seed=333) * 0.1
    _, dw_ref, _ = autograd_conv(x.double(), wt.double(), torch.zeros(cout).double(), dy.double())
    dw = ops.conv_wgrad(nhwc(x), nhwc(dy), (cout, cin, 3, 3))
    assert relerr(dw.cpu().double(), dw_ref) <= TOL


def test_wgrad_bf16_variant_matches_bf16_rounded_autograd(monkeypatch):
    monkeypatch.setenv("VIRNET_CONV_FORM", "bf16")
    cin, cout, h, w, n = 96, 192, 10, 72, 2
    x, dy = rnd(n, cin, h, w, seed=340), rnd(n, cout, h, w, seed=341)
    a = F.leaky_relu(x, 0.2).to(torch.bfloat16).float()          # what virnet_chsplit(bf16) stores
    dyr = dy.to(torch.bfloat16).float()
    wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(a, wt, None, padding=1).backward(dyr)               # products of bf16 values are exact in fp32: only the sum order differs
    dw, db = ops.conv_wgrad(nhwc(x), nhwc(dy), (cout, cin, 3, 3), in_slope=0.2, bias_channels=cout)
    assert relerr(dw.cpu(), wt.grad) <= TOL
    assert relerr(db.cpu(), dy.sum((0, 2, 3))) <= TOL            # the bias gradient stays fp32


def test_wgrad_f16_abi_rejects_bad_arguments():
    lib = nat.load()
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    f = torch.zeros(1 << 16, dtype=torch.float32, device="cuda")
    st = nat.stream_handle()
    assert lib.virnet_chsplit(None, 1, 8, 8, 32, 0, 0.0, None, None, 0, nat.ptr(buf), None, None, 0, st) != 0
    assert lib.virnet_chsplit(nat.ptr(f), 1, 8, 8, 30, 0, 0.0, None, None, 0, nat.ptr(buf), None, None, 0, st) != 0      # c % 4
    assert lib.virnet_chsplit(nat.ptr(f), 1, 8, 8, 32, 0, 0.0, nat.ptr(f), None, 0, nat.ptr(buf), None, None, 0, st) != 0  # mul without add
    assert lib.virnet_chsplit(nat.ptr(f), 1, 8, 8, 32, 0, 0.0, None, None, 0, nat.ptr(buf), None, nat.ptr(f), 8, st) != 0  # db without scratch
    assert lib.virnet_conv_wgrad_f16(nat.ptr(buf), nat.ptr(buf), nat.ptr(f), nat.ptr(f), 1, 4, 8, 32, 32, 32, 32, 0, st) != 0   # h < 5
    assert lib.virnet_conv_wgrad_f16(nat.ptr(buf), nat.ptr(buf), nat.ptr(f), nat.ptr(f), 1, 8, 8, 32, 32, 40, 32, 0, st) != 0   # cin > cx
    assert lib.virnet_conv_wgrad_f16(nat.ptr(buf), nat.ptr(buf), None, nat.ptr(f), 1, 8, 8, 32, 32, 32, 32, 0, st) != 0
    assert b"virnet_conv_wgrad_f16" in nat.last_error() if hasattr(nat, "last_error") else True


@pytest.mark.parametrize("n,h,w,c", [(32, 128, 128, 96), (32, 64, 64, 192), (32, 32, 32, 288), (8, 256, 256, 96)])
def test_wgrad_f16_full_size_against_the_fp32_kernel(n, h, w, c, monkeypatch):
    """BASELINE configs[4]'s layer shapes (and a 256^2 one): the f16-pipe gradient against the round-1 fp32 MFMA kernel on the same
    device tensors -- an independent implementation (different tiling, split and reduction), so index arithmetic at full size is covered."""
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand(n, h, w, c, device="cuda", generator=g) - 0.5
    dy = (torch.rand(n, h, w, c, device="cuda", generator=g) - 0.5) * 0.1
    dw, db = ops.conv_wgrad(x, dy, (c, c, 3, 3), in_slope=0.2, bias_channels=c)
    monkeypatch.setenv("VIRNET_WGRAD_FORM", "f32")
    dw32, db32 = ops.conv_wgrad(x, dy, (c, c, 3, 3), in_slope=0.2, bias_channels=c)
    scale = float(dw32.abs().max())
    assert float((dw - dw32).abs().max()) <= 3e-5 * scale
    assert float((db - db32).abs().max()) <= 3e-5 * float(db32.abs().max())


# ---- the stride-2 layers on the same kernel (S = 2): DownBlock.downsampler AttResUNet.py:67, UpBlock.upsampler :80 -----------------------
@pytest.mark.parametrize("bf16", [0, 1])
def test_chsplit_s2_layout_is_exact(bf16):
    """Column-phase T: block par*cb + k of row r holds the pixels x = 2*ox + par of source block k, pixel ox at index ox + 8."""
    n, h, w, c = 2, 6, 74, 64
    x = rnd(n, c, h, w, seed=350)
    lib = nat.load()
    nbytes = lib.virnet_chsplit_s2_bytes(n, h, w, c)
    nseg, cb = _nseg(w // 2), c // 32
    assert nbytes == n * (h + 2) * 2 * cb * 2 * nseg * 512
    out = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device="cuda")
    xd = nhwc(x)
    nat.check(lib.virnet_chsplit_s2(nat.ptr(xd), n, h, w, c, 1, 0.2, None, None, bf16, nat.ptr(out), None, None, 0, nat.stream_handle()), "chsplit_s2")
    t = out.cpu().numpy().view(np.uint16).reshape(n, h + 2, 2 * cb, 2, nseg, 32, 8)
    a = F.leaky_relu(x, 0.2)
    if bf16:
        planes = [a.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)]
    else:
        hi16 = a.to(torch.float16)
        lo16 = (a - hi16.float()).to(torch.float16)
        planes = [hi16.view(torch.int16).numpy().view(np.uint16), lo16.view(torch.int16).numpy().view(np.uint16)]
    for p, ref in enumerate(planes):
        exp = np.zeros((n, h + 2, 2 * c, nseg * 8), np.uint16)
        for par in range(2):
            exp[:, 1:h + 1, par * c:(par + 1) * c, 8:8 + w // 2] = ref[:, :, :, par::2].transpose(0, 2, 1, 3)
        got = t[:, :, :, p].transpose(0, 1, 2, 4, 3, 5).reshape(n, h + 2, 2 * c, nseg * 8)
        same = (got == exp) | ((got & 0x7FFF) == 0) & ((exp & 0x7FFF) == 0)
        assert same.all(), f"plane {p}: {int((~same).sum())} elements differ"


S2_SHAPES = [  # cin, cout, h, w (input), n
    (96, 192, 10, 128, 1),     # minimum ring height; exactly one 64-pixel strip of outputs
    (96, 192, 14, 130, 2),     # second strip one output pixel wide
    (192, 288, 16, 64, 2),     # 32-column outputs: 32-pixel steps (six-wave workgroups), nine output blocks = three groups
    (32, 64, 12, 20, 3),       # two output blocks, narrow
    (160, 96, 22, 300, 1),     # three strips, five input blocks per phase
]


@pytest.mark.parametrize("cin,cout,h,w,n", S2_SHAPES)
def test_wgrad_f16_stride2_conv_vs_autograd(cin, cout, h, w, n):
    x, dy = rnd(n, cin, h, w, seed=360), rnd(n, cout, h // 2, w // 2, seed=361)
    wt = rnd(cout, cin, 3, 3, seed=362) * 0.1
    _, dw_ref, db_ref = autograd_conv(x, wt, torch.zeros(cout), dy, stride=2, in_slope=0.2)
    assert ops._wgrad_s2_ok(h // 2, cin)
    dw, db = ops.conv_wgrad(nhwc(x), nhwc(dy), (cout, cin, 3, 3), stride=2, in_slope=0.2, bias_channels=cout)
    assert relerr(dw.cpu(), dw_ref) <= TOL, relerr(dw.cpu(), dw_ref)
    assert relerr(db.cpu(), db_ref) <= TOL, relerr(db.cpu(), db_ref)


@pytest.mark.parametrize("cin,cout,h,w,n", [(192, 96, 5, 64, 1), (192, 96, 9, 65, 2), (288, 192, 8, 32, 2), (64, 32, 7, 10, 3), (160, 96, 11, 150, 1)])
def test_convt_backward_f16_vs_autograd(cin, cout, h, w, n):
    """UpBlock.upsampler: weight + bias gradient on the f16 pipe from the high-res gradient, input gradient on the stride-2 conv kernel."""
    from test_ops_gpu import make_conv, maxerr, nchw
    cp = make_conv(cin, cout, ks=2, stride=2, transposed=True)
    x = rnd(n, cin, h, w, seed=370).requires_grad_(True)
    wt = cp.weight.detach().clone().requires_grad_(True)
    bt = torch.zeros(cout, requires_grad=True)
    dy = rnd(n, cout, 2 * h, 2 * w, seed=371)
    F.conv_transpose2d(x, wt, bt, stride=2).backward(dy)
    cp.cuda()
    assert ops._wgrad_s2_ok(h, cout)
    dw, db = ops.convt_wgrad(nhwc(x.detach()), nhwc(dy), tuple(cp.weight.shape))
    assert relerr(dw.cpu(), wt.grad) <= TOL, relerr(dw.cpu(), wt.grad)
    assert relerr(db.cpu(), bt.grad) <= TOL, relerr(db.cpu(), bt.grad)
    pw = ops.pack_weight(cp.weight, None, transposed=True, dgrad=True)
    assert pw.s2 is not None and pw.s2.f16 is not None
    dx = ops.convt_dgrad(nhwc(dy), pw)
    assert maxerr(nchw(dx), x.grad) <= TOL, maxerr(nchw(dx), x.grad)


def test_wgrad_f16_stride2_reproducible_and_close_to_the_fp32_kernels(monkeypatch):
    n = 4
    x, dy = rnd(n, 96, 48, 140, seed=380), rnd(n, 192, 24, 70, seed=381)
    xd, dyd = nhwc(x), nhwc(dy)
    a, ab = ops.conv_wgrad(xd, dyd, (192, 96, 3, 3), stride=2, bias_channels=192)
    b, bb = ops.conv_wgrad(xd, dyd, (192, 96, 3, 3), stride=2, bias_channels=192)
    assert torch.equal(a, b)
    xl, dyh = nhwc(rnd(n, 192, 24, 70, seed=382)), nhwc(rnd(n, 96, 48, 140, seed=383))
    c, cb = ops.convt_wgrad(xl, dyh, (192, 96, 2, 2))
    d, _ = ops.convt_wgrad(xl, dyh, (192, 96, 2, 2))
    assert torch.equal(c, d)
    monkeypatch.setenv("VIRNET_WGRAD_FORM", "f32")
    assert not ops._wgrad_s2_ok(24, 96)
    a32, ab32 = ops.conv_wgrad(xd, dyd, (192, 96, 3, 3), stride=2, bias_channels=192)
    c32, cb32 = ops.convt_wgrad(xl, dyh, (192, 96, 2, 2))
    assert relerr(a.cpu(), a32.cpu()) <= TOL and relerr(ab.cpu(), ab32.cpu()) <= TOL
    assert relerr(c.cpu(), c32.cpu()) <= TOL and relerr(cb.cpu(), cb32.cpu()) <= TOL


def test_wgrad_f16_stride2_bf16_variant(monkeypatch):
    monkeypatch.setenv("VIRNET_CONV_FORM", "bf16")
    cin, cout, h, w, n = 96, 192, 20, 136, 2
    x, dy = rnd(n, cin, h, w, seed=390), rnd(n, cout, h // 2, w // 2, seed=391)
    wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    F.conv2d(x.to(torch.bfloat16).float(), wt, None, stride=2, padding=1).backward(dy.to(torch.bfloat16).float())
    dw = ops.conv_wgrad(nhwc(x), nhwc(dy), (cout, cin, 3, 3), stride=2)
    assert relerr(dw.cpu(), wt.grad) <= TOL
    xl, dyh = rnd(n, cout, h // 2, w // 2, seed=392), rnd(n, cin, h, w, seed=393)
    wtt = torch.zeros(cout, cin, 2, 2, requires_grad=True)
    F.conv_transpose2d(xl.to(torch.bfloat16).float().requires_grad_(False), wtt, None, stride=2).backward(dyh.to(torch.bfloat16).float())
    dwt, dbt = ops.convt_wgrad(nhwc(xl), nhwc(dyh), (cout, cin, 2, 2))
    assert relerr(dwt.cpu(), wtt.grad) <= TOL
    assert relerr(dbt.cpu(), dyh.sum((0, 2, 3))) <= TOL


@pytest.mark.parametrize("n,h,w,cin,cout", [(32, 128, 128, 96, 192), (32, 64, 64, 192, 288)])
def test_wgrad_f16_stride2_full_size_against_the_fp32_kernels(n, h, w, cin, cout, monkeypatch):
    """configs[4]'s stride-2 layer shapes: the f16-pipe forms against the round-1 fp32 kernels on the same device tensors."""
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.rand(n, h, w, cin, device="cuda", generator=g) - 0.5
    dy = (torch.rand(n, h // 2, w // 2, cout, device="cuda", generator=g) - 0.5) * 0.1
    dw, db = ops.conv_wgrad(x, dy, (cout, cin, 3, 3), stride=2, bias_channels=cout)
    xl = torch.rand(n, h // 2, w // 2, cout, device="cuda", generator=g) - 0.5
    dyh = (torch.rand(n, h, w, cin, device="cuda", generator=g) - 0.5) * 0.1
    dwt, dbt = ops.convt_wgrad(xl, dyh, (cout, cin, 2, 2))
    monkeypatch.setenv("VIRNET_WGRAD_FORM", "f32")
    dw32, db32 = ops.conv_wgrad(x, dy, (cout, cin, 3, 3), stride=2, bias_channels=cout)
    dwt32, dbt32 = ops.convt_wgrad(xl, dyh, (cout, cin, 2, 2))
    for got, ref in ((dw, dw32), (db, db32), (dwt, dwt32), (dbt, dbt32)):
        assert float((got - ref).abs().max()) <= 3e-5 * float(ref.abs().max())
