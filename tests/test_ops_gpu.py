"""GPU parity of every C-ABI kernel against the CPU oracle (oracle/cpu_ref.py) on the same seeded inputs.

Tolerance: the contract is 1e-3 max-abs fp32 (BASELINE.json north_star); the fp32 MFMA path differs from the oracle only
by re-association, so the tests hold it to 2e-5 on O(1) data (about 20x the measured noise floor, 50x inside the contract).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_ref
from virnet_amd import _native as nat
from virnet_amd import ops
from virnet_amd.networks.AttResUNet import AttLayer
from virnet_amd.networks.params import ConvParam

pytestmark = pytest.mark.gpu
TOL = 2e-5


def rnd(*shape, seed=0, lo=-1.0, hi=1.0):
    g = np.random.Generator(np.random.Philox(key=[seed, int(np.prod(shape)) & 0xFFFFFFFF]))
    return torch.from_numpy((g.random(size=shape, dtype=np.float32) * (hi - lo) + lo).astype(np.float32))


def nhwc(t):  # NCHW cpu -> NHWC cuda
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):  # NHWC cuda -> NCHW cpu
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def maxerr(a, b):
    return float((a - b).abs().max())


def make_conv(cin, cout, ks=3, stride=1, transposed=False, seed=1):
    cp = ConvParam(cin, cout, ks, transposed=transposed, stride=stride)
    with torch.no_grad():
        fan = cp.weight[0].numel() if not transposed else cin * ks * ks
        cp.weight.copy_(rnd(*cp.weight.shape, seed=seed) * (3.0 / fan) ** 0.5)
        cp.bias.copy_(rnd(cout, seed=seed + 1) * 0.1)
    return cp


@pytest.mark.parametrize("c,h,w,n", [(64, 12, 20, 2), (96, 17, 33, 2), (160, 9, 40, 1), (192, 8, 32, 2), (224, 7, 11, 1),
                                      (288, 12, 20, 1), (128, 5, 65, 1), (96, 64, 96, 1), (96, 40, 64, 8)])
def test_conv3x3_dual_store(c, h, w, n):
    """AttResBlock conv2 shape: conv + bias + residual -> raw, lrelu(raw*mul+add) -> act (AttResUNet.py:57-59)."""
    cp = make_conv(c, c)
    x, res = rnd(n, c, h, w, seed=3), rnd(n, c, h, w, seed=4)
    mul, add = rnd(n, c, seed=5, lo=0.2, hi=1.0), rnd(n, c, seed=6)
    raw_ref, act_ref = cpu_ref.conv_fused(x, cp.weight.detach(), cp.bias.detach(), residual=res,
                                          mul=mul.view(n, c, 1, 1), add=add.view(n, c, 1, 1), slope=0.2)
    cp.cuda()
    raw, act = ops.conv_mfma(nhwc(x), cp.packed(), res=nhwc(res), mul=mul.cuda(), add=add.cuda(), want_raw=True,
                             want_act=True, slope=0.2)
    assert maxerr(nchw(raw), raw_ref) <= TOL and maxerr(nchw(act), act_ref) <= TOL


def test_conv3x3_act_only_and_cross_channels():
    """DnCNN mid conv (post-activation, slope 0.25, DnCNN.py:25-28) and a Cin != Cout case."""
    cp = make_conv(64, 64)
    x = rnd(2, 64, 15, 18, seed=7)
    _, act_ref = cpu_ref.conv_fused(x, cp.weight.detach(), cp.bias.detach(), slope=0.25)
    cp.cuda()
    raw, act = ops.conv_mfma(nhwc(x), cp.packed(), want_raw=False, want_act=True, slope=0.25)
    assert raw is None and maxerr(nchw(act), act_ref) <= TOL
    cp2 = make_conv(32, 96)
    x2 = rnd(1, 32, 10, 37, seed=8)
    raw_ref, _ = cpu_ref.conv_fused(x2, cp2.weight.detach(), cp2.bias.detach())
    cp2.cuda()
    raw2, _ = ops.conv_mfma(nhwc(x2), cp2.packed())
    assert maxerr(nchw(raw2), raw_ref) <= TOL


@pytest.mark.parametrize("c,h,w,n,sft", [(96, 17, 33, 2, False), (96, 12, 40, 2, True), (160, 9, 20, 1, True), (192, 16, 32, 2, False),
                                          (288, 8, 32, 1, True), (64, 5, 70, 1, True)])
def test_conv3x3_fused_preactivation(c, h, w, n, sft):
    """AttResBlock conv1: conv(lrelu(x*mul1+add1)) with the activation applied while staging (AttResUNet.py:54-55), then
    lrelu(.*mul2+add2) in the epilogue (AttResUNet.py:57-58).  x*mul+add != 0 at the border, so a wrong padding order shows."""
    cp = make_conv(c, c)
    x = rnd(n, c, h, w, seed=41)
    mul1, add1 = rnd(n, c, seed=42, lo=0.2, hi=1.0), rnd(n, c, seed=43)
    mul2, add2 = rnd(n, c, seed=44, lo=0.2, hi=1.0), rnd(n, c, seed=45)
    a = x * mul1.view(n, c, 1, 1) + add1.view(n, c, 1, 1) if sft else x
    raw_ref, act_ref = cpu_ref.conv_fused(F.leaky_relu(a, 0.2), cp.weight.detach(), cp.bias.detach(),
                                          mul=mul2.view(n, c, 1, 1) if sft else None, add=add2.view(n, c, 1, 1) if sft else None)
    cp.cuda()
    kw = dict(in_mul=mul1.cuda(), in_add=add1.cuda(), mul=mul2.cuda(), add=add2.cuda()) if sft else {}
    raw, act = ops.conv_mfma(nhwc(x), cp.packed(), in_slope=0.2, want_raw=True, want_act=True, slope=0.2, **kw)
    assert maxerr(nchw(raw), raw_ref) <= TOL and maxerr(nchw(act), act_ref) <= TOL


def test_zero_padding_is_applied_after_activation():
    """A constant field: border outputs must see zeros, not lrelu(add) (SURVEY.md 'pad-after-activation trap')."""
    cp = make_conv(64, 64)
    x = torch.full((1, 64, 8, 8), -1.0)
    a = F.leaky_relu(x, 0.2)
    ref = F.conv2d(a, cp.weight.detach(), cp.bias.detach(), padding=1)
    cp.cuda()
    raw, _ = ops.conv_mfma(nhwc(a), cp.packed())
    assert maxerr(nchw(raw), ref) <= TOL
    raw2, _ = ops.conv_mfma(nhwc(x), cp.packed(), in_slope=0.2)     # same thing with the activation fused into the staging
    assert maxerr(nchw(raw2), ref) <= TOL


@pytest.mark.parametrize("cin,cout,h,w,n", [(96, 192, 16, 24, 2), (192, 288, 8, 64, 1), (96, 160, 10, 6, 1), (160, 224, 4, 68, 2),
                                            (64, 128, 130, 66, 1)])
def test_downsampler_stride2(cin, cout, h, w, n):
    """DownBlock.downsampler: Conv2d(k3,s2,p1) on the raw tensor (AttResUNet.py:67,74)."""
    cp = make_conv(cin, cout, stride=2)
    x = rnd(n, cin, h, w, seed=9)
    raw_ref, act_ref = cpu_ref.conv_fused(x, cp.weight.detach(), cp.bias.detach(), stride=2)
    cp.cuda()
    raw, act = ops.conv_mfma(nhwc(x), cp.packed(), stride=2, want_raw=True, want_act=True)
    assert tuple(raw.shape) == (n, h // 2, w // 2, cout)
    assert maxerr(nchw(raw), raw_ref) <= TOL and maxerr(nchw(act), act_ref) <= TOL


@pytest.mark.parametrize("cin,cout,h,w,n", [(192, 96, 9, 20, 2), (288, 192, 8, 8, 1), (224, 160, 5, 33, 1), (160, 96, 6, 7, 2),
                                            (128, 64, 3, 70, 1)])
def test_upsampler_convtranspose_bridge(cin, cout, h, w, n):
    """UpBlock: ConvTranspose2d(k2,s2) + bridge -> raw, lrelu(raw) (AttResUNet.py:80,84-87)."""
    cp = make_conv(cin, cout, ks=2, stride=2, transposed=True)
    x, bridge = rnd(n, cin, h, w, seed=10), rnd(n, cout, 2 * h, 2 * w, seed=11)
    raw_ref, act_ref = cpu_ref.conv_transpose_fused(x, cp.weight.detach(), cp.bias.detach(), bridge)
    cp.cuda()
    raw, act = ops.conv_mfma(nhwc(x), cp.packed(), res=nhwc(bridge), want_raw=True, want_act=True)
    assert maxerr(nchw(raw), raw_ref) <= TOL and maxerr(nchw(act), act_ref) <= TOL


def test_thin_output_epilogues():
    """tail: conv -> crop -> + x_in (AttResUNet.py:173); SNet last: exp(clamp(.)) (VIRNet.py:43); plain planar store."""
    cp = make_conv(96, 3)
    x, xin = rnd(2, 96, 12, 36, seed=12), rnd(2, 3, 10, 33, seed=13)
    ref = F.conv2d(x, cp.weight.detach(), cp.bias.detach(), padding=1)[..., :10, :33] + xin
    cp.cuda()
    out = ops.conv_mfma_nchw(nhwc(x), cp.packed(), (10, 33), op=nat.NCHW_ADD, res=xin.cuda())
    assert maxerr(out.cpu(), ref) <= TOL
    # nearest-upsampled residual (VIRNet.py:83 fused): res is the low-resolution image
    xlr = rnd(2, 3, 6, 18, seed=14)
    ref2 = F.conv2d(x, cp.weight.detach().cpu(), cp.bias.detach().cpu(), padding=1) + F.interpolate(xlr, scale_factor=2, mode="nearest")
    out2 = ops.conv_mfma_nchw(nhwc(x), cp.packed(), (12, 36), op=nat.NCHW_ADD, res=xlr.cuda(), res_sf=2)
    assert maxerr(out2.cpu(), ref2) <= TOL
    cs = make_conv(64, 1)
    with torch.no_grad():
        cs.weight.mul_(4.0)   # push some values past the clamp
    xs = rnd(1, 64, 9, 40, seed=15, lo=-3, hi=3)
    v = F.conv2d(xs, cs.weight.detach(), cs.bias.detach(), padding=1)
    refs = torch.exp(torch.clamp(v, min=-2.0, max=1.5))
    cs.cuda()
    outs = ops.conv_mfma_nchw(nhwc(xs), cs.packed(), (9, 40), op=nat.NCHW_EXPCLAMP, clamp=(-2.0, 1.5))
    assert float((v > 1.5).sum()) > 0 and float((v < -2.0).sum()) > 0
    assert maxerr(outs.cpu(), refs) <= 5e-6 * float(refs.max())
    outp = ops.conv_mfma_nchw(nhwc(xs), cs.packed(), (9, 40))
    assert maxerr(outp.cpu(), v) <= 1e-4   # |v| reaches ~40 here


@pytest.mark.parametrize("c,cout,h,w,n", [(96, 3, 12, 36, 2), (64, 1, 9, 40, 1), (64, 3, 17, 33, 2), (160, 4, 8, 70, 1)])
def test_thin_conv_kernel(c, cout, h, w, n):
    """Bandwidth-bound few-channel conv (tail / SNet last / KNet tail): plain, crop + residual (also through nearest x2), exp(clamp)."""
    cp = make_conv(c, cout)
    x = rnd(n, c, h, w, seed=50)
    v = F.conv2d(x, cp.weight.detach(), cp.bias.detach(), padding=1)
    cp.cuda()
    pw = cp.packed_thin()
    assert maxerr(ops.conv3x3_thin(nhwc(x), pw, (h, w)).cpu(), v) <= TOL
    ch, cw = h - 2, w - 3
    xin = rnd(n, cout, ch, cw, seed=51)
    assert maxerr(ops.conv3x3_thin(nhwc(x), pw, (ch, cw), op=nat.NCHW_ADD, res=xin.cuda()).cpu(), v[..., :ch, :cw] + xin) <= TOL
    if h % 2 == 0 and w % 2 == 0:
        xlr = rnd(n, cout, h // 2, w // 2, seed=52)
        ref = v + F.interpolate(xlr, scale_factor=2, mode="nearest")
        assert maxerr(ops.conv3x3_thin(nhwc(x), pw, (h, w), op=nat.NCHW_ADD, res=xlr.cuda(), res_sf=2).cpu(), ref) <= TOL
    ref = torch.exp(torch.clamp(v, min=-0.5, max=0.7))
    assert maxerr(ops.conv3x3_thin(nhwc(x), pw, (h, w), op=nat.NCHW_EXPCLAMP, clamp=(-0.5, 0.7)).cpu(), ref) <= TOL
    with pytest.raises(ValueError, match="1..4 output channels"):
        ops.pack_thin_weight(torch.zeros(5, 16, 3, 3, device="cuda"), None)


@pytest.mark.parametrize("form", ["rows", "f16"])
@pytest.mark.parametrize("c,cout,h,w,n", [(96, 3, 12, 36, 2), (64, 1, 9, 40, 1), (64, 3, 17, 33, 2), (64, 2, 8, 32, 3), (160, 3, 8, 70, 1), (96, 3, 1, 1, 1),
                                           (96, 3, 67, 130, 2)])
def test_exit_conv_planar(monkeypatch, form, c, cout, h, w, n):
    """The few-output-channel exits on the split-fp16 path: the taps-as-rows kernel (csrc/conv_exit.hip, default for cout <= 3) and
    conv_f16's planar form -- plain, crop + residual (also through nearest x2), exp(clamp), against fp64 and the oracle's conv."""
    monkeypatch.setenv("VIRNET_CONV_FORM", "wx4")
    monkeypatch.setenv("VIRNET_EXIT_FORM", form)
    cp = make_conv(c, cout)
    x = rnd(n, c, h, w, seed=50)
    v = F.conv2d(x.double(), cp.weight.detach().double(), cp.bias.detach().double(), padding=1).float()
    assert maxerr(cpu_ref.conv_fused(x, cp.weight.detach(), cp.bias.detach())[0], v) <= TOL
    cp.cuda()
    pw = cp.packed()
    assert (pw.exit is not None) == (cout <= 3)
    calls = []
    lib = nat.load()
    real = lib.virnet_conv_exit
    monkeypatch.setattr(lib, "virnet_conv_exit", lambda *a: (calls.append(1), real(*a))[1])
    assert maxerr(ops.conv_f16_nchw(nhwc(x), pw, (h, w)).cpu(), v) <= TOL
    assert bool(calls) == (form == "rows")
    if h > 2 and w > 3:
        ch, cw = h - 2, w - 3
        xin = rnd(n, cout, ch, cw, seed=51)
        assert maxerr(ops.conv_f16_nchw(nhwc(x), pw, (ch, cw), op=nat.NCHW_ADD, res=xin.cuda()).cpu(), v[..., :ch, :cw] + xin) <= TOL
    if h % 2 == 0 and w % 2 == 0:
        xlr = rnd(n, cout, h // 2, w // 2, seed=52)
        ref = v + F.interpolate(xlr, scale_factor=2, mode="nearest")
        assert maxerr(ops.conv_f16_nchw(nhwc(x), pw, (h, w), op=nat.NCHW_ADD, res=xlr.cuda(), res_sf=2).cpu(), ref) <= TOL
    ref = torch.exp(torch.clamp(v, min=-0.5, max=0.7))
    assert maxerr(ops.conv_f16_nchw(nhwc(x), pw, (h, w), op=nat.NCHW_EXPCLAMP, clamp=(-0.5, 0.7)).cpu(), ref) <= TOL


def test_exit_conv_range_flag_and_bad_descriptors():
    cp = make_conv(96, 3).cuda()
    pw = cp.packed()
    dev = torch.device("cuda", torch.cuda.current_device())
    ops.range_flag(dev).zero_()
    x = rnd(1, 96, 8, 40, seed=3)
    x[0, 5, 3, 3] = 7.0e4
    out = ops.conv_f16_nchw(nhwc(x), pw, (8, 40))
    assert ops.range_overflowed(dev) and not bool(torch.isfinite(out).all())
    with pytest.raises(RuntimeError, match="cout <= 3"):
        lib = nat.load()
        nat.check(lib.virnet_pack_exit_weight(nat.ptr(cp.weight.detach()), 4, 96, 96, nat.ptr(pw.exit), nat.stream_handle()), "pack_exit")


def test_pack_input_reflect_upsample_concat():
    """Entry kernel vs pad_input / interpolate / cat (util_net.py:20-25, VIRNet.py:83-95, AttResUNet.py:153)."""
    x, sig = rnd(2, 3, 37, 45, seed=16, lo=0, hi=1), rnd(2, 1, 37, 45, seed=17, lo=0.1, hi=2)
    ref = cpu_ref.pad_to_multiple(torch.cat([x, sig.sqrt()], 1), 4)
    out = ops.pack_input(x.cuda(), 40, 48, map_=sig.cuda(), map_sqrt=True)
    assert tuple(out.shape) == (2, 40, 48, 16)
    assert maxerr(nchw(out)[:, :4], ref) <= 2.4e-7 and float(out[..., 4:].abs().max()) == 0.0
    # SISR: nearest x3, per-image vector repeated, LR sigma map up-sampled x3, pad 27x33 -> 28x36
    xl, vec, sl = rnd(1, 3, 9, 11, seed=18), rnd(1, 3, seed=19), rnd(1, 1, 9, 11, seed=20, lo=0.1, hi=1)
    xu = F.interpolate(xl, scale_factor=3, mode="nearest")
    ex = torch.cat([xu, vec.view(1, 3, 1, 1).repeat(1, 1, 27, 33), F.interpolate(sl.sqrt(), scale_factor=3, mode="nearest")], 1)
    ref2 = cpu_ref.pad_to_multiple(ex, 4)
    out2 = ops.pack_input(xl.cuda(), 28, 36, sf=3, vec=vec.cuda(), map_=sl.cuda(), map_sf=3, map_sqrt=True)
    assert maxerr(nchw(out2)[:, :7], ref2) <= 2.4e-7   # sqrtf may differ by 1 ulp
    with pytest.raises(RuntimeError, match="pad < dim"):   # F.pad reflect contract (util_net.py:24)
        ops.pack_input(rnd(1, 3, 2, 2).cuda(), 4, 4)


def test_knet_pieces():
    """KernelNet.head 9x9/s4 (KNet.py:45), CALayer gate + scale + skip (KNet.py:15-26,38), pooled finishing ops (KNet.py:56-59)."""
    w = rnd(64, 3, 9, 9, seed=21) * 0.06
    x = rnd(2, 3, 21, 30, seed=22, lo=0, hi=1)
    ref = F.conv2d(x, w, None, stride=4, padding=4)
    out = ops.conv_head_s4(x.cuda(), w.cuda())
    assert tuple(out.shape) == (2, 6, 8, 64) and maxerr(nchw(out), ref) <= TOL
    h, skip = rnd(2, 64, 6, 8, seed=23), rnd(2, 64, 6, 8, seed=24)
    w1, b1, w2, b2 = rnd(4, 64, 1, 1, seed=25) * 0.2, rnd(4, seed=26) * 0.1, rnd(64, 4, 1, 1, seed=27), rnd(64, seed=28) * 0.1
    sd = {"ca.body.0.weight": w1, "ca.body.0.bias": b1, "ca.body.2.weight": w2, "ca.body.2.bias": b2}
    ref_rb = cpu_ref.ca_layer(sd, "ca.", h) + skip
    gate = ops.ca_gate(nhwc(h), w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda())
    got = ops.scale_add(nhwc(h), gate, nhwc(skip))
    assert maxerr(nchw(got), ref_rb) <= 1e-6
    fused = ops.ca_scale_add(nhwc(h), w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda(), nhwc(skip))       # the one-launch form of the same
    assert maxerr(nchw(fused), ref_rb) <= 1e-6
    for (hh, ww) in ((16, 16), (31, 33), (64, 64), (65, 64)):                                       # 4 / 16 items per thread; 64 x 64 and larger: over the limit -> gate + scale launches
        hb, sb = rnd(3, 64, hh, ww, seed=31), rnd(3, 64, hh, ww, seed=32)
        ref_b = cpu_ref.ca_layer(sd, "ca.", hb) + sb
        assert maxerr(nchw(ops.ca_scale_add(nhwc(hb), w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda(), nhwc(sb))), ref_b) <= 1e-6, (hh, ww)
    t = rnd(3, 3, 7, 9, seed=29, lo=-12, hi=3)
    m = t.mean(dim=(2, 3))
    refk = torch.cat([torch.exp(torch.clamp(m[:, :2], min=cpu_ref.K_LOG_MIN, max=cpu_ref.K_LOG_MAX)), torch.tanh(m[:, 2:])], 1)
    assert maxerr(ops.gap_nchw(t.cuda(), ops.GAP_KINFO, (cpu_ref.K_LOG_MIN, cpu_ref.K_LOG_MAX)).cpu(), refk) <= 1e-6
    assert maxerr(ops.gap_nchw(t.cuda()).cpu(), m) <= 1e-6
    big = rnd(1, 2, 128, 96, seed=30)
    assert maxerr(ops.gap_nchw(big.cuda(), ops.GAP_EXPCLAMP, (-1.0, 1.0)).cpu(), torch.exp(big.mean(dim=(2, 3)).clamp(-1, 1))) <= 1e-6


@pytest.mark.parametrize("nf,e", [(96, 4), (160, 4), (224, 4), (64, 1)])
def test_sft_generators(nf, e):
    """AttLayer (AttResUNet.py:27-32): per-image vector form and per-pixel form with the nearest-resized extra maps (:168)."""
    att = AttLayer(nf, e)
    sd = {"a." + k: v.detach() for k, v in att.state_dict().items()}
    vec = rnd(3, e, seed=31, lo=0, hi=1.5)
    mul_ref, add_ref = cpu_ref.att_layer(sd, "a.", vec.view(3, e, 1, 1))
    att.cuda()
    mul, add = ops.sft_vec(vec.cuda(), att)
    assert maxerr(mul.cpu(), mul_ref.view(3, nf)) <= 1e-6 and maxerr(add.cpu(), add_ref.view(3, nf)) <= 1e-6
    # per-pixel: full-res records [x(3) | extra(e)], level 1 (step 2)
    n, H, W = 2, 12, 20
    full = rnd(n, 3 + e, H, W, seed=32, lo=0, hi=1)
    rec = torch.zeros(n, H, W, 16)
    rec[..., :3 + e] = full.permute(0, 2, 3, 1)
    raw = rnd(n, nf, H // 2, W // 2, seed=33)
    ex = F.interpolate(full[:, 3:], (H // 2, W // 2), mode="nearest")
    m, a = cpu_ref.att_layer(sd, "a.", ex)
    ref = F.leaky_relu(raw * m + a, 0.2)
    got = ops.sft_apply(nhwc(raw), rec.cuda(), 3, e, 2, att)
    assert maxerr(nchw(got), ref) <= 2e-6


def test_sft_vec_multi_is_bitwise_the_per_layer_launches():
    """virnet_sft_vec_multi (every SFT layer of a down path in one launch, 16 layers per launch) against one virnet_sft_vec per layer."""
    torch.manual_seed(34)
    atts = [AttLayer(nf, 4).cuda() for nf in (96, 96, 160, 160, 224, 224) * 3]      # 18 layers: two launches
    vec = rnd(3, 4, seed=35, lo=0, hi=1.5).cuda()
    got = ops.sft_vec_multi(vec, atts)
    assert len(got) == len(atts)
    for att, (mul, add) in zip(atts, got):
        m1, a1 = ops.sft_vec(vec, att)
        assert torch.equal(mul, m1) and torch.equal(add, a1)
    with pytest.raises(ValueError, match="conditioning vector"):
        ops.sft_vec_multi(rnd(3, 5, seed=36).cuda(), atts[:2])


def test_abi_rejects_bad_shapes():
    cp = make_conv(96, 96).cuda()
    with pytest.raises(ValueError, match="channels"):
        ops.conv_mfma(torch.zeros(1, 8, 8, 64, device="cuda"), cp.packed())
    with pytest.raises(RuntimeError, match="must be even"):
        ops.conv_mfma(torch.zeros(1, 7, 8, 96, device="cuda"), make_conv(96, 192, stride=2).cuda().packed(), stride=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv_mfma(torch.zeros(1, 8, 8, 96), cp.packed())


def test_repack_follows_parameter_updates():
    cp = make_conv(64, 64).cuda()
    x = rnd(1, 64, 8, 32, seed=40)
    r0, _ = ops.conv_mfma(nhwc(x), cp.packed())
    with torch.no_grad():
        cp.weight.mul_(2.0)      # in-place update bumps ._version -> repack
        cp.bias.zero_()
    r1, _ = ops.conv_mfma(nhwc(x), cp.packed())
    ref = F.conv2d(x, cp.weight.detach().cpu(), None, padding=1)
    assert maxerr(nchw(r1), ref) <= TOL and maxerr(nchw(r0), ref) > 1e-2


def test_conv_tensor_beyond_4gib():
    """Image bases are 64-bit, in-image offsets 32-bit: a 4.4 GB NHWC tensor (> 2**32 bytes, > 2**30 elements) must convolve
    its LAST image exactly like that image alone (288 GB parts hold batches this large; BASELINE configs[2] is 6.4 GB/tensor)."""
    n, h, w, c = 44, 512, 512, 96
    assert n * h * w * c * 4 > 2 ** 32
    cp = make_conv(c, c).cuda()
    x = torch.empty(n, h, w, c, device="cuda")
    x.uniform_(-1, 1)
    raw, _ = ops.conv_mfma(x, cp.packed(), in_slope=0.2, res=x, want_raw=True)
    for i in (0, n - 1):
        ri, _ = ops.conv_mfma(x[i:i + 1].contiguous(), cp.packed(), in_slope=0.2, res=x[i:i + 1].contiguous(), want_raw=True)
        assert torch.equal(ri[0], raw[i])
    ref = F.conv2d(F.leaky_relu(x[n - 1, :40, :40].permute(2, 0, 1)[None].cpu(), 0.2), cp.weight.detach().cpu(), cp.bias.detach().cpu(), padding=1)
    assert maxerr(raw[n - 1, :38, :38].permute(2, 0, 1).cpu(), ref[0, :, :38, :38] + x[n - 1, :38, :38].permute(2, 0, 1).cpu()) <= TOL


# ---- Winograd F(2x2,3x3) form of the stride-1 3x3 conv (csrc/wino.hip) ----------------------------------------------------------
def test_winograd_weight_image_is_G_g_Gt():
    """virnet_pack_wino_weight: U = G g G^T per channel pair, forward and input-gradient (flipped, transposed) packings."""
    cout, cin = 64, 32
    w = rnd(cout, cin, 3, 3, seed=70)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    for dgrad in (False, True):
        g = w.double().flip(2, 3).transpose(0, 1) if dgrad else w.double()          # [rows][k][3][3]
        ref = torch.einsum("ia,rkab,jb->rkij", G, g, G).float()                       # [rows][k][4][4]
        rows, ks = ref.shape[:2]
        u = ops.pack_wino_weight(w.cuda(), dgrad=dgrad).cpu().view(rows // 32, ks // 4, 16, 2, 32, 2)
        # [slab][chunk][pos][half][col][s]: row = slab*32+col, k = chunk*4 + half*2 + s
        got = u.permute(0, 4, 1, 3, 5, 2).reshape(rows, ks, 4, 4)
        assert maxerr(got, ref) <= 1e-6


@pytest.mark.parametrize("nw", ["4", "8"])
@pytest.mark.parametrize("c,h,w,n", [(64, 9, 33, 2), (96, 17, 70, 1), (192, 6, 31, 2), (288, 8, 32, 1), (160, 5, 7, 1), (32, 3, 2, 1)])
def test_winograd_vs_direct_and_oracle(monkeypatch, nw, c, h, w, n):
    """Both workgroup forms of the Winograd kernel against the direct MFMA kernel and the CPU oracle: odd sizes (partial tiles and
    units), every channel-block mix (64: role A only, 32/160: role B present, 96/288: both), residual + dual store."""
    monkeypatch.setenv("VIRNET_WINO_NW", nw)
    monkeypatch.setenv("VIRNET_WINOGRAD", "1")
    cp = make_conv(c, c, seed=80)
    x, res = rnd(n, c, h, w, seed=81), rnd(n, c, h, w, seed=82)
    raw_ref, act_ref = cpu_ref.conv_fused(F.leaky_relu(x, 0.2), cp.weight.detach(), cp.bias.detach(), residual=res, slope=0.25)
    cp.cuda()
    pw = cp.packed()
    assert pw.wino is not None
    kw = dict(in_slope=0.2, res=nhwc(res), want_raw=True, want_act=True, slope=0.25)
    raw_w, act_w = ops.conv_mfma(nhwc(x), pw, **kw)
    monkeypatch.setenv("VIRNET_WINOGRAD", "0")
    raw_d, act_d = ops.conv_mfma(nhwc(x), pw, **kw)
    assert maxerr(nchw(raw_w), raw_ref) <= TOL and maxerr(nchw(act_w), act_ref) <= TOL
    assert maxerr(nchw(raw_d), raw_ref) <= TOL
    assert maxerr(raw_w.cpu(), raw_d.cpu()) <= TOL


def test_winograd_backward_epilogue_and_dgrad_packing(monkeypatch):
    """Input gradient of a res-block conv through the Winograd kernel: dgrad packing + LeakyReLU-derivative mask + residual add."""
    monkeypatch.setenv("VIRNET_WINOGRAD", "1")
    c, n, h, w = 96, 2, 10, 37
    cp = make_conv(c, c, seed=90)
    dy, saved, skip = rnd(n, c, h, w, seed=91), rnd(n, c, h, w, seed=92), rnd(n, c, h, w, seed=93)
    ref = F.conv_transpose2d(dy, cp.weight.detach(), padding=1) * torch.where(saved > 0, 1.0, 0.2) + skip
    cp.cuda()
    pw = cp.packed_dgrad()
    assert pw.wino is not None
    dx, _ = ops.conv_mfma(nhwc(dy), pw, mask=nhwc(saved), mask_slope=0.2, res=nhwc(skip), want_raw=True)
    assert maxerr(nchw(dx), ref) <= TOL


def test_winograd_abi_rejects_bad_descriptors(monkeypatch):
    monkeypatch.setenv("VIRNET_WINOGRAD", "1")
    cp = make_conv(64, 64).cuda()
    pw = cp.packed()
    x = torch.zeros(1, 4, 4, 64, device="cuda")
    y = torch.empty(1, 4, 4, 64, device="cuda")
    import ctypes as C
    def desc(**over):
        d = dict(x=nat.ptr(x), wpack=nat.ptr(pw.wino), bias=0, res=0, mul=0, add=0, mask=0, mask_slope=0.0, in_mul=0, in_add=0, y_raw=nat.ptr(y),
                 y_act=0, n=1, h=4, w=4, cin_pad=64, cout=64, n_pad=64, nrep=1, ks=3, stride=1, epi=nat.EPI_NHWC, nchw_op=0, crop_h=0, crop_w=0,
                 res_sf=1, in_act=0, in_slope=0.0, slope=0.0, clamp_lo=0.0, clamp_hi=0.0)
        d.update(over)
        return nat.ConvDesc(**d)
    lib = nat.load()
    assert lib.virnet_conv_wino(C.byref(desc()), nat.stream_handle()) == 0
    for bad in (dict(stride=2), dict(ks=1), dict(cout=48, n_pad=48), dict(epi=nat.EPI_NCHW), dict(y_raw=0), dict(cin_pad=24), dict(in_mul=nat.ptr(x))):
        assert lib.virnet_conv_wino(C.byref(desc(**bad)), nat.stream_handle()) != 0, bad
        assert lib.virnet_last_error()
    torch.cuda.synchronize()


def test_winograd_randomised_sweep_against_direct_kernel(monkeypatch):
    """Seeded sweep over shapes / channel mixes / epilogue options: the Winograd kernel (both workgroup forms) must agree with the
    direct MFMA kernel run on the same tensors -- catches ordering bugs (buffer reuse, partial units) that fixed shapes can miss."""
    g = np.random.Generator(np.random.Philox(key=[77, 3]))
    chans = [32, 64, 96, 128, 160, 192, 224, 288]
    worst = 0.0
    for case in range(40):
        cin, cout = int(g.choice(chans)), int(g.choice(chans))
        n, h, w = int(g.integers(1, 4)), int(g.integers(1, 41)), int(g.integers(1, 75))
        opts = dict(pre=bool(g.integers(0, 2)), res=bool(g.integers(0, 2)), mask=bool(g.integers(0, 2)), sft=bool(g.integers(0, 3) == 0),
                    dual=bool(g.integers(0, 2)))
        cp = make_conv(cin, cout, seed=200 + case).cuda()
        x = nhwc(rnd(n, cin, h, w, seed=300 + case))
        kw = dict(want_raw=True, want_act=opts["dual"], slope=0.2)
        if opts["pre"] or opts["sft"]:
            kw["in_slope"] = 0.2
        if opts["sft"]:
            kw.update(in_mul=rnd(n, cin, seed=400 + case, lo=0.3, hi=1.0).cuda(), in_add=rnd(n, cin, seed=500 + case).cuda(),
                      mul=rnd(n, cout, seed=600 + case, lo=0.3, hi=1.0).cuda(), add=rnd(n, cout, seed=700 + case).cuda())
        if opts["res"]:
            kw["res"] = nhwc(rnd(n, cout, h, w, seed=800 + case))
        if opts["mask"]:
            kw.update(mask=nhwc(rnd(n, cout, h, w, seed=900 + case)), mask_slope=0.25)
        monkeypatch.setenv("VIRNET_WINOGRAD", "1")
        pw = cp.packed()
        assert pw.wino is not None
        outs = {}
        for form in ("4", "8", "direct"):
            monkeypatch.setenv("VIRNET_WINOGRAD", "0" if form == "direct" else "1")
            monkeypatch.setenv("VIRNET_WINO_NW", form if form != "direct" else "4")
            outs[form] = ops.conv_mfma(x, pw, **kw)
        for form in ("4", "8"):
            for a, b in zip(outs[form], outs["direct"]):
                if a is not None:
                    e = maxerr(a.cpu(), b.cpu())
                    worst = max(worst, e)
                    assert e <= 5e-5, (case, form, cin, cout, n, h, w, opts, e)
    assert worst > 0.0            # the two algorithms are different roundings of the same sums


ENTRY_CASES = [
    (2, 3, 45, 70, 1, 48, 72, 0, 1, 1, True, 96, False),      # denoiser head: image + sqrt(sigma) map, reflect pad on both axes
    (1, 3, 64, 64, 1, 64, 64, 0, 0, 1, False, 64, True),      # DnCNN.conv1: image only, activated store
    (3, 3, 16, 20, 4, 64, 80, 4, 0, 1, False, 96, False),     # SISR head: nearest x4, kernel / noise vector (7 channels: two k-steps per row)
    (2, 3, 9, 7, 3, 28, 24, 3, 1, 3, True, 96, False),        # SISR with a per-pixel variance map (nearest x3) + vector, padded
    (5, 1, 33, 31, 2, 68, 64, 0, 2, 2, False, 32, True),      # one image channel, two map channels
    (1, 3, 7, 5, 1, 8, 8, 0, 1, 1, True, 96, False),          # a single partial tile
    (2, 3, 40, 100, 1, 40, 100, 0, 1, 1, True, 96, False),    # width not a multiple of the tile: the last tile's rows are cut
]


@pytest.mark.parametrize("n,c0,h,w,sf,hp,wp,ev,em,msf,msqrt,cout,act", ENTRY_CASES)
def test_conv_entry_kernel_vs_oracle_and_two_launch_path(n, c0, h, w, sf, hp, wp, ev, em, msf, msqrt, cout, act, monkeypatch):
    """virnet_conv_entry (csrc/conv_entry.hip, round 5: the store-bound entry kernel, K walked one kernel row per MFMA k-step) against an fp64
    convolution of the packed record (AttResUNet.py:153-155 / DnCNN.py:38 on util_net.py:20-25's padding, VIRNet.py:83,94,44) and
    against the two-launch path: the same split-fp16 products in another summation order -- fp32 rounding apart, nothing else."""
    cp = make_conv(c0 + ev + em, cout, seed=61).cuda()
    x = rnd(n, c0, h, w, seed=62, lo=0.0, hi=1.0).cuda()
    vec = rnd(n, ev, seed=63).cuda() if ev else None
    mp = rnd(n, em, h * sf // msf, w * sf // msf, seed=64, lo=0.01, hi=2.0).cuda() if em else None
    kw = dict(sf=sf, vec=vec, map_=mp, map_sf=msf, map_sqrt=msqrt, want_act=act, slope=0.25)
    assert cp.packed().entry is not None
    got = ops.conv_entry(x, cp.packed(), hp, wp, **kw)
    assert tuple(got.shape) == (n, hp, wp, cout)
    monkeypatch.setenv("VIRNET_ENTRY_FUSED", "0")
    two = ops.conv_entry(x, cp.packed(), hp, wp, **kw)
    rec = ops.pack_input(x, hp, wp, sf=sf, vec=vec, map_=mp, map_sf=msf, map_sqrt=msqrt)          # NHWC [n][hp][wp][16]
    ref = torch.nn.functional.conv2d(rec[..., :c0 + ev + em].permute(0, 3, 1, 2).cpu().double(), cp.weight.detach().cpu().double(),
                                     cp.bias.detach().cpu().double(), padding=1)
    if act:
        ref = torch.nn.functional.leaky_relu(ref, 0.25)
    ref = ref.permute(0, 2, 3, 1).float()
    scale = max(1.0, float(ref.abs().max()))
    assert maxerr(got.cpu(), ref) <= 2e-6 * scale, maxerr(got.cpu(), ref)
    assert maxerr(got.cpu(), two.cpu()) <= 2e-6 * scale


@pytest.mark.parametrize("n,c0,h,w,sf,hp,wp,ev,em,msf,msqrt,cout,act", [
    (2, 3, 45, 70, 1, 48, 72, 0, 1, 1, True, 96, False),      # denoiser head: image + sqrt(sigma) map, reflect pad on both axes
    (1, 3, 64, 64, 1, 64, 64, 0, 0, 1, False, 64, True),      # DnCNN.conv1: image only, activated store
    (3, 3, 16, 20, 4, 64, 80, 4, 0, 1, False, 96, False),     # SISR head: nearest x4, kernel / noise vector
    (2, 3, 9, 7, 3, 28, 24, 3, 1, 3, True, 96, False),        # SISR with a per-pixel variance map (nearest x3) + vector, padded
    (5, 1, 33, 31, 2, 68, 64, 0, 2, 2, False, 32, True),      # one image channel, two map channels
])
def test_conv_entry_is_bitwise_pack_plus_conv(n, c0, h, w, sf, hp, wp, ev, em, msf, msqrt, cout, act, monkeypatch):
    """virnet_conv_f16_entry (the entry packing folded into the first conv's staging) against virnet_pack_input + virnet_conv_f16: the
    same values are staged, so the result is bit for bit the two-launch one (util_net.py:20-25 reflect pad, VIRNet.py:83,94 nearest
    up-sampling, VIRNet.py:44 sqrt, AttResUNet.py:153 concat)."""
    cp = make_conv(c0 + ev + em, cout, seed=61).cuda()
    x = rnd(n, c0, h, w, seed=62, lo=0.0, hi=1.0).cuda()
    vec = rnd(n, ev, seed=63).cuda() if ev else None
    mp = rnd(n, em, h * sf // msf, w * sf // msf, seed=64, lo=0.01, hi=2.0).cuda() if em else None
    kw = dict(sf=sf, vec=vec, map_=mp, map_sf=msf, map_sqrt=msqrt, want_act=act, slope=0.25)
    monkeypatch.setenv("VIRNET_ENTRY_FORM", "f16")            # (round 4's fused form; the default since round 5 is csrc/conv_entry.hip: test above)
    fused = ops.conv_entry(x, cp.packed(), hp, wp, **kw)
    monkeypatch.setenv("VIRNET_ENTRY_FUSED", "0")
    two = ops.conv_entry(x, cp.packed(), hp, wp, **kw)
    rec = ops.pack_input(x, hp, wp, sf=sf, vec=vec, map_=mp, map_sf=msf, map_sqrt=msqrt)
    raw, a2 = ops.conv_mfma(rec, cp.packed(), want_raw=not act, want_act=act, slope=0.25)
    assert torch.equal(two, a2 if act else raw)
    assert tuple(fused.shape) == (n, hp, wp, cout) and torch.equal(fused, two)


def test_conv_entry_abi_rejects_bad_descriptors():
    lib = nat.load()
    cp = make_conv(3, 64).cuda()
    pw = cp.packed()
    x = torch.zeros(1, 3, 8, 8, device="cuda")
    y = torch.empty(1, 8, 8, 64, device="cuda")
    import ctypes as C
    def desc(**over):
        kw = dict(x=nat.ptr(x), wpack=nat.ptr(pw.f16), bias=nat.ptr(pw.bias), res=0, mul=0, add=0, mask=0, mask_slope=0.0, in_mul=0, in_add=0,
                  y_raw=nat.ptr(y), y_act=0, n=1, h=8, w=8, cin_pad=16, cout=64, n_pad=64, nrep=pw.nrep, ks=3, stride=1, epi=nat.EPI_NHWC,
                  nchw_op=0, crop_h=0, crop_w=0, res_sf=1, in_act=0, in_slope=0.0, slope=0.2, clamp_lo=0.0, clamp_hi=0.0)
        kw.update(over)
        return nat.ConvDesc(**kw)
    def ent(**over):
        kw = dict(x=nat.ptr(x), vec=0, map=0, out=0, n=1, c0=3, h=8, w=8, sf=1, ev=0, em=0, mh=0, mw=0, msf=1, map_sqrt=0, hp=8, wp=8, zero_pad=0)
        kw.update(over)
        return nat.PackDesc(**kw)
    st = nat.stream_handle()
    assert lib.virnet_conv_f16_entry(C.byref(desc()), C.byref(ent()), st) == 0
    assert lib.virnet_conv_f16_entry(C.byref(desc(res=nat.ptr(y))), C.byref(ent()), st) != 0          # residual: not an entry
    assert lib.virnet_conv_f16_entry(C.byref(desc()), C.byref(ent(hp=12)), st) != 0                   # geometry mismatch
    assert lib.virnet_conv_f16_entry(C.byref(desc()), C.byref(ent(c0=6, ev=3)), st) != 0              # more than 8 record channels
    assert lib.virnet_conv_f16_entry(C.byref(desc()), C.byref(ent(ev=2)), st) != 0                    # vector channels without a vector
    torch.cuda.synchronize()


def test_conv_entry_image_packed_for_another_channel_count_is_loud():
    """slot j = dx*cin + ch is baked into virnet_pack_entry_weight's image and the host cannot look into device memory --
    the image carries a trailer (cin, k-steps per row) that the kernel compares with the launch's channel count: a mismatch gives NaN,
    not a plausible picture.  (ops.conv_entry checks pw.cin_real itself; this is the raw C-ABI caller.)"""
    import ctypes as C
    lib = nat.load()
    cp3, cp4 = make_conv(3, 64, seed=5).cuda(), make_conv(4, 64, seed=6).cuda()
    x = rnd(1, 3, 8, 8, seed=7, lo=0.0, hi=1.0).cuda()
    y = torch.zeros(1, 8, 8, 64, device="cuda")
    def call(pw):
        d = nat.ConvDesc(x=nat.ptr(x), wpack=nat.ptr(pw.entry), bias=nat.ptr(pw.bias), res=0, mul=0, add=0, mask=0, mask_slope=0.0, in_mul=0, in_add=0,
                         y_raw=nat.ptr(y), y_act=0, n=1, h=8, w=8, cin_pad=16, cout=64, n_pad=64, nrep=pw.nrep, ks=3, stride=1, epi=nat.EPI_NHWC,
                         nchw_op=0, crop_h=0, crop_w=0, res_sf=1, in_act=0, in_slope=0.0, slope=0.2, clamp_lo=0.0, clamp_hi=0.0)
        e = nat.PackDesc(x=nat.ptr(x), vec=0, map=0, out=0, n=1, c0=3, h=8, w=8, sf=1, ev=0, em=0, mh=0, mw=0, msf=1, map_sqrt=0, hp=8, wp=8, zero_pad=0)
        assert lib.virnet_conv_entry(C.byref(d), C.byref(e), nat.stream_handle()) == 0
        torch.cuda.synchronize()
        return y.clone()
    good = call(cp3.packed())
    assert bool(torch.isfinite(good).all()) and float(good.abs().max()) > 0
    bad = call(cp4.packed())                                  # packed for 4 input channels, launched with 3
    assert bool(torch.isnan(bad).all())
    tr = cp3.packed().entry[-4:].view(torch.int32).tolist()
    assert tr[:3] == [3, 1, 64]
