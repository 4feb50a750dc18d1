"""GPU parity of the split-fp16 convolution (csrc/conv_f16.hip, VIRNET_CONV_FORM=f16x3) against the CPU oracle and the fp32 direct
kernel.  The form is held to the SAME 2e-5 bar as the fp32 kernels: three fp16 products per fp32 product with fp32 accumulation
measure no worse than the fp32 MFMA chain against fp64 (profiles/r02_probes.md)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_ref
from virnet_amd import _native as nat
from virnet_amd import ops
from test_ops_gpu import make_conv, maxerr, nchw, nhwc, rnd

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(autouse=True)
def _form(monkeypatch):
    monkeypatch.delenv("VIRNET_WINOGRAD", raising=False)
    monkeypatch.setenv("VIRNET_CONV_FORM", "f16x3")


def unpack_f16(img, rows, ks):
    """packed tensor -> (inverse scales [rows], hi + lo as float64 [rows][ks][3][3]) following include/virnet_hip.h's layout."""
    inv = img[:rows].cpu().double()
    raw = img[rows:].cpu().view(torch.float16).double().view(rows // 32, ks // 16, 9, 2, 64, 8)   # [slab][chunk][dx*3+dy][hi|lo][lane][e]
    val = raw[:, :, :, 0] + raw[:, :, :, 1]                                                     # [slab][chunk][tap][lane][e]
    val = val.view(rows // 32, ks // 16, 3, 3, 2, 32, 8)                                          # [slab][chunk][dx][dy][khalf][col][e]
    w = val.permute(0, 5, 1, 4, 6, 3, 2).reshape(rows, ks, 3, 3)                                  # row = slab*32+col, k = chunk*16+khalf*8+e, [dy][dx]
    return inv, w


def test_f16_weight_image_is_scaled_split():
    """virnet_pack_f16_weight: (hi + lo) * inv_scale reproduces the fp32 weight to 2^-21 relative, scales are powers of two, the
    scaled row maxima sit in [8192, 16384), forward and input-gradient (flipped, transposed) packings."""
    cout, cin = 64, 32
    w = rnd(cout, cin, 3, 3, seed=70) * 0.05
    w[5] *= 1e-3                                      # a small-norm output channel gets its own scale
    for dgrad in (False, True):
        ref = w.double().flip(2, 3).transpose(0, 1) if dgrad else w.double()
        rows, ks = ref.shape[:2]
        inv, got = unpack_f16(ops.pack_f16_weight(w.cuda(), dgrad=dgrad), rows, ks)
        assert torch.all(torch.log2(inv) == torch.log2(inv).round())
        scaled_max = (ref.abs().amax(dim=(1, 2, 3)) / inv)
        assert torch.all((scaled_max >= 8192) & (scaled_max < 16384))
        err = ((got * inv.view(-1, 1, 1, 1) - ref).abs() / ref.abs().clamp_min(1e-30)).max()
        assert float(err) <= 2.0 ** -21, float(err)


@pytest.mark.parametrize("mrep", ["1", "2"])
@pytest.mark.parametrize("c,h,w,n", [(64, 9, 33, 2), (96, 17, 70, 1), (192, 6, 31, 2), (288, 8, 32, 1), (160, 5, 7, 1), (32, 3, 2, 1),
                                      (96, 40, 64, 4)])
def test_f16x3_vs_direct_and_oracle(monkeypatch, mrep, c, h, w, n):
    """Both tile heights against the fp32 direct kernel and the CPU oracle: odd sizes (partial tiles), every slab mix
    (1, 2, 3 slabs per workgroup; 5 = 160 channels as single slabs), pre-activation, residual + dual store."""
    monkeypatch.setenv("VIRNET_F16_MREP", mrep)
    cp = make_conv(c, c, seed=80)
    x, res = rnd(n, c, h, w, seed=81), rnd(n, c, h, w, seed=82)
    raw_ref, act_ref = cpu_ref.conv_fused(F.leaky_relu(x, 0.2), cp.weight.detach(), cp.bias.detach(), residual=res, slope=0.25)
    cp.cuda()
    pw = cp.packed()
    assert pw.f16 is not None
    kw = dict(in_slope=0.2, res=nhwc(res), want_raw=True, want_act=True, slope=0.25)
    raw_h, act_h = ops.conv_mfma(nhwc(x), pw, **kw)
    monkeypatch.setenv("VIRNET_CONV_FORM", "direct")
    raw_d, _ = ops.conv_mfma(nhwc(x), pw, **kw)
    assert maxerr(nchw(raw_h), raw_ref) <= TOL and maxerr(nchw(act_h), act_ref) <= TOL
    assert maxerr(raw_h.cpu(), raw_d.cpu()) <= TOL


def test_f16x3_wide_dynamic_range(monkeypatch):
    """Activations spanning 1e-6 .. 2e3 and weight rows spanning 1e-4 .. 1: the split keeps fp32-class RELATIVE accuracy
    (error measured against an fp64 convolution, relative to sum |w||x|)."""
    c, n, h, w = 96, 1, 12, 40
    g = np.random.Generator(np.random.Philox(key=[5, 5]))
    x = torch.from_numpy((g.standard_normal((n, c, h, w)) * np.exp(g.uniform(-14, 7.5, (n, c, h, w)))).astype(np.float32))
    cp = make_conv(c, c, seed=11)
    with torch.no_grad():
        cp.weight.mul_(torch.from_numpy(np.exp(g.uniform(-9, 0, (c, 1, 1, 1))).astype(np.float32)))
    ref = F.conv2d(x.double(), cp.weight.detach().double(), cp.bias.detach().double(), padding=1)
    mag = F.conv2d(x.double().abs(), cp.weight.detach().double().abs(), None, padding=1) + cp.bias.detach().double().abs().view(1, -1, 1, 1)
    cp.cuda()
    raw, _ = ops.conv_mfma(nhwc(x), cp.packed(), want_raw=True)
    rel = float(((nchw(raw).double() - ref).abs() / mag).max())
    monkeypatch.setenv("VIRNET_CONV_FORM", "direct")
    raw_d, _ = ops.conv_mfma(nhwc(x), cp.packed(), want_raw=True)
    rel_d = float(((nchw(raw_d).double() - ref).abs() / mag).max())
    print(f"relative error vs fp64: f16x3 {rel:.2e}, fp32 direct {rel_d:.2e}")
    # fp32-class: within a few fp32 ulps of sum|w||x| like the fp32 kernel; a plain fp16 path would be ~5e-4, bf16x3 ~1e-5
    assert rel <= 1e-6 and rel <= 4 * max(rel_d, 1.5e-7), (rel, rel_d)


def test_f16x3_backward_epilogue_and_dgrad_packing():
    """Input gradient of a res-block conv: dgrad packing + LeakyReLU-derivative mask + residual add."""
    c, n, h, w = 96, 2, 10, 37
    cp = make_conv(c, c, seed=90)
    dy, saved, skip = rnd(n, c, h, w, seed=91), rnd(n, c, h, w, seed=92), rnd(n, c, h, w, seed=93)
    ref = F.conv_transpose2d(dy, cp.weight.detach(), padding=1) * torch.where(saved > 0, 1.0, 0.2) + skip
    cp.cuda()
    pw = cp.packed_dgrad()
    assert pw.f16 is not None
    dx, _ = ops.conv_mfma(nhwc(dy), pw, mask=nhwc(saved), mask_slope=0.2, res=nhwc(skip), want_raw=True)
    assert maxerr(nchw(dx), ref) <= TOL


def test_f16x3_abi_rejects_bad_descriptors():
    cp = make_conv(64, 64).cuda()
    pw = cp.packed()
    x = torch.zeros(1, 4, 4, 64, device="cuda")
    y = torch.empty(1, 4, 4, 64, device="cuda")

    def desc(**over):
        d = dict(x=nat.ptr(x), wpack=nat.ptr(pw.f16), bias=0, res=0, mul=0, add=0, mask=0, mask_slope=0.0, in_mul=0, in_add=0, y_raw=nat.ptr(y),
                 y_act=0, n=1, h=4, w=4, cin_pad=64, cout=64, n_pad=64, nrep=1, ks=3, stride=1, epi=nat.EPI_NHWC, nchw_op=0, crop_h=0, crop_w=0,
                 res_sf=1, in_act=0, in_slope=0.0, slope=0.0, clamp_lo=0.0, clamp_hi=0.0)
        d.update(over)
        return nat.ConvDesc(**d)
    lib = nat.load()
    assert lib.virnet_conv_f16(C.byref(desc()), nat.stream_handle()) == 0
    for bad in (dict(stride=3), dict(stride=2, res=nat.ptr(y)), dict(stride=2, h=3), dict(ks=1), dict(cout=48, n_pad=48), dict(epi=nat.EPI_NCHW),
                dict(y_raw=0), dict(cin_pad=24), dict(in_mul=nat.ptr(x))):
        assert lib.virnet_conv_f16(C.byref(desc(**bad)), nat.stream_handle()) != 0, bad
        assert lib.virnet_last_error()
    torch.cuda.synchronize()


def test_f16x3_randomised_sweep_against_direct_kernel(monkeypatch):
    """Seeded sweep over shapes / channel mixes / epilogue options: the split-fp16 kernel (both tile heights) must agree with the
    fp32 direct kernel run on the same tensors -- catches ordering bugs (buffer reuse, partial tiles, odd chunk counts)."""
    g = np.random.Generator(np.random.Philox(key=[78, 3]))
    chans = [32, 48, 64, 96, 128, 160, 192, 224, 288]
    worst = 0.0
    for case in range(40):
        cin, cout = int(g.choice(chans)), int(g.choice([c for c in chans if c % 32 == 0]))
        n, h, w = int(g.integers(1, 4)), int(g.integers(1, 41)), int(g.integers(1, 75))
        opts = dict(pre=bool(g.integers(0, 2)), res=bool(g.integers(0, 2)), mask=bool(g.integers(0, 2)), sft=bool(g.integers(0, 3) == 0),
                    dual=bool(g.integers(0, 2)))
        cp = make_conv(cin, cout, seed=200 + case).cuda()
        cpad = (cin + 15) // 16 * 16
        xc = rnd(n, cin, h, w, seed=300 + case)
        x = nhwc(F.pad(xc, (0, 0, 0, 0, 0, cpad - cin)))
        kw = dict(want_raw=True, want_act=opts["dual"], slope=0.2)
        if opts["pre"] or opts["sft"]:
            kw["in_slope"] = 0.2
        if opts["sft"]:
            kw.update(in_mul=rnd(n, cpad, seed=400 + case, lo=0.3, hi=1.0).cuda(), in_add=rnd(n, cpad, seed=500 + case).cuda(),
                      mul=rnd(n, cout, seed=600 + case, lo=0.3, hi=1.0).cuda(), add=rnd(n, cout, seed=700 + case).cuda())
        if opts["res"]:
            kw["res"] = nhwc(rnd(n, cout, h, w, seed=800 + case))
        if opts["mask"]:
            kw.update(mask=nhwc(rnd(n, cout, h, w, seed=900 + case)), mask_slope=0.25)
        monkeypatch.setenv("VIRNET_CONV_FORM", "f16x3")
        pw = cp.packed()
        assert pw.f16 is not None
        outs = {}
        for form in ("1", "2", "direct"):
            monkeypatch.setenv("VIRNET_CONV_FORM", "direct" if form == "direct" else "f16x3")
            monkeypatch.setenv("VIRNET_F16_MREP", form if form != "direct" else "1")
            outs[form] = ops.conv_mfma(x, pw, **kw)
        for form in ("1", "2"):
            for a, b in zip(outs[form], outs["direct"]):
                if a is not None:
                    e = maxerr(a.cpu(), b.cpu())
                    worst = max(worst, e)
                    assert e <= 5e-5, (case, form, cin, cout, n, h, w, opts, e)
    assert worst > 0.0


@pytest.mark.parametrize("c,cout,h,w,n", [(96, 3, 12, 36, 2), (64, 1, 9, 40, 1), (64, 3, 16, 16, 3), (96, 3, 40, 70, 1), (32, 2, 5, 33, 2)])
def test_f16x3_planar_store(c, cout, h, w, n):
    """Few-output-channel exits on the split-fp16 kernel (one slab, planar store): tail conv + crop + `+ x_in` (AttResUNet.py:139,173),
    the same through a nearest x2 residual (VIRNet.py:83), SNet's last conv with exp(clamp(.)) (DnCNN.py:29, VIRNet.py:43)."""
    cp = make_conv(c, cout, seed=40)
    x = rnd(n, c, h, w, seed=41)
    v = F.conv2d(x, cp.weight.detach(), cp.bias.detach(), padding=1)
    cp.cuda()
    pw = cp.packed()
    assert pw.f16 is not None
    assert maxerr(ops.conv_f16_nchw(nhwc(x), pw, (h, w)).cpu(), v) <= TOL
    ch, cw = h - 2, w - 3
    xin = rnd(n, cout, ch, cw, seed=42)
    assert maxerr(ops.conv_f16_nchw(nhwc(x), pw, (ch, cw), op=nat.NCHW_ADD, res=xin.cuda()).cpu(), v[..., :ch, :cw] + xin) <= TOL
    if h % 2 == 0 and w % 2 == 0:
        xlr = rnd(n, cout, h // 2, w // 2, seed=43)
        ref = v + F.interpolate(xlr, scale_factor=2, mode="nearest")
        assert maxerr(ops.conv_f16_nchw(nhwc(x), pw, (h, w), op=nat.NCHW_ADD, res=xlr.cuda(), res_sf=2).cpu(), ref) <= TOL
    ref = torch.exp(torch.clamp(v, min=-0.5, max=0.7))
    assert maxerr(ops.conv_f16_nchw(nhwc(x), pw, (h, w), op=nat.NCHW_EXPCLAMP, clamp=(-0.5, 0.7)).cpu(), ref) <= TOL


@pytest.mark.parametrize("cin,cout,h,w,n", [(4, 96, 24, 40, 2), (3, 64, 17, 33, 1), (7, 96, 8, 70, 2)])
def test_f16x3_entry_conv_single_chunk(cin, cout, h, w, n):
    """Few-input-channel entries (AttResUNet.head 4/7 -> 96, DnCNN.conv1 3 -> 64): one 16-channel chunk, raw and activated stores."""
    cp = make_conv(cin, cout, seed=44)
    x = rnd(n, cin, h, w, seed=45)
    raw_ref, act_ref = cpu_ref.conv_fused(x, cp.weight.detach(), cp.bias.detach(), slope=0.25)
    cp.cuda()
    pw = cp.packed()
    assert pw.f16 is not None and pw.cin_pad == 16
    xr = nhwc(F.pad(x, (0, 0, 0, 0, 0, 16 - cin)))
    raw, _ = ops.conv_mfma(xr, pw, want_raw=True)
    _, act = ops.conv_mfma(xr, pw, want_raw=False, want_act=True, slope=0.25)
    assert maxerr(nchw(raw), raw_ref) <= TOL and maxerr(nchw(act), act_ref) <= TOL


@pytest.mark.parametrize("cin,cout,h,w,n", [(96, 192, 16, 64, 2), (192, 288, 10, 14, 1), (96, 160, 8, 70, 2), (160, 224, 6, 6, 1), (64, 96, 34, 66, 1),
                                             (32, 32, 2, 2, 1), (48, 64, 12, 36, 3), (96, 192, 64, 128, 4)])
def test_f16x3_stride2_down_conv(monkeypatch, cin, cout, h, w, n):
    """DownBlock.downsampler (AttResUNet.py:67,74) on the split-fp16 stride-2 kernel: 8-wave (6 slabs) and 4-wave (3 / 2 / 1 slabs)
    workgroup forms, partial tiles, odd chunk counts, raw and activated single stores -- against the oracle and the fp32 direct kernel."""
    cp = make_conv(cin, cout, stride=2, seed=60)
    x = rnd(n, cin, h, w, seed=61)
    raw_ref, act_ref = cpu_ref.conv_fused(x, cp.weight.detach(), cp.bias.detach(), stride=2, slope=0.2)
    cp.cuda()
    pw = cp.packed()
    assert pw.f16 is not None
    with ops_timer() as t:
        raw, _ = ops.conv_mfma(nhwc(x), pw, stride=2, want_raw=True)
    assert [k[0] for k in t.summary()] == ["f16x3_s2"]
    _, act = ops.conv_mfma(nhwc(x), pw, stride=2, want_raw=False, want_act=True, slope=0.2)
    assert tuple(raw.shape) == (n, h // 2, w // 2, cout)
    assert maxerr(nchw(raw), raw_ref) <= TOL and maxerr(nchw(act), act_ref) <= TOL
    monkeypatch.setenv("VIRNET_CONV_FORM", "direct")
    raw_d, _ = ops.conv_mfma(nhwc(x), cp.packed(), stride=2, want_raw=True)
    assert maxerr(raw.cpu(), raw_d.cpu()) <= TOL


def test_slab_grouping_rules_do_not_change_a_bit(monkeypatch):
    """The launch-shape rules of under-filled launches regroup the 32-channel slabs of a workgroup (csrc/conv_f16.hip: a mixed 3 + 2 grouping
    becomes one single-slab launch up to 256 workgroups; csrc/conv_f16_s2.hip: one slab per workgroup at <= 64 tiles): channels are
    independent, so every grouping must give the SAME bits -- checked at the shapes of the single-image SISR forward that the rules target."""
    # 160 -> 160 stride-1 at 128 x 128: 128 tiles x (3 + 2 slabs) = 256 workgroups
    cp = make_conv(160, 160, seed=62).cuda()
    x = nhwc(rnd(1, 160, 128, 128, seed=63))
    res = nhwc(rnd(1, 160, 128, 128, seed=64))
    outs = []
    for split in ("0", "256", "100000"):
        monkeypatch.setenv("VIRNET_F16_SPLIT_WGS", split)
        outs.append(ops.conv_mfma(x, cp.packed(), in_slope=0.2, res=res, want_raw=True)[0])
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    monkeypatch.delenv("VIRNET_F16_SPLIT_WGS")
    # 160 -> 224 stride-2 onto 64 x 64: 32 tiles
    cp2 = make_conv(160, 224, stride=2, seed=65).cuda()
    outs = []
    for tiles in ("0", "64", "100000"):
        monkeypatch.setenv("VIRNET_S2_SPLIT_TILES", tiles)
        outs.append(ops.conv_mfma(x, cp2.packed(), stride=2, want_raw=True)[0])
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


class ops_timer:
    """Context manager: route the launches through ops.LaunchTimer to see which kernel form ran."""
    def __enter__(self):
        self.t = ops.LaunchTimer()
        ops.set_launch_timer(self.t)
        return self.t

    def __exit__(self, *exc):
        ops.set_launch_timer(None)
        return False


@pytest.mark.parametrize("cin,cout,h,w,n", [(288, 192, 8, 16, 2), (192, 96, 7, 9, 1), (224, 160, 5, 5, 2), (160, 96, 6, 33, 1), (64, 32, 3, 3, 1),
                                             (96, 64, 16, 16, 3), (288, 192, 32, 32, 4)])
def test_f16x3_transposed_conv(monkeypatch, cin, cout, h, w, n):
    """UpBlock.upsampler + bridge (AttResUNet.py:80,84-87) on the split-fp16 pointwise kernel: contraction lengths that are and are not
    multiples of 48 (zero-padded stages), 8-wave / 4-wave forms, partial pixel tiles, with and without the bridge, activated store."""
    cp = make_conv(cin, cout, ks=2, stride=2, transposed=True, seed=63)
    x, bridge = rnd(n, cin, h, w, seed=64), rnd(n, cout, 2 * h, 2 * w, seed=65)
    raw_ref, act_ref = cpu_ref.conv_transpose_fused(x, cp.weight.detach(), cp.bias.detach(), bridge, slope=0.2)
    plain_ref = F.conv_transpose2d(x, cp.weight.detach(), cp.bias.detach(), stride=2)
    cp.cuda()
    pw = cp.packed()
    assert pw.f16 is not None
    with ops_timer() as t:
        raw, _ = ops.conv_mfma(nhwc(x), pw, res=nhwc(bridge), want_raw=True)
    assert [k[0] for k in t.summary()] == ["f16x3_t"]
    _, act = ops.conv_mfma(nhwc(x), pw, res=nhwc(bridge), want_raw=False, want_act=True, slope=0.2)
    plain, _ = ops.conv_mfma(nhwc(x), pw, want_raw=True)
    assert tuple(raw.shape) == (n, 2 * h, 2 * w, cout)
    assert maxerr(nchw(raw), raw_ref) <= TOL and maxerr(nchw(act), act_ref) <= TOL and maxerr(nchw(plain), plain_ref) <= TOL
    monkeypatch.setenv("VIRNET_CONV_FORM", "direct")
    raw_d, _ = ops.conv_mfma(nhwc(x), cp.packed(), res=nhwc(bridge), want_raw=True)
    assert maxerr(raw.cpu(), raw_d.cpu()) <= TOL


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("c,h,w,n", [(96, 17, 70, 1), (64, 9, 33, 2), (192, 12, 31, 2), (160, 5, 7, 1)])
def test_bf16_operand_variant_is_exactly_bf16_rounded_operands(monkeypatch, c, h, w, n):
    """VIRNET_CONV_FORM=bf16 (BASELINE configs[4]'s training precision): the kernel must equal an fp32 convolution of the bf16-ROUNDED
    pre-activated input and bf16-rounded weights (products of bf16 pairs are exact in fp32, so only the summation order differs) --
    tolerance 2e-5 as for the fp32 forms; against the unrounded fp32 oracle it is off by the stated bf16 operand error (~1e-2)."""
    monkeypatch.setenv("VIRNET_CONV_FORM", "bf16")
    cp = make_conv(c, c, seed=80)
    x, res = rnd(n, c, h, w, seed=81), rnd(n, c, h, w, seed=82)
    a = _bf16_round(F.leaky_relu(x, 0.2))
    raw_ref, act_ref = cpu_ref.conv_fused(a, _bf16_round(cp.weight.detach()), cp.bias.detach(), residual=res, slope=0.25)
    raw_f32, _ = cpu_ref.conv_fused(F.leaky_relu(x, 0.2), cp.weight.detach(), cp.bias.detach(), residual=res, slope=0.25)
    cp.cuda()
    pw = cp.packed()
    assert pw.bf16 is not None and pw.f16 is not None
    with ops_timer() as t:
        raw, act = ops.conv_mfma(nhwc(x), pw, in_slope=0.2, res=nhwc(res), want_raw=True, want_act=True, slope=0.25)
    assert [k[0] for k in t.summary()] == ["bf16"]
    assert maxerr(nchw(raw), raw_ref) <= TOL and maxerr(nchw(act), act_ref) <= TOL
    e32 = maxerr(nchw(raw), raw_f32)
    assert 1e-4 < e32 < 5e-2, e32
    # input-gradient GEMM through the same variant
    dy, saved = rnd(n, c, h, w, seed=91), rnd(n, c, h, w, seed=92)
    ref = F.conv_transpose2d(_bf16_round(dy), _bf16_round(cp.weight.detach().cpu()), padding=1) * torch.where(saved > 0, 1.0, 0.2)
    dx, _ = ops.conv_mfma(nhwc(dy), cp.packed_dgrad(), mask=nhwc(saved), mask_slope=0.2, want_raw=True)
    assert maxerr(nchw(dx), ref) <= TOL
