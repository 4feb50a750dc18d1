"""GPU parity of the Winograd-along-x split-fp16 convolution (csrc/conv_f16_wx4.hip, VIRNET_CONV_FORM=wx4) against the CPU oracle, an
fp64 convolution and the other kernel forms, plus the range guard of the split-fp16 family.

Tolerance.  Each position product is evaluated exactly like conv_f16's (three fp16 products, fp32 accumulation); what the Winograd form
adds is the rounding of the fp32 input transform (coefficients up to 5, |B^T| row sums up to 10) and of the inverse transform
(coefficients up to 8), both relative to the largest magnitude inside the 6-pixel window / the 6 position sums, not to the single
output.  On O(1) data that measures ~2.7x conv_f16's error (5e-6 against 1.8e-6 at 96 channels); the tests hold it to the SAME 2e-5 bar
as every other kernel form and to a window-relative fp32-class bound on data with 9 decades of dynamic range."""
import ctypes as C
import warnings

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


@pytest.fixture(autouse=True, params=[16, 8], ids=["rows16", "rows8"])
def _form(monkeypatch, request):
    # both tile forms of the kernel: 16-row tiles / 8 waves (conv_f16_wx4.hip) and 8-row tiles / 4 waves / weight ring (conv_f16_wx4h.hip);
    # the library picks one per launch size otherwise (virnet_conv_wx4)
    monkeypatch.setenv("VIRNET_WX4_ROWS", str(request.param))
    monkeypatch.delenv("VIRNET_WINOGRAD", raising=False)
    monkeypatch.setenv("VIRNET_CONV_FORM", "wx4")
    monkeypatch.setenv("VIRNET_WX4_MIN_TILES", "0")      # every shape through the kernel under test, also the ones the shape rule
    monkeypatch.setenv("VIRNET_WX4_MIN_COUT", "0")       # would hand to conv_f16 (small images, 32 / 64 channels)
    monkeypatch.setenv("VIRNET_WX4_MIN_FILL", "0")
    monkeypatch.setenv("VIRNET_WX4_MIN_WGS", "0")


class ops_timer:
    def __enter__(self):
        self.t = ops.LaunchTimer()
        ops.set_launch_timer(self.t)
        return self.t

    def __exit__(self, *exc):
        ops.set_launch_timer(None)
        return False


def unpack_wx4(img, rows, ks):
    """packed tensor -> (inverse scales [rows], U = hi + lo as float64 [rows][ks][3 dy][6 positions]) following include/virnet_hip.h."""
    inv = img[:rows].cpu().double()
    raw = img[rows:].cpu().view(torch.float16).double().view(rows // 32, ks // 16, 6, 3, 2, 64, 8)   # [slab][chunk][j][dy][hi|lo][lane][e]
    val = (raw[:, :, :, :, 0] + raw[:, :, :, :, 1]).view(rows // 32, ks // 16, 6, 3, 2, 32, 8)        # [slab][chunk][j][dy][khalf][col][e]
    u = val.permute(0, 5, 1, 4, 6, 3, 2).reshape(rows, ks, 3, 6)                                      # row, k, dy, j
    return inv, u


G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                 dtype=torch.float64)


def test_wx4_weight_image_is_the_scaled_split_transform():
    """virnet_pack_wx4_weight: (hi + lo) * inv_scale reproduces U[dy][j] = sum_b G[j][b] w[dy][b] (formed in fp64) to 2^-21 relative to
    the row's largest entry, scales are powers of two with the scaled row maxima in [8192, 16384); forward and input-gradient packings."""
    cout, cin = 64, 32
    w = rnd(cout, cin, 3, 3, seed=70) * 0.05
    w[5] *= 1e-3
    for dgrad in (False, True):
        ref_w = w.double().flip(2, 3).transpose(0, 1) if dgrad else w.double()
        ref = torch.einsum("jb,rkdb->rkdj", G, ref_w)
        rows, ks = ref.shape[:2]
        inv, got = unpack_wx4(ops.pack_wx4_weight(w.cuda(), dgrad=dgrad), rows, ks)
        assert torch.all(torch.log2(inv) == torch.log2(inv).round())
        scaled_max = ref.abs().amax(dim=(1, 2, 3)) / inv
        assert torch.all((scaled_max >= 8192) & (scaled_max < 16384))
        err = ((got * inv.view(-1, 1, 1, 1) - ref).abs() / ref.abs().amax(dim=(1, 2, 3), keepdim=True)).max()
        assert float(err) <= 2.0 ** -21, float(err)


@pytest.mark.parametrize("c,h,w,n", [(64, 9, 33, 2), (96, 17, 70, 1), (192, 6, 31, 2), (288, 16, 64, 1), (160, 18, 40, 1), (224, 33, 31, 1),
                                      (32, 3, 2, 1), (96, 40, 64, 4), (96, 16, 32, 1)])
def test_wx4_vs_oracle_direct_and_f16x3(monkeypatch, c, h, w, n):
    """Odd sizes (partial tiles in both directions, halo rows beyond the image), every slab mix (3, 3+2, 3+2+2, 2, 1 slabs per
    workgroup), pre-activation, residual + dual store: against the CPU oracle, the fp32 direct kernel and the split-fp16 direct form."""
    cp = make_conv(c, c, seed=80)
    x, res = rnd(n, c, h, w, seed=81), rnd(n, c, h, w, seed=82)
    raw_ref, act_ref = cpu_ref.conv_fused(F.leaky_relu(x, 0.2), cp.weight.detach(), cp.bias.detach(), residual=res, slope=0.25)
    cp.cuda()
    pw = cp.packed()
    assert pw.wx4 is not None and pw.f16 is not None
    kw = dict(in_slope=0.2, res=nhwc(res), want_raw=True, want_act=True, slope=0.25)
    with ops_timer() as t:
        raw_w, act_w = ops.conv_mfma(nhwc(x), pw, **kw)
    assert [k[0] for k in t.summary()] == ["wx4"]
    monkeypatch.setenv("VIRNET_CONV_FORM", "direct")
    raw_d, _ = ops.conv_mfma(nhwc(x), pw, **kw)
    monkeypatch.setenv("VIRNET_CONV_FORM", "f16x3")
    raw_h, _ = ops.conv_mfma(nhwc(x), pw, **kw)
    assert maxerr(nchw(raw_w), raw_ref) <= TOL and maxerr(nchw(act_w), act_ref) <= TOL
    assert maxerr(raw_w.cpu(), raw_d.cpu()) <= TOL and maxerr(raw_w.cpu(), raw_h.cpu()) <= TOL


@pytest.mark.parametrize("epi", ["plain", "act", "res", "mask", "mask_res", "sft_out"])
def test_wx4_epilogue_forms(epi):
    """One instantiation per epilogue form (EPI 0..4) and pre-activation level (PRE 0..2) at 96 channels against fp64."""
    c, n, h, w = 96, 2, 21, 45
    cp = make_conv(c, c, seed=83)
    x, res, saved = rnd(n, c, h, w, seed=84), rnd(n, c, h, w, seed=85), rnd(n, c, h, w, seed=86)
    wd, bd = cp.weight.detach().double(), cp.bias.detach().double()
    cp.cuda()
    pw = cp.packed()
    conv = F.conv2d(x.double(), wd, bd, padding=1)
    if epi == "plain":
        raw, _ = ops.conv_mfma(nhwc(x), pw, want_raw=True)
        ref = conv
    elif epi == "act":
        _, raw = ops.conv_mfma(nhwc(x), pw, want_raw=False, want_act=True, slope=0.2)
        ref = F.leaky_relu(conv, 0.2)
    elif epi == "res":
        raw, _ = ops.conv_mfma(nhwc(x), pw, in_slope=0.2, res=nhwc(res), want_raw=True)
        ref = F.conv2d(F.leaky_relu(x.double(), 0.2), wd, bd, padding=1) + res.double()
    elif epi == "mask":
        raw, _ = ops.conv_mfma(nhwc(x), pw, mask=nhwc(saved), mask_slope=0.2, want_raw=True)
        ref = conv * torch.where(saved > 0, 1.0, 0.2)
    elif epi == "mask_res":
        raw, _ = ops.conv_mfma(nhwc(x), pw, mask=nhwc(saved), mask_slope=0.2, res=nhwc(res), want_raw=True)
        ref = conv * torch.where(saved > 0, 1.0, 0.2) + res.double()
    else:
        imul, iadd = rnd(n, c, seed=87, lo=0.3, hi=1.0), rnd(n, c, seed=88)
        omul, oadd = rnd(n, c, seed=89, lo=0.3, hi=1.0), rnd(n, c, seed=90)
        a_in = F.leaky_relu(x.double() * imul.double().view(n, c, 1, 1) + iadd.double().view(n, c, 1, 1), 0.2)
        r = F.conv2d(a_in, wd, bd, padding=1) + res.double()
        raw, act = ops.conv_mfma(nhwc(x), pw, in_slope=0.2, in_mul=imul.cuda(), in_add=iadd.cuda(), res=nhwc(res), mul=omul.cuda(), add=oadd.cuda(),
                                 want_raw=True, want_act=True, slope=0.2)
        assert float((nchw(act).double() - F.leaky_relu(r * omul.double().view(n, c, 1, 1) + oadd.double().view(n, c, 1, 1), 0.2)).abs().max()) <= TOL
        ref = r
    assert float((nchw(raw).double() - ref).abs().max()) <= TOL


def test_wx4_dgrad_packing_and_backward_epilogue():
    """Input gradient of a res-block conv on the Winograd kernel: dgrad packing + LeakyReLU-derivative mask + residual add."""
    c, n, h, w = 96, 2, 10, 37
    cp = make_conv(c, c, seed=90)
    dy, saved, skip = rnd(n, c, h, w, seed=91), rnd(n, c, h, w, seed=92), rnd(n, c, h, w, seed=93)
    ref = F.conv_transpose2d(dy, cp.weight.detach(), padding=1) * torch.where(saved > 0, 1.0, 0.2) + skip
    cp.cuda()
    pw = cp.packed_dgrad()
    assert pw.wx4 is not None
    with ops_timer() as t:
        dx, _ = ops.conv_mfma(nhwc(dy), pw, mask=nhwc(saved), mask_slope=0.2, res=nhwc(skip), want_raw=True)
    assert [k[0] for k in t.summary()] == ["wx4"]
    assert maxerr(nchw(dx), ref) <= TOL


def test_wx4_randomised_sweep_against_direct_kernel(monkeypatch):
    """Seeded sweep over shapes / channel mixes / epilogue options: catches ordering bugs (plane reuse, partial tiles, odd chunk counts,
    the last chunk's self-refetch) -- the kernel must agree with the fp32 direct kernel run on the same tensors."""
    import os
    g = np.random.Generator(np.random.Philox(key=[78, int(os.environ.get("VIRNET_TEST_SWEEP_KEY", "4"))]))
    chans = [32, 48, 64, 96, 128, 160, 192, 224, 288]
    worst = 0.0
    for case in range(int(os.environ.get("VIRNET_TEST_SWEEP_CASES", "36"))):     # (a longer one-off sweep: set the two variables)
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
        monkeypatch.setenv("VIRNET_CONV_FORM", "wx4")
        pw = cp.packed()
        if pw.wx4 is None:                       # (fewer than 32 input channels: not a layer of this form)
            continue
        with ops_timer() as t:
            outs_w = ops.conv_mfma(x, pw, **kw)
        assert [k[0] for k in t.summary()] == ["wx4"], (case, cin, cout)
        monkeypatch.setenv("VIRNET_CONV_FORM", "direct")
        outs_d = ops.conv_mfma(x, pw, **kw)
        for a, b in zip(outs_w, outs_d):
            if a is not None:
                e = maxerr(a.cpu(), b.cpu())
                worst = max(worst, e)
                assert e <= 5e-5, (case, cin, cout, n, h, w, opts, e)
    assert worst > 0.0


def test_wx4_wide_dynamic_range_window_relative():
    """Activations spanning 1e-6 .. 2e3 and weight rows spanning 1e-4 .. 1.  A Winograd output carries the rounding of its WINDOW: the
    transform mixes the six pixels of a 4-pixel x-tile (and the inverse transform the six position sums), so the bound is relative to
    the largest sum |w||x| over the outputs of the same x-tile -- fp32-class there (a few 1e-6), while a plain-fp16 path would be ~5e-4."""
    c, n, h, w = 96, 1, 12, 40
    g = np.random.Generator(np.random.Philox(key=[5, 5]))
    x = torch.from_numpy((g.standard_normal((n, c, h, w)) * np.exp(g.uniform(-14, 7.5, (n, c, h, w)))).astype(np.float32))
    cp = make_conv(c, c, seed=11)
    with torch.no_grad():
        cp.weight.mul_(torch.from_numpy(np.exp(g.uniform(-9, 0, (c, 1, 1, 1))).astype(np.float32)))
    ref = F.conv2d(x.double(), cp.weight.detach().double(), cp.bias.detach().double(), padding=1)
    # magnitude available to the tile of an output: every pixel of its 6-wide window with the row's largest weight magnitudes
    wmax = cp.weight.detach().double().abs().amax(dim=3, keepdim=True).expand(-1, -1, -1, 3).contiguous()
    mag = F.conv2d(x.double().abs(), wmax, None, padding=1)
    mag = F.max_pool2d(F.pad(mag, (5, 5, 0, 0)), kernel_size=(1, 11), stride=1) + cp.bias.detach().double().abs().view(1, -1, 1, 1)
    cp.cuda()
    raw, _ = ops.conv_mfma(nhwc(x), cp.packed(), want_raw=True)
    rel = float(((nchw(raw).double() - ref).abs() / mag).max())
    print(f"window-relative error vs fp64: wx4 {rel:.2e}")
    assert rel <= 4e-6, rel


def test_range_guard_flags_operands_outside_fp16(monkeypatch):
    """Range guard (virnet_set_range_flag): the split-fp16 kernels raise the sticky flag when a staged operand reaches 65520 -- the direct
    form at |x| >= 65520, the Winograd form already when the TRANSFORMED value does (|x| ~ 1e4: coefficients up to 5) -- and not below."""
    c, n, h, w = 96, 1, 16, 32
    cp = make_conv(c, c, seed=12).cuda()
    dev = torch.device("cuda", torch.cuda.current_device())
    flag = ops.range_flag(dev)
    assert flag is not None
    flag.zero_()
    x = rnd(n, c, h, w, seed=13)
    for form, amp, expect in (("f16x3", 3.0e4, False), ("f16x3", 7.0e4, True), ("wx4", 5.0e3, False), ("wx4", 3.0e4, True)):
        monkeypatch.setenv("VIRNET_CONV_FORM", form)
        xs = x.clone()
        xs[0, 5, 7, 9] = amp
        raw, _ = ops.conv_mfma(nhwc(xs), cp.packed(), want_raw=True)
        assert ops.range_overflowed(dev) is expect, (form, amp)
        assert bool(torch.isfinite(raw).all()) is (not expect), (form, amp)
    assert not ops.range_overflowed(dev)


def test_range_guard_reruns_the_forward_in_fp32(monkeypatch):
    """An image whose activations leave fp16's range inside the network: the module warns, repeats the forward with the fp32 kernels and
    returns THAT result (finite, equal to the fp32 form's); without the guard the split-fp16 forward returns non-finite pixels."""
    import json
    import os
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_images, synth_state_dict
    monkeypatch.delenv("VIRNET_WX4_MIN_TILES", raising=False)
    monkeypatch.delenv("VIRNET_WX4_MIN_COUT", raising=False)
    monkeypatch.delenv("VIRNET_WX4_MIN_FILL", raising=False)
    monkeypatch.setenv("VIRNET_WX4_MIN_WGS", "0")        # (one 64 x 64 image: the launch-size rule would keep it off the Winograd form)
    cfg = dict(n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input", noise_avg=False)
    net = VIRAttResUNet(im_chn=3, sigma_chn=1, **cfg)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}), strict=True)
    net = net.cuda().eval()
    x = synth_images(1, 3, 64, 64).cuda()
    x[0, :, 20:24, 20:24] = 3.0e4              # (the head conv amplifies this beyond 65504 inside RNet)
    with torch.no_grad():
        monkeypatch.setenv("VIRNET_CONV_FORM", "wino")
        mu_ref, sig_ref = net(x)
        assert bool(torch.isfinite(mu_ref).all())
        monkeypatch.setenv("VIRNET_CONV_FORM", "wx4")
        with pytest.warns(RuntimeWarning, match="fp16's range"):
            mu, sigma = net(x)
        assert torch.equal(mu, mu_ref) and torch.equal(sigma, sig_ref)
        monkeypatch.setenv("VIRNET_RANGE_GUARD", "0")
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            mu_raw, _ = net(x)
        assert not bool(torch.isfinite(mu_raw).all())


def test_tile_form_rule_runs_one_of_the_two_forms(monkeypatch):
    """Without VIRNET_WX4_ROWS the library picks the tile height per launch size (csrc/conv_f16_wx4.hip: rounds of workgroups over the
    CUs); whichever it picks, the result is bit for bit the pinned form's, and the rule's two regimes are both exercised."""
    monkeypatch.setenv("VIRNET_WX4_MIN_WGS", "1")          # ("0" is the deterministic switch: it pins the tile height as well)
    c = 96
    cp = make_conv(c, c, seed=21).cuda()
    picked = {}
    for name, (n, h, w) in {"one_small_image": (1, 64, 64), "one_round_of_16_row_tiles": (8, 128, 128)}.items():
        x = nhwc(rnd(n, c, h, w, seed=22))
        outs = {}
        for rows in ("16", "8", None):
            if rows is None:
                monkeypatch.delenv("VIRNET_WX4_ROWS", raising=False)
            else:
                monkeypatch.setenv("VIRNET_WX4_ROWS", rows)
            outs[rows], _ = ops.conv_mfma(x, cp.packed(), want_raw=True)
        assert not torch.equal(outs["16"], outs["8"])          # (the halo rows are transformed with differently associated fp32 sums)
        assert float((outs["16"] - outs["8"]).abs().max()) <= TOL
        picked[name] = "8" if torch.equal(outs[None], outs["8"]) else "16"
        assert torch.equal(outs[None], outs[picked[name]])
    assert picked == {"one_small_image": "8", "one_round_of_16_row_tiles": "16"}, picked


def test_wx4_abi_rejects_bad_descriptors():
    cp = make_conv(96, 96).cuda()
    pw = cp.packed()
    x = torch.zeros(1, 16, 32, 96, device="cuda")
    y = torch.empty(1, 16, 32, 96, device="cuda")

    def desc(**over):
        d = dict(x=nat.ptr(x), wpack=nat.ptr(pw.wx4), bias=0, res=0, mul=0, add=0, mask=0, mask_slope=0.0, in_mul=0, in_add=0, y_raw=nat.ptr(y),
                 y_act=0, n=1, h=16, w=32, cin_pad=96, cout=96, n_pad=96, nrep=1, ks=3, stride=1, epi=nat.EPI_NHWC, nchw_op=0, crop_h=0, crop_w=0,
                 res_sf=1, in_act=0, in_slope=0.0, slope=0.0, clamp_lo=0.0, clamp_hi=0.0)
        d.update(over)
        return nat.ConvDesc(**d)
    lib = nat.load()
    assert lib.virnet_conv_wx4(C.byref(desc()), nat.stream_handle()) == 0
    for bad in (dict(stride=2), dict(ks=1), dict(cout=48, n_pad=48), dict(epi=nat.EPI_NCHW), dict(y_raw=0), dict(cin_pad=24), dict(in_mul=nat.ptr(x)),
                dict(h=0), dict(x=0)):
        assert lib.virnet_conv_wx4(C.byref(desc(**bad)), nat.stream_handle()) != 0, bad
        assert lib.virnet_last_error()
    torch.cuda.synchronize()


@pytest.mark.parametrize("rows", ["16", "8"])
def test_store_policy_does_not_change_a_bit(rows, monkeypatch):
    """VIRNET_NT_STORE_MB (round 5: non-temporal stores / residual loads for tensors larger than the Infinity Cache) is a cache-policy bit on
    the same instructions: the Winograd convolution, the stride-2 and the transposed conv and the entry kernel give identical bits with it
    forced on (1 MB) and off (0)."""
    monkeypatch.setenv("VIRNET_CONV_FORM", "wx4")
    monkeypatch.setenv("VIRNET_WX4_ROWS", rows)
    from virnet_amd.networks.params import ConvParam
    torch.manual_seed(5)
    x = (torch.rand(2, 40, 72, 96, device="cuda") - 0.5)
    res = (torch.rand(2, 40, 72, 96, device="cuda") - 0.5)
    c3 = ConvParam(96, 96, 3).cuda(); cs = ConvParam(96, 192, 3, stride=2).cuda(); ct = ConvParam(96, 64, 2, transposed=True, stride=2).cuda()
    ce = ConvParam(3, 64, 3).cuda()
    img = torch.rand(2, 3, 40, 72, device="cuda")
    br = torch.rand(2, 80, 144, 64, device="cuda")
    outs = {}
    for mb in ("0", "1"):
        monkeypatch.setenv("VIRNET_NT_STORE_MB", mb)
        a = ops.conv_mfma(x, c3.packed(), res=res, want_raw=True)[0]
        b = ops.conv_mfma(x, c3.packed(), in_slope=0.2, want_raw=False, want_act=True)[1]
        c = ops.conv_mfma(x, cs.packed(), stride=2, want_raw=True)[0]
        d = ops.conv_mfma(x, ct.packed(), res=br, want_raw=True)[0]
        e = ops.conv_entry(img, ce.packed(), 40, 72, want_act=True, slope=0.25)
        outs[mb] = [t.clone() for t in (a, b, c, d, e)]
    for u, v in zip(outs["0"], outs["1"]):
        assert torch.equal(u, v)
