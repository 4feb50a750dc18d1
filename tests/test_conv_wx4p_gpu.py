"""The persistent, overlapped form of the Winograd-along-x convolution (csrc/conv_f16_wx4p.hip, round 6) against the one-item-per-workgroup
form it replaces (csrc/conv_f16_wx4.hip): the K loop and the arithmetic are the same, so the results must be BIT FOR BIT equal -- whatever
the number of items a workgroup walks (VIRNET_WX4_PERSIST_WGS caps the workgroups per XCD so that small inputs exercise long walks), for
every epilogue it serves (plain / activated store, residual, mask), both pre-activation levels, one / two / three channel blocks per tile,
ragged image sizes and several images.  Reference ops: networks/AttResUNet.py:43,46,55,58.  One case also against an fp64 convolution."""
import pytest
import torch
import torch.nn.functional as F

from virnet_amd import ops
from test_ops_gpu import make_conv, nchw, nhwc, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _form(monkeypatch):
    monkeypatch.setenv("VIRNET_CONV_FORM", "wx4")
    monkeypatch.setenv("VIRNET_WX4_ROWS", "16")
    for k in ("VIRNET_WX4_MIN_TILES", "VIRNET_WX4_MIN_COUT", "VIRNET_WX4_MIN_FILL", "VIRNET_WX4_MIN_WGS"):
        monkeypatch.setenv(k, "0")
    monkeypatch.delenv("VIRNET_WINOGRAD", raising=False)
    monkeypatch.setenv("VIRNET_WX4_NREP", "3")               # (three-slab workgroups also on launches that leave CUs empty: the form under test)
    monkeypatch.setenv("VIRNET_WX4_PERSIST", "1")            # (the form is opt-in: measured slower at the power cap, profiles/r06_probes.md 2)


def _run(x, pw, kind, res, saved):
    if kind == "pre_act":       # conv1-type launch (AttResUNet.py:55): LeakyReLU while staging, activated store
        return ops.conv_mfma(x, pw, in_slope=0.2, want_raw=False, want_act=True, slope=0.2)[1]
    if kind == "res":           # conv2-type (AttResUNet.py:58-59): residual in the epilogue, raw store
        return ops.conv_mfma(x, pw, res=res, want_raw=True)[0]
    if kind == "pre_res":
        return ops.conv_mfma(x, pw, in_slope=0.2, res=res, want_raw=True)[0]
    if kind == "mask":          # input-gradient GEMM's epilogue (training step)
        return ops.conv_mfma(x, pw, mask=saved, mask_slope=0.2, want_raw=True)[0]
    return ops.conv_mfma(x, pw, want_raw=True)[0]


@pytest.mark.parametrize("c,n,h,w,wgs", [
    (96, 3, 64, 96, 1),        # 36 tiles: 8 workgroups walk 4-5 items each
    (96, 2, 45, 70, 2),        # ragged: partial tiles on both axes, two images
    (96, 9, 16, 32, 1),        # one tile per image: every item switches image
    (192, 2, 33, 64, 1),       # two channel blocks per tile: consecutive items share the pixels, not the weights
    (288, 1, 32, 50, 1),       # three channel blocks
    (96, 1, 130, 200, 32),     # the product's grid shape (32 workgroups per XCD), 63 tiles
])
@pytest.mark.parametrize("kind", ["plain", "pre_act", "res", "pre_res", "mask"])
def test_persistent_form_is_bitwise_the_per_item_form(c, n, h, w, wgs, kind, monkeypatch):
    cp = make_conv(c, c, seed=11).cuda()
    pw = cp.packed()
    x, res, saved = (nhwc(rnd(n, c, h, w, seed=s)) for s in (12, 13, 14))
    monkeypatch.setenv("VIRNET_WX4_PERSIST", "0")
    ref = _run(x, pw, kind, res, saved).clone()
    assert ops.wx4_last_plan()["persistent"] is False and ops.wx4_last_plan()["rows"] == 16
    monkeypatch.setenv("VIRNET_WX4_PERSIST", "1")
    monkeypatch.setenv("VIRNET_WX4_PERSIST_MIN", "0")
    monkeypatch.setenv("VIRNET_WX4_PERSIST_WGS", str(wgs))
    got = _run(x, pw, kind, res, saved)
    assert ops.wx4_last_plan() == {"rows": 16, "persistent": True, "slabs": 3, "launches": 1}
    assert torch.equal(got, ref), float((got - ref).abs().max())
    got2 = _run(x, pw, kind, res, saved)                       # (and again: nothing left behind in the workgroup's state)
    assert torch.equal(got2, ref)


def test_persistent_form_vs_fp64(monkeypatch):
    c, n, h, w = 96, 2, 40, 72
    cp = make_conv(c, c, seed=21)
    x, res = rnd(n, c, h, w, seed=22), rnd(n, c, h, w, seed=23)
    ref = F.conv2d(F.leaky_relu(x.double(), 0.2), cp.weight.detach().double(), cp.bias.detach().double(), padding=1) + res.double()
    cp.cuda()
    monkeypatch.setenv("VIRNET_WX4_PERSIST_MIN", "0")
    monkeypatch.setenv("VIRNET_WX4_PERSIST_WGS", "1")
    got = ops.conv_mfma(nhwc(x), cp.packed(), in_slope=0.2, res=nhwc(res), want_raw=True)[0]
    assert ops.wx4_last_plan()["persistent"]
    assert float((nchw(got).double() - ref).abs().max()) <= 2e-5


def test_persistent_form_raises_the_range_flag(monkeypatch):
    """the range guard's sticky flag is reported once per workgroup, after its last item"""
    monkeypatch.setenv("VIRNET_WX4_PERSIST_MIN", "0")
    monkeypatch.setenv("VIRNET_WX4_PERSIST_WGS", "1")
    cp = make_conv(96, 96, seed=31).cuda()
    x = nhwc(rnd(2, 96, 32, 64, seed=32))
    flag = ops.range_flag(x.device)
    flag.zero_()
    ops.conv_mfma(x, cp.packed(), want_raw=True)
    assert ops.wx4_last_plan()["persistent"]
    assert not ops.range_overflowed(x.device)
    x[1, 20, 40, 5] = 3.0e4                                     # in the SECOND item of some workgroup's walk
    ops.conv_mfma(x, cp.packed(), want_raw=True)
    assert ops.range_overflowed(x.device)


def test_persistent_form_is_opt_in(monkeypatch):
    """default: the per-item form (the persistent one measured 2 % slower per launch at the socket's power cap)"""
    monkeypatch.delenv("VIRNET_WX4_PERSIST", raising=False)
    cp = make_conv(96, 96, seed=41).cuda()
    x = nhwc(rnd(20, 96, 64, 128, seed=42))                    # 20 x 16 = 320 tiles... the size rule alone would admit it with MIN = 0
    monkeypatch.setenv("VIRNET_WX4_PERSIST_MIN", "0")
    ops.conv_mfma(x, cp.packed(), want_raw=True)
    assert ops.wx4_last_plan() == {"rows": 16, "persistent": False, "slabs": 3, "launches": 1}
