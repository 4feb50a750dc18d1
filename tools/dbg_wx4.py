"""Debug / parity probe of csrc/conv_f16_wx4.hip on the GPU box: the kernel against an fp64 convolution (torch, on the device) and
the f16x3 kernel, with the error broken down by tile coordinates so an indexing mistake shows where it lives.
Usage: python tools/dbg_wx4.py [--quick]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["VIRNET_WX4_MIN_TILES"] = "0"
os.environ["VIRNET_WX4_MIN_COUT"] = "0"
os.environ["VIRNET_WX4_MIN_FILL"] = "0"
os.environ["VIRNET_WX4_MIN_WGS"] = "0"
from virnet_amd import ops  # noqa: E402
from test_ops_gpu import make_conv, nchw, nhwc, rnd  # noqa: E402


def run(form, x, cp, dgrad=False, **kw):
    os.environ["VIRNET_CONV_FORM"] = form
    pw = cp.packed_dgrad() if dgrad else cp.packed()
    return ops.conv_mfma(x, pw, **kw)


def report(name, got, ref):
    err = (got.double() - ref).abs()
    e = float(err.max())
    print(f"  {name}: max-abs {e:.3e} (ref max {float(ref.abs().max()):.3f})")
    if e > 1e-4:
        n, c, h, w = err.shape
        print("    by x%4      :", [f"{float(err[..., k::4].max()):.2e}" for k in range(4)])
        print("    by x//4 (8) :", [f"{float(err[..., (w_ * 4):(w_ * 4 + 4)].max()):.2e}" for w_ in range(min(8, (w + 3) // 4))])
        print("    by row (18) :", [f"{float(err[:, :, r].max()):.2e}" for r in range(min(h, 18))])
        print("    by slab     :", [f"{float(err[:, s * 32:(s + 1) * 32].max()):.2e}" for s in range(c // 32)])
        print("    by image    :", [f"{float(err[i].max()):.2e}" for i in range(n)])
    return e


def main():
    quick = "--quick" in sys.argv
    shapes = [(32, 32, 1, 16, 32), (96, 96, 1, 16, 32), (96, 96, 2, 37, 70), (64, 64, 2, 9, 33), (192, 192, 1, 20, 45), (160, 160, 1, 18, 40),
              (224, 224, 1, 33, 31), (288, 288, 1, 16, 64), (96, 96, 4, 64, 64)]
    if quick:
        shapes = shapes[:3]
    worst = 0.0
    for cin, cout, n, h, w in shapes:
        print(f"cin {cin} cout {cout} n {n} h {h} w {w}")
        cp = make_conv(cin, cout, seed=80)
        x, res = rnd(n, cin, h, w, seed=81), rnd(n, cout, h, w, seed=82)
        wd, bd = cp.weight.detach().double().cuda(), cp.bias.detach().double().cuda()
        cp.cuda()
        xg = nhwc(x)
        # plain
        ref = F.conv2d(x.double().cuda(), wd, bd, padding=1)
        raw, _ = run("wx4", xg, cp, want_raw=True)
        worst = max(worst, report("plain   wx4", nchw(raw).cuda(), ref))
        raw3, _ = run("f16x3", xg, cp, want_raw=True)
        report("plain f16x3", nchw(raw3).cuda(), ref)
        # pre-activation + residual + dual store
        ref2 = F.conv2d(F.leaky_relu(x.double().cuda(), 0.2), wd, bd, padding=1) + res.double().cuda()
        raw, act = run("wx4", xg, cp, in_slope=0.2, res=nhwc(res), want_raw=True, want_act=True, slope=0.25)
        worst = max(worst, report("pre+res raw", nchw(raw).cuda(), ref2))
        worst = max(worst, report("pre+res act", nchw(act).cuda(), F.leaky_relu(ref2, 0.25)))
        # residual only, single store (EPI 1), activated store (EPI 0)
        raw, _ = run("wx4", xg, cp, in_slope=0.2, res=nhwc(res), want_raw=True)
        worst = max(worst, report("epi1       ", nchw(raw).cuda(), ref2))
        _, act = run("wx4", xg, cp, want_raw=False, want_act=True, slope=0.2)
        worst = max(worst, report("epi0 act   ", nchw(act).cuda(), F.leaky_relu(ref, 0.2)))
        if cin == cout:
            dy, saved = rnd(n, cout, h, w, seed=91), rnd(n, cin, h, w, seed=92)
            refd = F.conv_transpose2d(dy.double().cuda(), wd, padding=1) * torch.where(saved.cuda() > 0, 1.0, 0.2) + res.double().cuda()
            dx, _ = run("wx4", nhwc(dy), cp, dgrad=True, mask=nhwc(saved), mask_slope=0.2, res=nhwc(res), want_raw=True)
            worst = max(worst, report("dgrad epi3 ", nchw(dx).cuda(), refd))
            dx, _ = run("wx4", nhwc(dy), cp, dgrad=True, mask=nhwc(saved), mask_slope=0.2, want_raw=True)
            worst = max(worst, report("dgrad epi2 ", nchw(dx).cuda(), refd - res.double().cuda()))
    print(f"WORST {worst:.3e}")


if __name__ == "__main__":
    main()
