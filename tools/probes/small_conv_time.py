#!/usr/bin/env python
"""Back-to-back launch time of ONE small conv launch (the single-image path's under-filled launches): 200 launches in a row / 200.
    python tools/probes/small_conv_time.py [--shapes q0,q1,q2,r1,r2] [--mode pre]   (VIRNET_HIP_LIB / VIRNET_CONV_FORM / VIRNET_WX4_* select builds and forms)"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from virnet_amd import ops  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402
from bench_conv import SHAPES  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--shapes", default="q0,q1,q2,r1,r2"); ap.add_argument("--mode", default="pre"); ap.add_argument("--n", type=int, default=200)
a = ap.parse_args()
for name in a.shapes.split(","):
    n, h, w, c = SHAPES[name]
    cp = ConvParam(c, c, 3).cuda(); pw = cp.packed()
    x = torch.rand(n, h, w, c, device="cuda") - 0.5; res = torch.rand(n, h, w, c, device="cuda") - 0.5
    kw = {"res": dict(res=res, want_raw=True), "pre": dict(in_slope=0.2, want_raw=False, want_act=True)}[a.mode]
    with ops.forward_scope():
        for _ in range(10):
            ops.conv_mfma(x, pw, **kw)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(a.n):
                ops.conv_mfma(x, pw, **kw)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / a.n)
    print(f"{name:4s} {n}x{h}x{w}x{c} {a.mode}: {best * 1e6:7.1f} us per launch (back to back, best of 5)", flush=True)
