#!/usr/bin/env python
"""Micro-benchmark of single MFMA conv launches (tuning aid; VIRNET_HIP_LIB selects the build).

    python tools/bench_conv.py [--shapes l0,l1,l2] [--iters 20]
Shapes are the denoise-syn res-block convs at batch 32 x 256x256: l0 = 96ch@256^2, l1 = 192ch@128^2, l2 = 288ch@64^2.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virnet_amd import ops  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402

SHAPES = {"l0": (32, 256, 256, 96), "l1": (32, 128, 128, 192), "l2": (32, 64, 64, 288), "s64": (32, 256, 256, 64),
          "l0s": (64, 128, 128, 96), "one": (1, 484, 324, 96),
          "one1": (1, 242, 162, 192), "one2": (1, 121, 81, 288), "q0": (1, 256, 256, 96), "q1": (1, 128, 128, 192),
          "q2": (1, 64, 64, 288), "r1": (1, 64, 64, 192), "r2": (1, 32, 32, 288), "r0": (1, 128, 128, 96), "b4": (4, 256, 256, 96), "b4_1": (4, 128, 128, 192), "b4_2": (4, 64, 64, 288), "t2": (32, 32, 32, 288), "t1": (32, 64, 64, 192),
          "l1s": (64, 64, 64, 192), "l2s": (64, 32, 32, 288), "s0": (16, 256, 256, 96), "s1": (16, 128, 128, 160), "s2": (16, 64, 64, 224)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="l0,l1,l2")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--down", action="store_true", help="time the stride-2 down convs (96->192 @256^2, 192->288 @128^2) instead")
    ap.add_argument("--mode", default="res", choices=["dual", "act", "raw", "res", "pre"])
    ap.add_argument("--ab", default=None, help="VAR=v1,v2,...: interleaved A/B over values of an environment variable the library reads per call")
    args = ap.parse_args()
    torch.manual_seed(0)
    if args.down:
        for (n, h, w, cin, cout) in [(32, 256, 256, 96, 192), (32, 128, 128, 192, 288)]:
            cp = ConvParam(cin, cout, 3, stride=2).cuda()
            x = torch.rand(n, h, w, cin, device="cuda") - 0.5
            pw = cp.packed()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            times = []
            for it in range(args.iters + 3):
                e0.record(); ops.conv_mfma(x, pw, stride=2, want_raw=True); e1.record(); e1.synchronize()
                if it >= 3:
                    times.append(e0.elapsed_time(e1))
            times.sort()
            flops = 2.0 * n * (h // 2) * (w // 2) * cin * cout * 9
            print(f"down {cin}->{cout} @{h}: median {times[len(times) // 2]:.3f} ms  {flops / times[len(times) // 2] / 1e9:.1f} TFLOP/s", flush=True)
        return
    for name in args.shapes.split(","):
        n, h, w, c = SHAPES[name]
        cp = ConvParam(c, c, 3).cuda()
        x = torch.rand(n, h, w, c, device="cuda") - 0.5
        res = torch.rand(n, h, w, c, device="cuda") - 0.5
        if os.environ.get("BENCH_ZEROS") == "1":            # power probe: zero activations AND weights -> nothing toggles in the multipliers
            x.zero_(); res.zero_()
            with torch.no_grad():
                cp.weight.zero_(); cp.bias.zero_()
        kw = {"dual": dict(res=res, want_raw=True, want_act=True), "act": dict(want_raw=False, want_act=True),
              "raw": dict(want_raw=True, want_act=False), "res": dict(res=res, want_raw=True),          # conv2 of a res-block
              "pre": dict(in_slope=0.2, want_raw=False, want_act=True)}[args.mode]                       # conv1 of a res-block
        pw = cp.packed()
        for _ in range(3):
            ops.conv_mfma(x, pw, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if args.ab:
            var, vals = args.ab.split("=")
            vals = vals.split(",")
            res_t = {v: [] for v in vals}
            for _ in range(args.iters):
                for v in vals:
                    os.environ[var] = v
                    e0.record()
                    ops.conv_mfma(x, pw, **kw)
                    e1.record()
                    e1.synchronize()
                    res_t[v].append(e0.elapsed_time(e1))
            flops = 2.0 * n * h * w * c * c * 9
            for v in vals:
                t = sorted(res_t[v])
                print(f"{name:4s} {args.mode:4s} {var}={v:6s} median {t[len(t) // 2]:8.3f} ms  min {t[0]:8.3f} ms  {flops / t[len(t) // 2] / 1e9:7.2f} TFLOP/s", flush=True)
            continue
        times = []
        for _ in range(args.iters):
            e0.record()
            ops.conv_mfma(x, pw, **kw)
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1))
        times.sort()
        med = times[len(times) // 2]
        flops = 2.0 * n * h * w * c * c * 9
        print(f"{os.path.basename(os.environ.get('VIRNET_HIP_LIB', 'default')):24s} {name:4s} {args.mode:4s} median {med:8.3f} ms  "
              f"min {times[0]:8.3f} ms  {flops / med / 1e9:7.2f} TFLOP/s (median)  {flops / times[0] / 1e9:7.2f} (best)", flush=True)


if __name__ == "__main__":
    main()
