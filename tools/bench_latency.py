#!/usr/bin/env python
"""Single-image latency of the script path (scripts/denoising_virnet_syn.py:133-134 runs batch 1, one CBSD68 image at a time).

    python tools/bench_latency.py [--graph]
    python tools/bench_latency.py --all-modes [--iters 200]      every call mode of one box in one process (profiles/r06_latency_modes.txt):
        net(x) as the scripts call it (round 6: replayed from an automatically captured graph from the third call of a shape on),
        net(x) with VIRNET_AUTOGRAPH=0 under the three guard modes, and the explicit net.graphed() object; denoiser shapes + SISR x4 N = 1
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import SYN_CFG  # noqa: E402
from virnet_amd.networks import VIRAttResUNet  # noqa: E402
from virnet_amd.utils.synth import synth_images, synth_state_dict  # noqa: E402


def _time(fwd, iters):
    with torch.no_grad():
        for _ in range(6):
            fwd()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):                                   # best of three blocks of `iters` calls (a box's host jitter is one-sided)
            t0 = time.perf_counter()
            for _ in range(iters):
                fwd()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / iters
            best = dt if best is None else min(best, dt)
    return best * 1e3


def all_modes(iters):
    from bench import build_net
    from virnet_amd import engine, graph
    den = VIRAttResUNet(im_chn=3, sigma_chn=1, **SYN_CFG)
    den.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in den.state_dict().items()}))
    den = den.cuda().eval()
    sr, sd = build_net(torch.device("cuda", 0), "sisr")
    sr.load_state_dict(sd, strict=True)
    sr = sr.cuda().eval()
    cases = [("denoise 481x321", den, synth_images(1, 3, 481, 321).cuda(), ()), ("denoise 256x256", den, synth_images(1, 3, 256, 256).cuda(), ()),
             ("denoise 128x128", den, synth_images(1, 3, 128, 128).cuda(), ()), ("denoise 4x256x256", den, synth_images(4, 3, 256, 256).cuda(), ()),
             ("SISR x4 64x64->256x256", sr, synth_images(1, 3, 64, 64).cuda(), (4,))]
    modes = [("net(x)  [default: auto-replay, sync guard]", {}, None),
             ("net(x)  VIRNET_AUTOGRAPH=0, sync guard", {"VIRNET_AUTOGRAPH": "0"}, None),
             ("net(x)  VIRNET_AUTOGRAPH=0, deferred guard", {"VIRNET_AUTOGRAPH": "0", "VIRNET_GUARD_CHECK": "deferred"}, None),
             ("net(x)  VIRNET_AUTOGRAPH=0, guard off", {"VIRNET_AUTOGRAPH": "0", "VIRNET_RANGE_GUARD": "0"}, None),
             ("net.graphed(check=sync)", {}, "sync"), ("net.graphed(check=off)  [= the kernels' own time]", {}, "off")]
    print(f"# tools/bench_latency.py --all-modes --iters {iters}: ms per forward, best of 3 blocks, one box, one process")
    print("| mode | " + " | ".join(c[0] for c in cases) + " |")
    print("|---|" + "---|" * len(cases))
    table = {}
    for name, env, gcheck in modes:
        for k in ("VIRNET_AUTOGRAPH", "VIRNET_GUARD_CHECK", "VIRNET_RANGE_GUARD"):
            os.environ.pop(k, None)
        os.environ.update(env)
        row = []
        for cname, net, x, extra in cases:
            if gcheck is None:
                ms = _time(lambda: net(x, *extra), iters)
                engine.guard_poll()
            else:
                g = net.graphed(check=gcheck)
                ms = _time(lambda: g(x, *extra), iters)
            row.append(ms)
        table[name] = row
        print(f"| {name} | " + " | ".join(f"{v:.3f}" for v in row) + " |", flush=True)
    for k in ("VIRNET_AUTOGRAPH", "VIRNET_GUARD_CHECK", "VIRNET_RANGE_GUARD"):
        os.environ.pop(k, None)
    st = graph.auto_stats(den)
    print(f"# auto-replay of the denoiser in this run: {st}")
    kern = table["net.graphed(check=off)  [= the kernels' own time]"]
    dflt = table["net(x)  [default: auto-replay, sync guard]"]
    print("# net(x) default minus the kernels' own time: " + ", ".join(f"{c[0]} {d - k:+.3f} ms" for c, d, k in zip(cases, dflt, kern)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", action="store_true", help="replay the forward from a captured hipGraph")
    ap.add_argument("--check", default="sync", choices=["sync", "deferred", "off"], help="range-guard mode of the replayed graph (graph.py)")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--all-modes", action="store_true")
    args = ap.parse_args()
    if args.all_modes:
        return all_modes(args.iters)
    net = VIRAttResUNet(im_chn=3, sigma_chn=1, **SYN_CFG)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}))
    net = net.cuda().eval()
    for shape in [(1, 3, 481, 321), (1, 3, 256, 256), (1, 3, 128, 128), (4, 3, 256, 256)]:
        x = synth_images(*shape).cuda()
        g = net.graphed(check=args.check) if args.graph else None
        with torch.no_grad():
            fwd = (lambda: g(x)) if args.graph else (lambda: net(x))
            for _ in range(5):
                fwd()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                out = fwd()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.iters
        print(f"{('graph/' + args.check) if args.graph else 'eager'} {shape}: {dt * 1e3:7.3f} ms / forward  ({shape[0] / dt:7.1f} img/s)", flush=True)


if __name__ == "__main__":
    main()
