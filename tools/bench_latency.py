#!/usr/bin/env python
"""Single-image latency of the script path (scripts/denoising_virnet_syn.py:133-134 runs batch 1, one CBSD68 image at a time).

    python tools/bench_latency.py [--graph]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", action="store_true", help="replay the forward from a captured hipGraph")
    ap.add_argument("--check", default="sync", choices=["sync", "deferred", "off"], help="range-guard mode of the replayed graph (graph.py)")
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
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
