#!/usr/bin/env python
"""Persistent form at mid-size batches (the launches are 2-8 rounds of workgroups; is the socket capped there?): whole denoise forward,
N x 256^2, VIRNET_WX4_PERSIST 0 / 1 interleaved, graph replay (kernels' own time)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import SYN_CFG, PowerSampler  # noqa: E402
from virnet_amd.networks import VIRAttResUNet  # noqa: E402
from virnet_amd.utils.synth import synth_images, synth_state_dict  # noqa: E402

net = VIRAttResUNet(im_chn=3, sigma_chn=1, **SYN_CFG)
net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}))
net = net.cuda().eval()
os.environ["VIRNET_AUTOGRAPH"] = "0"
for n in (2, 4, 8, 16):
    x = synth_images(n, 3, 256, 256).cuda()
    gs = {}
    for p in ("0", "1"):
        os.environ["VIRNET_WX4_PERSIST"] = p
        g = net.graphed(check="off")
        with torch.no_grad():
            for _ in range(3):
                g(x)
        gs[p] = g
    res = {"0": [], "1": []}
    with torch.no_grad():
        for rnd in range(7):
            for p in ("0", "1"):
                torch.cuda.synchronize()
                with PowerSampler(0, period=0.01) as ps:
                    t0 = time.perf_counter()
                    for _ in range(60):
                        gs[p](x)
                    torch.cuda.synchronize()
                    res[p].append(((time.perf_counter() - t0) / 60 * 1e3, (ps.summary(skip_s=0.02) or {}).get("socket_w_mean")))
    for p in ("0", "1"):
        v = sorted(res[p])
        print(f"N={n:2d} persist={p}: median {v[len(v)//2][0]:.3f} ms per forward ({n / v[len(v)//2][0] * 1e3:.0f} img/s), {v[len(v)//2][1]} W", flush=True)
