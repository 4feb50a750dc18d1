"""Single-image / batch-16 latency of the SISR x4 forward (LR 64 x 64 -> 256 x 256): python tools/probes/sisr_latency.py [n ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_net  # noqa: E402
from virnet_amd.utils.synth import synth_images  # noqa: E402

dev = torch.device("cuda", 0)
net, sd = build_net(dev, "sisr")
net.load_state_dict(sd, strict=True)
net = net.to(dev).eval()
for n in [int(a) for a in sys.argv[1:]] or [1, 16]:
    x = synth_images(n, 3, 64, 64).to(dev)
    with torch.no_grad():
        for _ in range(5):
            net(x, 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            net(x, 4)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
    print(f"sisr x4 eager ({n}, 3, 64, 64) -> 256x256: {dt * 1e3:7.3f} ms / forward", flush=True)
    g = net.graphed(check=os.environ.get("GRAPH_CHECK", "sync"))
    with torch.no_grad():
        for _ in range(5):
            g(x, 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g(x, 4)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
    print(f"sisr x4 graph ({n}, 3, 64, 64) -> 256x256: {dt * 1e3:7.3f} ms / forward", flush=True)
