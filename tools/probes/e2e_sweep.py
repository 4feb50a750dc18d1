"""Random image / batch sizes through both networks: the default forms (Winograd-along-x where the launch rule says so, direct split-fp16
elsewhere) against the fp32 kernels (VIRNET_CONV_FORM=wino).  python tools/probes/e2e_sweep.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_net  # noqa: E402
from virnet_amd.utils.synth import synth_images  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = np.random.Generator(np.random.Philox(key=[91, int(sys.argv[2]) if len(sys.argv) > 2 else 1]))
dev = torch.device("cuda", 0)
nets = {}
for kind in ("denoise", "sisr"):
    net, sd = build_net(dev, kind)
    net.load_state_dict(sd, strict=True)
    nets[kind] = net.to(dev).eval()
worst = 0.0
for case in range(cases):
    kind = "sisr" if g.integers(0, 3) == 0 else "denoise"
    if kind == "sisr":
        n, h, w = int(g.integers(1, 20)), int(g.integers(8, 80)), int(g.integers(8, 100))
    else:
        n, h, w = int(g.integers(1, 40)), int(g.integers(8, 300)), int(g.integers(8, 300))
        while n * h * w > 40 * 128 * 128:
            n = max(1, n // 2)
    x = synth_images(n, 3, h, w, seed=1000 + case).to(dev)
    fwd = (lambda t: nets[kind](t, 4)) if kind == "sisr" else nets[kind]
    with torch.no_grad():
        os.environ.pop("VIRNET_CONV_FORM", None)
        out = fwd(x)
        os.environ["VIRNET_CONV_FORM"] = "wino"
        ref = fwd(x)
        os.environ.pop("VIRNET_CONV_FORM", None)
    err = max(float((a - b).abs().max()) for a, b in zip(out, ref))
    worst = max(worst, err)
    flag = "" if err <= 1e-4 and all(bool(torch.isfinite(a).all()) for a in out) else "   <-- FAIL"
    print(f"{case:3d} {kind:7s} n={n:2d} {h:3d}x{w:3d}  max|default - fp32| = {err:.2e}{flag}", flush=True)
print("worst", worst)
sys.exit(0 if worst <= 1e-4 else 1)
