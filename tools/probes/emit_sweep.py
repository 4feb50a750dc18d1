"""Random batch / image sizes through the denoiser's training step: T emission on vs VIRNET_T_EMIT=0 under the pinned kernel form
(VIRNET_DETERMINISTIC=1) -- every weight gradient must be bitwise identical (the emitted image is the re-laid one), bias gradients to fp32
noise.  python tools/probes/emit_sweep.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from virnet_amd.networks import VIRAttResUNet  # noqa: E402
from virnet_amd.utils.synth import synth_images, synth_state_dict  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = np.random.Generator(np.random.Philox(key=[17, int(sys.argv[2]) if len(sys.argv) > 2 else 1]))
os.environ["VIRNET_DETERMINISTIC"] = "1"
bad = 0
for case in range(cases):
    form = "bf16" if g.integers(0, 3) == 0 else "wx4"
    os.environ["VIRNET_CONV_FORM"] = form
    feats = [[64, 96], [96, 192, 288], [32, 64, 96]][int(g.integers(0, 3))]
    nres = int(g.integers(1, 3))
    cfg = dict(im_chn=3, sigma_chn=1, n_feat=feats, dep_S=int(g.integers(3, 6)), n_resblocks=nres, noise_cond=True, extra_mode="Input")
    net = VIRAttResUNet(**cfg)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=5))
    net = net.cuda().train()
    m = 1 << (len(feats) - 1)
    n, h, w = int(g.integers(1, 9)), int(g.integers(6 * m, 150)), int(g.integers(6 * m, 170))
    x = synth_images(n, 3, h, w, seed=100 + case).cuda()
    gt = synth_images(n, 3, h, w, seed=200 + case).cuda()

    def step():
        for p in net.parameters():
            p.grad = None
        mu, sigma = net(x)
        (((mu - gt) ** 2).mean() * 50 + (sigma.log() ** 2).mean() * 0.01).backward()
        return {k: p.grad.clone() for k, p in net.named_parameters()}

    os.environ.pop("VIRNET_T_EMIT", None)
    ga = step()
    os.environ["VIRNET_T_EMIT"] = "0"
    gb = step()
    os.environ.pop("VIRNET_T_EMIT", None)
    thin = ("SNet.conv1.weight", "SNet.conv_last.weight", "RNet.head.weight", "RNet.tail.weight")
    worst_w, worst_b = 0.0, 0.0
    ok = True
    for k in ga:
        scale = max(float(gb[k].abs().max()), 1e-30)
        err = float((ga[k] - gb[k]).abs().max()) / scale
        if k.endswith(".weight") and not (form == "bf16" and k in thin):
            ok &= bool(torch.equal(ga[k], gb[k]))
            worst_w = max(worst_w, err)
        else:
            ok &= err <= (2e-2 if form == "bf16" else 5e-5)
            worst_b = max(worst_b, err)
    bad += 0 if ok else 1
    print(f"{case:3d} {form:4s} feats={feats} nres={nres} n={n} {h}x{w}  weights max rel diff {worst_w:.1e}  biases {worst_b:.1e}{'' if ok else '   <-- FAIL'}", flush=True)
print("failures", bad)
