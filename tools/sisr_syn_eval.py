#!/usr/bin/env python
"""Counterpart of the reference's scripts/sisr_virnet_syn.py (BASELINE configs[3]) on the MI355X path: PSNR-Y / SSIM-Y of the SISR
model per dataset and per synthetic blur kernel (seven anisotropic Gaussians), noise level 2.55 by default.

Restated, not copied (virnet_amd/sisr_eval.py: kernels, blur, antialiased bicubic downscale, seeded noise -- pinned to the reference's
helpers by tests/golden/sisr_harness.npz).  LPIPS is not reported (it needs the `lpips` package's pretrained AlexNet).  Without
--ckpt_path the deterministic synthetic weights are used: a plumbing check, not restoration quality.  Needs a ROCm device.

    python tools/sisr_syn_eval.py --sf 4 --data tests/golden/set5:bmp [--ckpt_path model_zoo/virnet_sisr_x4.pth] [--nlevel 2.55]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virnet_amd import sisr_eval  # noqa: E402

# scripts/sisr_virnet_syn.py:53-63
CFG = dict(im_chn=3, sigma_chn=1, kernel_chn=3, n_feat=[96, 160, 224], dep_S=5, dep_K=8, noise_cond=True, kernel_cond=True,
           n_resblocks=2, extra_mode="Both", noise_avg=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt_path", default="")
    ap.add_argument("--sf", type=int, default=4)
    ap.add_argument("--nlevel", type=float, default=2.55)
    ap.add_argument("--data", nargs="+", default=["test_data/Set14:bmp", "test_data/CBSD68:png"], help="folder:extension, in script order")
    args = ap.parse_args()
    from virnet_amd.networks import VIRAttResUNetSR
    net = VIRAttResUNetSR(**CFG)
    if args.ckpt_path:
        sd = torch.load(args.ckpt_path, map_location="cpu")
        sd = sd.get("model_state_dict", sd)
        sd = {(k[7:] if k.startswith("module.") else k): v.float() for k, v in sd.items()}
    else:
        from virnet_amd.utils.synth import synth_state_dict
        print("no --ckpt_path: deterministic synthetic weights (the numbers below are a plumbing check, not restoration quality)")
        sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()

    def forward(lr_hwc, sf):
        x = torch.from_numpy(np.ascontiguousarray(lr_hwc.transpose(2, 0, 1)[np.newaxis]))
        with torch.no_grad():
            return net(x.cuda(), sf)[0].squeeze(0).cpu().numpy().transpose(1, 2, 0)

    rows = sisr_eval.sisr_table(forward, args.data, args.sf, nlevel=args.nlevel)
    if not rows:
        print("no images found under", args.data)
    for r in rows:
        print(f"Dataset: {r['dataset']:>8s}, Kernel: {r['kernel']:d}, PSNRY: {r['psnr_y']:5.2f}, SSIMY: {r['ssim_y']:6.4f}", flush=True)


if __name__ == "__main__":
    main()
