#!/usr/bin/env python
"""Counterpart of the reference's scripts/denoising_virnet_syn.py (BASELINE configs[0]) on the MI355X path: the PSNR / SSIM table of the
denoise-syn model over image folders, iid (sigma 15/25/50) or niid (three variance maps) noise.

Restated, not copied.  What makes the table comparable with the reference's (scripts/denoising_virnet_syn.py:93-156):
  * ONE numpy Generator (seed 1000) shared by every dataset and case, in the script's order -- datasets in the order given, cases in
    order, images sorted by path; the niid mixture map consumes the first draws;
  * noisy = img_as_float32(uint8 image) + float32(noise), NOT clipped; output = img_as_ubyte(clip(mu, 0, 1));
  * PSNR / SSIM on uint8 RGB, border 0 (virnet_amd/eval.py, pinned against the reference's helpers in tests/test_eval_harness.py).
Without --ckpt_path the deterministic synthetic weights are used (no checkpoint can be downloaded in the build environment): the table
is then only a plumbing check.  Needs a ROCm device: the product path has no CPU fallback.

    python tools/denoising_syn_eval.py --data tests/golden/cbsd68:png --noise_type iid [--ckpt_path model_state_niidgauss.pt]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virnet_amd import eval as veval  # noqa: E402

CFG = dict(n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input", noise_avg=False)   # :62-71


def load_state(ckpt_path, shapes):
    if ckpt_path:
        sd = torch.load(ckpt_path, map_location="cpu")
        sd = sd.get("model_state_dict", sd)
        return {(k[7:] if k.startswith("module.") else k): v.float() for k, v in sd.items()}
    from virnet_amd.utils.synth import synth_state_dict
    print("no --ckpt_path: deterministic synthetic weights (the numbers below are a plumbing check, not restoration quality)")
    return synth_state_dict(shapes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt_path", default="")
    ap.add_argument("--data", nargs="+", default=["test_data/CBSD68:png", "test_data/McMaster:tif"], help="folder:extension, in script order")
    ap.add_argument("--noise_type", default="niid", choices=["iid", "niid"])
    args = ap.parse_args()

    from virnet_amd.networks import VIRAttResUNet
    net = VIRAttResUNet(im_chn=3, sigma_chn=1, **CFG)
    sd = load_state(args.ckpt_path, {k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()

    def forward(noisy_hwc):
        x = torch.from_numpy(np.ascontiguousarray(noisy_hwc.transpose(2, 0, 1)[np.newaxis]))
        with torch.no_grad():
            return net(x.cuda())[0].squeeze(0).cpu().numpy().transpose(1, 2, 0)

    rows = veval.denoise_table(forward, args.data, args.noise_type)
    if not rows:
        print("no images found under", args.data)
    for r in rows:
        tag = f"case: {r['case']:d}" if args.noise_type == "niid" else f"sigma: {r['case']:d}"
        print(f"Dataset: {r['dataset']:8s}, {tag}, PSNR: {r['psnr']:5.2f}, SSIM: {r['ssim']:6.4f}", flush=True)


if __name__ == "__main__":
    main()
