#!/usr/bin/env python
"""Counterpart of the reference's scripts/testing_demo.py on the MI355X path (same flags: --task --ckpt_path -i -o --prefix --sf).

Restated, not copied: builds the three script configurations (scripts/testing_demo.py:21-66) from virnet_amd.networks, loads the
checkpoint's ['model_state_dict'] (stripping a DDP 'module.' prefix), runs one image at a time under no_grad, clamps to [0,1]
and writes PNGs.  Without --ckpt_path it runs on the deterministic synthetic weights (there is no network access to the release
checkpoints in the build environment), which exercises the path but does not restore images.
"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virnet_amd import eval as veval  # noqa: E402
from virnet_amd.networks import VIRAttResUNet, VIRAttResUNetSR  # noqa: E402
from virnet_amd.utils.synth import synth_state_dict  # noqa: E402

TASKS = {
    "denoising-syn": (VIRAttResUNet, dict(im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True,
                                          extra_mode="Input", noise_avg=False)),
    "denoising-real": (VIRAttResUNet, dict(im_chn=3, sigma_chn=3, n_feat=[96, 160, 224, 288], dep_S=8, n_resblocks=3,
                                           noise_cond=True, extra_mode="Input", noise_avg=False)),
    "sisr": (VIRAttResUNetSR, dict(im_chn=3, sigma_chn=1, dep_S=5, dep_K=8, n_feat=[96, 160, 224], n_resblocks=2, extra_mode="Both",
                                   noise_avg=True, noise_cond=True, kernel_cond=True)),
}


def load_model(task, ckpt_path):
    cls, kw = TASKS[task]
    net = cls(**kw)
    if ckpt_path:
        sd = torch.load(ckpt_path, map_location="cpu")["model_state_dict"]
        if all(k.startswith("module.") for k in sd):
            sd = {k[7:]: v for k, v in sd.items()}
    else:
        print("no --ckpt_path: using deterministic synthetic weights (outputs are not restorations)")
        sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict(sd, strict=True)
    return net.cuda().eval()


def process_image(net, im_lq, task, sf):
    """im_lq: h x w x c float32 in [0,1] -> restored h' x w' x c in [0,1] (scripts/testing_demo.py:77-97)."""
    if im_lq.ndim == 2:
        im_lq = np.stack([im_lq] * 3, axis=2)
    x = torch.from_numpy(np.ascontiguousarray(im_lq.transpose(2, 0, 1)[np.newaxis])).float().cuda()
    with torch.no_grad():
        mu = net(x, sf)[0] if task == "sisr" else net(x)[0]
        mu.clamp_(0.0, 1.0)
    return mu.squeeze(0).cpu().numpy().transpose(1, 2, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt_path", default="", type=str)
    ap.add_argument("-i", "--in_path", type=str, required=True, help="input image or folder")
    ap.add_argument("-o", "--out_path", type=str, default="outputs")
    ap.add_argument("--task", type=str, default="denoising-syn", choices=sorted(TASKS))
    ap.add_argument("--prefix", type=str, default="")
    ap.add_argument("--sf", default=4, type=int)
    args = ap.parse_args()
    net = load_model(args.task, args.ckpt_path)
    out = Path(args.out_path)
    out.mkdir(parents=True, exist_ok=True)
    src = Path(args.in_path)
    paths = sorted(p for p in src.iterdir() if p.suffix.lower() in (".png", ".jpg", ".jpeg", ".bmp", ".tif")) if src.is_dir() else [src]
    from PIL import Image
    for p in paths:
        im = veval.img_as_float32(veval.imread_rgb_uint8(str(p)))
        pred = process_image(net, im, args.task, args.sf)
        name = f"{p.stem}_{args.prefix}.png" if args.prefix else f"{p.stem}.png"
        Image.fromarray(veval.img_as_ubyte(pred)).save(out / name)
    print(f"Please enjoy the result in {args.out_path}!")


if __name__ == "__main__":
    main()
