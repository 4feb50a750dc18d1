#!/usr/bin/env python
"""Full evaluation sets through the REFERENCE itself (build container only: imports /root/reference, torch CPU).

    python tests/golden/make_fullset_golden.py            -> tests/golden/fullset.json (+ copies the data files it reads)

What BASELINE.json's metric names -- "PSNR parity CBSD68 sigma=50 / Set5 x4" -- on the FULL sets the reference's scripts walk:
  * scripts/denoising_virnet_syn.py:96-135: all 68 CBSD68 images, iid sigma = 50 reached after replaying the sigma = 15 / 25 draws of
    the shared generator, VIRAttResUNet (denoise-syn config) forward, img_as_ubyte, calculate_psnr;
  * scripts/sisr_virnet_syn.py:85-170: the 5 Set5 images x all 7 test kernels, x4, nlevel 2.55, bicubic degradation,
    VIRAttResUNetSR forward, clamp, img_as_ubyte, Y-PSNR with border sf^2.
The networks are the reference's own classes (networks/VIRNet.py) with the deterministic synthetic state_dict the parity tests use
(checkpoints are not shipped).  The JSON holds the reference's per-image PSNRs (and a few output statistics); the GPU
tests run the HIP forward on the same inputs and must land within 0.01 dB per image and in the set means
(tests/test_fullset_gpu.py).  The images are copied into tests/golden/{cbsd68,cbsd68_rest,set5}/ (data files the reference's scripts read; cbsd68/ holds the 12 images
the oracle-based tests walk, cbsd68_rest/ the other 56)."""
import glob
import json
import os
import shutil
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VIRNET_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
for name in ("cv2", "thop", "lpips", "skimage", "skimage.metrics", "skimage.color"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["skimage"].img_as_ubyte = None
sys.modules["skimage"].img_as_float32 = None
sys.modules["skimage"].img_as_float64 = None
sys.modules["skimage.metrics"].structural_similarity = None
sys.modules["thop"].profile = None

from networks.VIRNet import VIRAttResUNet, VIRAttResUNetSR  # noqa: E402  (the reference)
from utils import util_denoising, util_image, util_sisr  # noqa: E402  (the reference)
from PIL import Image  # noqa: E402
from virnet_amd.utils.synth import synth_state_dict  # noqa: E402  (deterministic weights: data, shared with the tests)

torch.set_num_threads(int(os.environ.get("FULLSET_THREADS", "8")))


def as_ubyte(a):            # skimage.img_as_ubyte on float images in [-1, 1]: clip to [0, 1] happens before in the scripts' use; round half to even
    return np.clip(np.rint(np.clip(a, -1.0, 1.0).astype(np.float64) * 255.0), 0, 255).astype(np.uint8)


out = {}
# ------------------------------------------------------------------ CBSD68, iid sigma = 50
files = sorted(str(x) for x in glob.glob(os.path.join(REF, "test_data", "CBSD68", "*.png")))
os.makedirs(os.path.join(HERE, "cbsd68_rest"), exist_ok=True)      # (cbsd68/ keeps the 12 images the oracle-based tests walk)
for f in files:
    if not os.path.exists(os.path.join(HERE, "cbsd68", os.path.basename(f))):
        dst = os.path.join(HERE, "cbsd68_rest", os.path.basename(f))
        if not os.path.exists(dst):
            shutil.copyfile(f, dst)
cfg = dict(im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input", noise_avg=False)
net = VIRAttResUNet(**cfg)
sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
net.load_state_dict(sd, strict=True)
rng = util_denoising.noise_generator()
ims = [np.asarray(Image.open(f).convert("RGB")) for f in files]
rows = []
t0 = time.time()
for sigma in (15, 25, 50):
    for idx, im in enumerate(ims):
        h, w = im.shape[:2]
        sig = np.ones([h, w], dtype=np.float32) * (sigma / 255.)
        noise = rng.standard_normal(size=(h, w, 3)) * sig[:, :, np.newaxis]                 # scripts/denoising_virnet_syn.py:128-130
        if sigma != 50:
            continue
        gt = im.astype(np.float32) * np.float32(1.0 / 255.0)
        noisy = (gt + noise).astype(np.float32)
        x = torch.from_numpy(noisy.transpose(2, 0, 1)[np.newaxis].copy())
        with torch.no_grad():
            mu, sigma_est = net(x)
        den = as_ubyte(mu.squeeze(0).numpy().transpose(1, 2, 0))
        rows.append(dict(name=os.path.basename(files[idx]), index=idx, psnr=float(util_image.calculate_psnr(den, im, border=0, ycbcr=False)),
                         mu_mean=float(mu.double().mean()), mu_absmax=float(mu.abs().max()), sigma_mean=float(sigma_est.double().mean())))
        print("cbsd68", idx, rows[-1]["name"], round(rows[-1]["psnr"], 4), f"{time.time() - t0:.0f}s", flush=True)
out["cbsd68_sigma50"] = dict(config=cfg, seed_note="synth_state_dict default seed", images=rows, mean_psnr=float(np.mean([r["psnr"] for r in rows])))

# ------------------------------------------------------------------ Set5 x4, 7 kernels
sfiles = sorted(glob.glob(os.path.join(REF, "test_data", "Set5", "*.bmp")))
os.makedirs(os.path.join(HERE, "set5"), exist_ok=True)
for f in sfiles:
    dst = os.path.join(HERE, "set5", os.path.basename(f))
    if not os.path.exists(dst):
        shutil.copyfile(f, dst)
scfg = dict(im_chn=3, sigma_chn=1, kernel_chn=3, n_feat=[96, 160, 224], dep_S=5, dep_K=8, noise_cond=True, kernel_cond=True,
            n_resblocks=2, extra_mode="Both", noise_avg=True)                               # scripts/sisr_virnet_syn.py:53-63
snet = VIRAttResUNetSR(**scfg)
ssd = synth_state_dict({k: tuple(v.shape) for k, v in snet.state_dict().items()}, seed=5)
snet.load_state_dict(ssd, strict=True)
snet.eval()
sf, p = 4, 21
SPEC = [(0.40, 0.40, 0.0), (0.60, 0.60, 0.0), (0.80, 0.80, 0.0), (0.4, 0.2, 0.0), (0.6, 0.3, 0.75 * np.pi), (0.8, 0.4, 0.25 * np.pi),
        (0.8, 0.4, 0.50 * np.pi)]                                                           # scripts/sisr_virnet_syn.py:103-116
srows = []
for kidx, (a, b, th) in enumerate(SPEC):
    kernel = util_sisr.shifted_anisotropic_Gaussian(p, sf, (a * sf) ** 2, (b * sf) ** 2, th, False)[0]
    for f in sfiles:
        im_gt = util_sisr.modcrop(np.asarray(Image.open(f).convert("RGB")), sf)
        im_lr = util_sisr.degrade_virnet(im_gt.astype(np.float32) * np.float32(1.0 / 255.0), kernel=kernel, sf=sf, nlevel=2.55, qf=None,
                                         downsampler="Bicubic")
        x = torch.from_numpy(im_lr.transpose((2, 0, 1))[np.newaxis, ]).type(torch.float32)
        with torch.no_grad():
            mu, kinfo, sig = snet(x, sf)
        im_sr = as_ubyte(mu.clamp(0.0, 1.0).squeeze(0).numpy().transpose((1, 2, 0)))
        srows.append(dict(name=os.path.basename(f), kernel=kidx, psnr_y=float(util_image.calculate_psnr(im_sr, im_gt, sf ** 2, True)),
                          kinfo=[float(v) for v in kinfo.reshape(-1)], sigma=float(sig.reshape(-1)[0]), lr_sum=float(np.asarray(im_lr, np.float64).sum())))
        print("set5", kidx, srows[-1]["name"], round(srows[-1]["psnr_y"], 4), f"{time.time() - t0:.0f}s", flush=True)
out["set5_x4"] = dict(config=scfg, seed=5, sf=sf, nlevel=2.55, images=srows,
                      mean_psnr_y_per_kernel=[float(np.mean([r["psnr_y"] for r in srows if r["kernel"] == k])) for k in range(7)])
json.dump(out, open(os.path.join(HERE, "fullset.json"), "w"), indent=1)
print("wrote fullset.json", f"{time.time() - t0:.0f}s")
