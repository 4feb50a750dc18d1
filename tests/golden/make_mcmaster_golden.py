#!/usr/bin/env python
"""McMaster, the SECOND dataset scripts/denoising_virnet_syn.py walks (:93 `datasets = ['CBSD68', 'McMaster']`), through the REFERENCE
itself (build container only: imports /root/reference, torch CPU).

    python tests/golden/make_mcmaster_golden.py            -> tests/golden/mcmaster.json (+ copies the 18 .tif files it reads)

The script shares ONE noise generator across datasets and cases (:95,130): McMaster's iid sigma = 50 case is reached after the
3 x 68 CBSD68 draws and McMaster's own sigma = 15 / 25 draws, all replayed here with the reference's generator.  Network = the reference's
VIRAttResUNet (denoise-syn config) with the deterministic synthetic state_dict of the parity tests (checkpoints are not shipped).
The JSON holds the reference's per-image PSNRs; tests/test_fullset_gpu.py runs the HIP forward on the same inputs (<= 0.01 dB)."""
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

from networks.VIRNet import VIRAttResUNet  # noqa: E402  (the reference)
from utils import util_denoising, util_image  # noqa: E402  (the reference)
from PIL import Image  # noqa: E402
from virnet_amd.utils.synth import synth_state_dict  # noqa: E402

torch.set_num_threads(int(os.environ.get("FULLSET_THREADS", "8")))


def as_ubyte(a):
    return np.clip(np.rint(np.clip(a, -1.0, 1.0).astype(np.float64) * 255.0), 0, 255).astype(np.uint8)


cbsd = sorted(str(x) for x in glob.glob(os.path.join(REF, "test_data", "CBSD68", "*.png")))
files = sorted(str(x) for x in glob.glob(os.path.join(REF, "test_data", "McMaster", "*.tif")))
assert len(cbsd) == 68 and len(files) == 18
os.makedirs(os.path.join(HERE, "mcmaster"), exist_ok=True)
for f in files:
    dst = os.path.join(HERE, "mcmaster", os.path.basename(f))
    if not os.path.exists(dst):
        shutil.copyfile(f, dst)
cfg = dict(im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input", noise_avg=False)
net = VIRAttResUNet(**cfg)
net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}), strict=True)
rng = util_denoising.noise_generator()
cbsd_shapes = [Image.open(f).size[::-1] for f in cbsd]
for sigma in (15, 25, 50):                                   # the first dataset's draws (scripts/denoising_virnet_syn.py:110-130)
    for (h, w) in cbsd_shapes:
        rng.standard_normal(size=(h, w, 3))
ims = [np.asarray(Image.open(f).convert("RGB")) for f in files]
rows, t0 = [], time.time()
for sigma in (15, 25, 50):
    for idx, im in enumerate(ims):
        h, w = im.shape[:2]
        sig = np.ones([h, w], dtype=np.float32) * (sigma / 255.)
        noise = rng.standard_normal(size=(h, w, 3)) * sig[:, :, np.newaxis]
        if sigma != 50:
            continue
        noisy = (im.astype(np.float32) * np.float32(1.0 / 255.0) + noise).astype(np.float32)
        x = torch.from_numpy(noisy.transpose(2, 0, 1)[np.newaxis].copy())
        with torch.no_grad():
            mu, sigma_est = net(x)
        den = as_ubyte(mu.squeeze(0).numpy().transpose(1, 2, 0))
        probe = dict(first=[float(v) for v in noise.astype(np.float32).reshape(-1)[:4]], sum=float(noise.astype(np.float32).astype(np.float64).sum()))
        rows.append(dict(name=os.path.basename(files[idx]), index=idx, noise_probe=probe, psnr=float(util_image.calculate_psnr(den, im, border=0, ycbcr=False)),
                         mu_mean=float(mu.double().mean()), mu_absmax=float(mu.abs().max()), sigma_mean=float(sigma_est.double().mean())))
        print("mcmaster", idx, rows[-1]["name"], round(rows[-1]["psnr"], 4), f"{time.time() - t0:.0f}s", flush=True)
out = dict(config=cfg, names=[os.path.basename(f) for f in files], shapes=[list(im.shape[:2]) for im in ims],
           cbsd68_shapes=[list(s) for s in cbsd_shapes], images=rows, mean_psnr=float(np.mean([r["psnr"] for r in rows])))
json.dump(out, open(os.path.join(HERE, "mcmaster.json"), "w"), indent=1)
print("wrote mcmaster.json", f"{time.time() - t0:.0f}s")
