#!/usr/bin/env python
"""Golden values for the evaluation-harness counterpart (virnet_amd/eval.py), produced by the REFERENCE's own helpers.

Build container only (imports /root/reference with cv2 / skimage / lpips / thop stubbed -- only pure-numpy helpers are called).
Writes tests/golden/harness.json: the CBSD68 shape list in the script's sorted order, the first noise values of the sigma=50
case for three fixture images (copied data files, tests/golden/cbsd68/*.png), and calculate_psnr on seeded arrays.
"""
import glob
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("VIRNET_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
for name in ("cv2", "thop", "lpips", "skimage", "skimage.metrics", "skimage.color"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["skimage"].img_as_ubyte = None
sys.modules["skimage"].img_as_float32 = None
sys.modules["skimage"].img_as_float64 = None
sys.modules["skimage.metrics"].structural_similarity = None
sys.modules["thop"].profile = None

from utils import util_denoising, util_image  # noqa: E402  (the reference)
from PIL import Image  # noqa: E402

files = sorted(str(x) for x in glob.glob(os.path.join(REF, "test_data", "CBSD68", "*.png")))   # script: sorted str paths
shapes = [tuple(reversed(Image.open(f).size)) for f in files]
names = [os.path.basename(f) for f in files]
fixtures = sorted(os.path.basename(f) for f in glob.glob(os.path.join(HERE, "cbsd68", "*.png")))
want = {names.index(n): n for n in fixtures}

rng = util_denoising.noise_generator()
noise_head = {}
for sigma in (15, 25, 50):
    for idx, (h, w) in enumerate(shapes):
        sig = np.ones([h, w], dtype=np.float32) * (sigma / 255.)
        noise = rng.standard_normal(size=(h, w, 3)) * sig[:, :, np.newaxis]     # scripts/denoising_virnet_syn.py:128-130
        if sigma == 50 and idx in want:
            noise_head[want[idx]] = dict(index=idx, first8=noise.astype(np.float32).reshape(-1)[:8].tolist(),
                                         sum=float(noise.astype(np.float32).astype(np.float64).sum()))

g = np.random.default_rng(7)
a = g.integers(0, 256, size=(37, 41, 3), dtype=np.uint8)
b = np.clip(a.astype(np.int32) + g.integers(-9, 10, size=a.shape), 0, 255).astype(np.uint8)
psnr = dict(seed=7, shape=[37, 41, 3], border0=util_image.calculate_psnr(a, b, border=0, ycbcr=False),
            border4=util_image.calculate_psnr(a, b, border=4, ycbcr=False))
# niid variance maps (utils/util_denoising.py:69-124) and the Y conversion (utils/util_image.py:129-153)
rng2 = util_denoising.noise_generator()
maps = [util_denoising.peaks(256), util_denoising.sincos_kernel(), util_denoising.generate_gauss_kernel_mix(256, 256, rng2)]
after = rng2.standard_normal(size=4).tolist()          # the stream position after the mixture map consumed its draws
niid = dict(stats=[dict(min=float(m.min()), max=float(m.max()), sum=float(np.asarray(m, dtype=np.float64).sum()),
                        probe=[float(m[17, 200]), float(m[255, 0]), float(m[128, 64])]) for m in maps], next_normals=after)
ycase = util_image.rgb2ycbcr(a, True)
ychk = dict(sum=int(ycase.astype(np.int64).sum()), first=ycase.reshape(-1)[:6].tolist(),
            psnr_y=util_image.calculate_psnr(a, b, border=4, ycbcr=True))
json.dump(dict(niid=niid, ycbcr=ychk, cbsd68_names=names, cbsd68_shapes=shapes, noise_sigma50=noise_head, psnr=psnr),
          open(os.path.join(HERE, "harness.json"), "w"), indent=1)
print("fixtures", want, "psnr", psnr)
