"""Host-side counterpart of the reference's denoising evaluation loop, restated -- not copied.

Reference: scripts/denoising_virnet_syn.py:88-156 (iid cases), utils/util_denoising.py:65-67 (rng), utils/util_image.py:68-89
(PSNR), skimage's img_as_float32 / img_as_ubyte.  Only what the PSNR-parity line needs: image reading, the exact noise stream of
the iid cases and uint8 PSNR.  (The niid variance maps, SSIM and the SISR degradation are "next" rows.)
"""
from __future__ import annotations

import math
from typing import Iterable, List, Sequence, Tuple

import numpy as np

IID_SIGMAS = (15, 25, 50)          # scripts/denoising_virnet_syn.py:103-104
NOISE_SEED = 1000                  # utils/util_denoising.py:65


def imread_rgb_uint8(path: str) -> np.ndarray:
    """H x W x 3 uint8 RGB (lossless PNG/BMP/TIF decode gives the same pixels as the reference's cv2 reader)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def img_as_float32(im_u8: np.ndarray) -> np.ndarray:
    """uint8 -> float32 in [0,1] (skimage.img_as_float32: multiply by 1/255 in float32)."""
    return im_u8.astype(np.float32) * np.float32(1.0 / 255.0)


def img_as_ubyte(im: np.ndarray) -> np.ndarray:
    """float in [0,1] -> uint8, round half to even like skimage (np.rint of x*255)."""
    return np.rint(np.clip(im, 0.0, 1.0).astype(np.float64) * 255.0).astype(np.uint8)


def calculate_psnr(im1: np.ndarray, im2: np.ndarray, border: int = 0) -> float:
    """PSNR of two uint8 images in dB over the RGB channels (utils/util_image.py:68-89 with ycbcr=False)."""
    if im1.shape != im2.shape:
        raise ValueError("Input images must have the same dimensions.")
    h, w = im1.shape[:2]
    a = im1[border:h - border, border:w - border].astype(np.float64)
    b = im2[border:h - border, border:w - border].astype(np.float64)
    mse = float(np.mean((a - b) ** 2))
    return float("inf") if mse == 0 else 20.0 * math.log10(255.0 / math.sqrt(mse))


def iid_noise_stream(shapes: Sequence[Tuple[int, int]], sigmas: Iterable[int] = IID_SIGMAS, seed: int = NOISE_SEED):
    """Yield (sigma, image index, noise[h,w,3] float32) in the reference's order for the FIRST dataset of the script.

    The script shares ONE generator across all cases (scripts/denoising_virnet_syn.py:95,130): sigma=50 on CBSD68 is reached
    only after the sigma=15 and sigma=25 draws over all 68 images, so the stream is replayed from the image shapes."""
    rng = np.random.default_rng(seed=seed)
    for sigma in sigmas:
        for idx, (h, w) in enumerate(shapes):
            noise = rng.standard_normal(size=(h, w, 3)) * (np.ones([h, w], dtype=np.float32) * (sigma / 255.0))[:, :, np.newaxis]
            yield sigma, idx, noise.astype(np.float32)


def noisy_inputs(images: dict, shapes: Sequence[Tuple[int, int]], sigma: int) -> List[Tuple[int, np.ndarray, np.ndarray]]:
    """For the images given as {index in the sorted dataset: uint8 HWC}, the (index, gt, noisy float32 HWC) triples of case `sigma`.

    `im_noisy = img_as_float32(gt) + noise`, NOT clipped (scripts/denoising_virnet_syn.py:131)."""
    out = []
    for s, idx, noise in iid_noise_stream(shapes):
        if s == sigma and idx in images:
            gt = images[idx]
            if gt.shape[:2] != tuple(shapes[idx]):
                raise ValueError(f"image {idx} has shape {gt.shape[:2]}, the dataset list says {shapes[idx]}")
            out.append((idx, gt, img_as_float32(gt) + noise))
        if s > sigma:
            break
    return out
