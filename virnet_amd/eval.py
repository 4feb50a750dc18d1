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


def iid_noise_stream(shapes: Sequence[Tuple[int, int]], sigmas: Iterable[int] = IID_SIGMAS, seed: int = NOISE_SEED,
                     before: Sequence[Sequence[Tuple[int, int]]] = ()):
    """Yield (sigma, image index, noise[h,w,3] float32) in the reference's order for one dataset of the script.

    The script shares ONE generator across all cases (scripts/denoising_virnet_syn.py:95,130): sigma=50 on CBSD68 is reached
    only after the sigma=15 and sigma=25 draws over all 68 images, so the stream is replayed from the image shapes.  ``before``: the
    shape lists of the datasets the script walks EARLIER (McMaster comes after all three CBSD68 cases, scripts/denoising_virnet_syn.py:93,
    110): their draws are replayed and discarded."""
    sigmas = tuple(sigmas)
    rng = np.random.default_rng(seed=seed)
    for earlier in before:
        for _ in sigmas:
            for (h, w) in earlier:
                rng.standard_normal(size=(h, w, 3))
    for sigma in sigmas:
        for idx, (h, w) in enumerate(shapes):
            noise = rng.standard_normal(size=(h, w, 3)) * (np.ones([h, w], dtype=np.float32) * (sigma / 255.0))[:, :, np.newaxis]
            yield sigma, idx, noise.astype(np.float32)


def noisy_inputs(images: dict, shapes: Sequence[Tuple[int, int]], sigma: int,
                 before: Sequence[Sequence[Tuple[int, int]]] = ()) -> List[Tuple[int, np.ndarray, np.ndarray]]:
    """For the images given as {index in the sorted dataset: uint8 HWC}, the (index, gt, noisy float32 HWC) triples of case `sigma`.

    `im_noisy = img_as_float32(gt) + noise`, NOT clipped (scripts/denoising_virnet_syn.py:131).  ``before``: see iid_noise_stream."""
    out = []
    for s, idx, noise in iid_noise_stream(shapes, before=before):
        if s == sigma and idx in images:
            gt = images[idx]
            if gt.shape[:2] != tuple(shapes[idx]):
                raise ValueError(f"image {idx} has shape {gt.shape[:2]}, the dataset list says {shapes[idx]}")
            out.append((idx, gt, img_as_float32(gt) + noise))
        if s > sigma:
            break
    return out


# ---------------------------------------------------------------------------------------------------------------------
# niid cases (scripts/denoising_virnet_syn.py:98-102,118-127): three variance maps rescaled to [10,75]/255 and resized to the image
# ---------------------------------------------------------------------------------------------------------------------
SIGMA_MIN, SIGMA_MAX = 10 / 255.0, 75 / 255.0      # scripts/denoising_virnet_syn.py:97-98


def peaks(n: int) -> np.ndarray:
    """MATLAB's peaks surface on [-3,3]^2 (utils/util_denoising.py:69-78)."""
    x = np.linspace(-3, 3, n)
    xx, yy = np.meshgrid(x, x)
    return (3 * (1 - xx) ** 2 * np.exp(-xx ** 2 - (yy + 1) ** 2) - 10 * (xx / 5.0 - xx ** 3 - yy ** 5) * np.exp(-xx ** 2 - yy ** 2)
            - 1 / 3.0 * np.exp(-(xx + 1) ** 2 - yy ** 2))


def sincos_kernel() -> np.ndarray:
    """sin(x)+cos(y) on [1,10]x[1,20], 256x256 (utils/util_denoising.py:120-124)."""
    xx, yy = np.meshgrid(np.linspace(1, 10, 256), np.linspace(1, 20, 256))
    return np.sin(xx) + np.cos(yy)


def gauss_kernel_mix(h: int, w: int, rng: np.random.Generator) -> np.ndarray:
    """Mixture of one Gaussian bump per 32x32 patch (utils/util_denoising.py:80-118); draw order: x-centres, y-centres, scales."""
    pch = 32
    kh, kw = h // pch, w // pch
    k = kh * kw
    cw = (rng.uniform(low=0, high=pch, size=(kh, kw)) + (np.arange(kw) * pch).reshape(1, -1)).reshape(1, 1, k).astype(np.float32)
    chh = (rng.uniform(low=0, high=pch, size=(kh, kw)) + (np.arange(kh) * pch).reshape(-1, 1)).reshape(1, 1, k).astype(np.float32)
    scale = rng.uniform(low=pch / 2, high=pch, size=(1, 1, k)).astype(np.float32)
    xx, yy = np.meshgrid(np.arange(0, w), np.arange(0, h))
    xx, yy = xx[:, :, None].astype(np.float32), yy[:, :, None].astype(np.float32)
    zz = 1.0 / (2 * np.pi * scale ** 2) * np.exp((-(xx - cw) ** 2 - (yy - chh) ** 2) / (2 * scale ** 2))
    return zz.sum(axis=2) / k


def resize_nearest_exact(a: np.ndarray, h: int, w: int) -> np.ndarray:
    """cv2.resize(..., INTER_NEAREST_EXACT): source index floor((dst + 0.5) * in/out)  (restated from its definition; cv2 is not
    available in the build container, so this one is not pinned against the reference)."""
    ih, iw = a.shape[:2]
    ys = np.minimum(np.floor((np.arange(h) + 0.5) * (ih / h)).astype(np.int64), ih - 1)
    xs = np.minimum(np.floor((np.arange(w) + 0.5) * (iw / w)).astype(np.int64), iw - 1)
    return a[ys][:, xs]


def niid_sigma_maps(rng: np.random.Generator) -> List[np.ndarray]:
    """The three 256x256 sigma maps of the niid cases, rescaled to [10/255, 75/255] (scripts/denoising_virnet_syn.py:99-102,118).
    `rng` must be the fresh evaluation generator: the mixture map consumes its first draws."""
    maps = [peaks(256), sincos_kernel(), gauss_kernel_mix(256, 256, rng)]
    return [SIGMA_MIN + (m - m.min()) / (m.max() - m.min()) * (SIGMA_MAX - SIGMA_MIN) for m in maps]


# ---------------------------------------------------------------------------------------------------------------------
# Y channel and SSIM (utils/util_image.py:17-66,129-153)
# ---------------------------------------------------------------------------------------------------------------------
def rgb2y_uint8(im: np.ndarray) -> np.ndarray:
    """MATLAB-style luma of a uint8 RGB image, rounded back to uint8 (utils/util_image.py:129-153 with only_y=True)."""
    y = np.dot(im.astype(np.float64), np.array([65.481, 128.553, 24.966]) / 255.0) + 16.0
    return y.round().astype(np.uint8)


def _gauss_window(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    g = np.exp(-((np.arange(size) - (size - 1) / 2.0) ** 2) / (2 * sigma ** 2))
    g /= g.sum()                                   # cv2.getGaussianKernel(11, 1.5)
    return np.outer(g, g)


def _filter_valid(a: np.ndarray, win: np.ndarray) -> np.ndarray:
    from numpy.lib.stride_tricks import sliding_window_view
    return np.einsum("ijkl,kl->ij", sliding_window_view(a, win.shape), win)   # == cv2.filter2D(...)[5:-5, 5:-5]


def ssim_channel(a: np.ndarray, b: np.ndarray) -> float:
    """SSIM of two single-channel [0,255] images: 11x11 Gaussian (sigma 1.5), valid region only (utils/util_image.py:17-37)."""
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b, win = a.astype(np.float64), b.astype(np.float64), _gauss_window()
    mu1, mu2 = _filter_valid(a, win), _filter_valid(b, win)
    s1 = _filter_valid(a * a, win) - mu1 ** 2
    s2 = _filter_valid(b * b, win) - mu2 ** 2
    s12 = _filter_valid(a * b, win) - mu1 * mu2
    return float((((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s1 + s2 + c2))).mean())


def calculate_ssim(im1: np.ndarray, im2: np.ndarray, border: int = 0, ycbcr: bool = False) -> float:
    """Mean SSIM over the RGB channels, or on Y (utils/util_image.py:39-66)."""
    if im1.shape != im2.shape:
        raise ValueError("Input images must have the same dimensions.")
    if ycbcr:
        im1, im2 = rgb2y_uint8(im1), rgb2y_uint8(im2)
    h, w = im1.shape[:2]
    im1, im2 = im1[border:h - border, border:w - border], im2[border:h - border, border:w - border]
    if im1.ndim == 2:
        return ssim_channel(im1, im2)
    return float(np.mean([ssim_channel(im1[:, :, i], im2[:, :, i]) for i in range(im1.shape[2])]))


def calculate_psnr_y(im1: np.ndarray, im2: np.ndarray, border: int = 0) -> float:
    """PSNR on the Y channel (utils/util_image.py:68-89 with ycbcr=True; the SISR scripts use border = sf**2)."""
    return calculate_psnr(rgb2y_uint8(im1), rgb2y_uint8(im2), border)


def denoise_table(forward, data: Sequence[str], noise_type: str = "niid", rng: "np.random.Generator | None" = None,
                  with_ssim: bool = True) -> List[dict]:
    """The PSNR / SSIM table of scripts/denoising_virnet_syn.py:93-156 for any ``forward(noisy float32 HWC) -> mu float32 HWC``.

    ``data`` = ["folder:ext", ...] in the script's order.  ONE Generator (seed 1000) is shared by every dataset and case (the niid
    mixture maps consume its first draws), noisy = img_as_float32(uint8) + float32(noise) unclipped, output = img_as_ubyte(clip(mu)).
    Returns one row per (dataset, case): {"dataset", "case", "psnr", "ssim", "images", "per_image_psnr"}."""
    import glob
    import os
    if noise_type not in ("iid", "niid"):
        raise ValueError(f"noise_type {noise_type!r}: expected iid or niid")
    rng = np.random.default_rng(seed=NOISE_SEED) if rng is None else rng
    cases = niid_sigma_maps(rng) if noise_type == "niid" else list(IID_SIGMAS)
    rows = []
    for spec in data:
        folder, ext = spec.rsplit(":", 1)
        files = sorted(glob.glob(os.path.join(folder, "*." + ext)))
        if not files:
            continue
        for jj, case in enumerate(cases):
            psnrs, ssims = [], []
            for f in files:
                gt = imread_rgb_uint8(f)
                h, w = gt.shape[:2]
                sigma = (resize_nearest_exact(case, h, w).astype(np.float32) if noise_type == "niid"
                         else np.ones([h, w], dtype=np.float32) * (case / 255.0))
                noise = rng.standard_normal(size=gt.shape) * sigma[:, :, np.newaxis]
                noisy = img_as_float32(gt) + noise.astype(np.float32)
                den = img_as_ubyte(np.clip(forward(noisy), 0.0, 1.0))
                psnrs.append(calculate_psnr(den, gt, border=0))
                if with_ssim:
                    ssims.append(calculate_ssim(den, gt, border=0))
            rows.append({"dataset": os.path.basename(folder.rstrip("/")), "case": (jj + 1) if noise_type == "niid" else int(case),
                         "psnr": float(np.mean(psnrs)), "ssim": float(np.mean(ssims)) if ssims else float("nan"),
                         "images": len(files), "per_image_psnr": psnrs})
    return rows


# ---- 8-way flip/rotation self-ensemble (scripts/denoising_virnet_real_sidd.py:120-136, utils/util_image.py:391-436) ------------------
def dihedral(im: np.ndarray, mode: int) -> np.ndarray:
    """Element ``mode`` of the square's symmetry group on an [h,w,c] image: k = mode//2 quarter turns counter-clockwise, then an
    up-down flip for odd modes (the reference's enumeration: 0 identity, 1 flip, 2 rot90, 3 rot90+flip, ...)."""
    if not 0 <= mode < 8:
        raise ValueError(f"mode {mode}: expected 0..7")
    out = np.rot90(im, k=mode // 2)
    return np.ascontiguousarray(np.flipud(out) if mode & 1 else out)


def dihedral_inverse(im: np.ndarray, mode: int) -> np.ndarray:
    """Undo :func:`dihedral`: flip back first, then turn back."""
    if not 0 <= mode < 8:
        raise ValueError(f"mode {mode}: expected 0..7")
    out = np.flipud(im) if mode & 1 else im
    return np.ascontiguousarray(np.rot90(out, k=-(mode // 2)))


def flip_ensemble(forward, noisy: np.ndarray) -> np.ndarray:
    """Mean of the eight back-transformed restorations of the eight transformed inputs (``--flip`` of the SIDD / DND scripts)."""
    acc = np.zeros(noisy.shape, dtype=np.float32)
    for mode in range(8):
        acc += dihedral_inverse(np.asarray(forward(dihedral(noisy, mode)), dtype=np.float32), mode)
    return acc / 8
