"""SISR evaluation harness around the hot path (SURVEY.md 8-a11 iii / 8-f2): the synthetic degradation the reference's
``scripts/sisr_virnet_syn.py:99-137`` applies to a ground-truth image before calling ``net(lr, sf)``.

Own restatements (numpy / scipy only), pinned by tests/golden/sisr_harness.npz which the reference's helpers produced:

  * ``anisotropic_gaussian_kernel``  -- utils/util_sisr.py:60-92 (softmax of the quadratic form on a k x k grid)
  * ``test_kernels``                 -- the seven kernels of scripts/sisr_virnet_syn.py:103-116
  * ``bicubic_downscale``            -- the antialiased cubic resampler the reference vendors as ResizeRight/
                                        (Shocher et al., "From Discrete to Continuous Convolution Layers"): output sample i sits at
                                        i/s + (n_in-1)/2 - (n_out-1)/(2 s), the cubic is stretched by 1/s, taps are mirrored at the
                                        borders (weights evaluated at the mirrored positions) and normalised
                                        to sum 1  (ResizeRight/resize_right.py:262-318)
  * ``degrade``                      -- utils/util_sisr.py:146-177: blur (true convolution, half-sample symmetric border), clip,
                                        downsample, seeded Gaussian noise in float64, cast to fp32, clip
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np
from scipy import ndimage

KERNEL_SIZE = 21          # scripts/sisr_virnet_syn.py:92
NOISE_SEED = 1234         # utils/util_sisr.py:146


def modcrop(im: np.ndarray, sf: int) -> np.ndarray:
    h, w = im.shape[:2]
    return im[:h - h % sf, :w - w % sf]


def anisotropic_gaussian_kernel(k_size: int, sf: int, lambda_1: float, lambda_2: float, theta: float,
                                shift: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """(kernel [k,k] float64 summing to 1, (var_x, var_y, rho))."""
    c, s = math.cos(theta), math.sin(theta)
    rot = np.array([[c, -s], [s, c]])
    cov = rot @ np.diag([lambda_1, lambda_2]) @ rot.T
    prec = np.linalg.inv(cov)
    center = k_size // 2 + (0.5 * (sf - k_size % 2) if shift else 0.0)
    # the grid is built in fp32 (util_sisr.py:79) and only then promoted, so a half-pixel centre stays exact
    g = (np.arange(k_size, dtype=np.float32) - np.float32(center)).astype(np.float64)
    dx, dy = np.meshgrid(g, g)                                        # dx varies along columns
    q = -0.5 * (prec[0, 0] * dx * dx + (prec[0, 1] + prec[1, 0]) * dx * dy + prec[1, 1] * dy * dy)
    q -= q.max()
    e = np.exp(q)
    kernel = e / e.sum()
    rho = cov[0, 1] / (math.sqrt(cov[0, 0]) * math.sqrt(cov[1, 1]))
    return kernel, np.array([cov[0, 0], cov[1, 1], rho])


def test_kernels(sf: int, k_size: int = KERNEL_SIZE) -> List[np.ndarray]:
    spec = [(0.40, 0.40, 0.0), (0.60, 0.60, 0.0), (0.80, 0.80, 0.0), (0.4, 0.2, 0.0),
            (0.6, 0.3, 0.75 * np.pi), (0.8, 0.4, 0.25 * np.pi), (0.8, 0.4, 0.50 * np.pi)]
    return [anisotropic_gaussian_kernel(k_size, sf, (a * sf) ** 2, (b * sf) ** 2, th, False)[0] for a, b, th in spec]


test_kernels.__test__ = False      # not a pytest case


def _cubic(x: np.ndarray) -> np.ndarray:
    a = np.abs(x)
    a2, a3 = a * a, a * a * a
    return (1.5 * a3 - 2.5 * a2 + 1.0) * (a <= 1.0) + (-0.5 * a3 + 2.5 * a2 - 4.0 * a + 2.0) * ((a > 1.0) & (a <= 2.0))


def _resample_axis0(x: np.ndarray, scale: float, n_out: int) -> np.ndarray:
    n_in = x.shape[0]
    eps = float(np.finfo(np.float32).eps)
    support = 4.0 / scale if scale < 1.0 else 4.0                     # antialiasing widens the cubic when shrinking
    pos = np.arange(n_out) / scale + (n_in - 1) / 2 - (n_out - 1) / (2 * scale)
    left = np.ceil(pos - support / 2 - eps).astype(np.int64)
    taps = left[:, None] + np.arange(math.ceil(support - eps))
    mirror = np.concatenate([np.arange(n_in), np.arange(n_in - 1, -1, -1)])
    idx = mirror[np.remainder(taps, 2 * n_in)]
    d = pos[:, None] - idx                 # distances to the MIRRORED sample positions, as the reference computes them (:307)
    wgt = scale * _cubic(scale * d) if scale < 1.0 else _cubic(d)
    tot = wgt.sum(1, keepdims=True)
    tot[tot == 0] = 1
    wgt = wgt / tot
    return (x[idx] * wgt.reshape(wgt.shape + (1,) * (x.ndim - 1))).sum(1)


def bicubic_downscale(im: np.ndarray, sf: int) -> np.ndarray:
    """[h,w,c] -> [ceil(h/sf), ceil(w/sf), c], float64 (fp32 samples times float64 taps)."""
    scale = 1.0 / sf
    out = _resample_axis0(im, scale, math.ceil(scale * im.shape[0]))
    out = np.swapaxes(_resample_axis0(np.swapaxes(out, 0, 1), scale, math.ceil(scale * im.shape[1])), 0, 1)
    return out


def degrade(im_hr: np.ndarray, kernel: np.ndarray, sf: int, nlevel: float = 2.55, seed: int = NOISE_SEED,
            downsampler: str = "bicubic") -> np.ndarray:
    """fp32 [h,w,3] in [0,1] -> LR fp32 [h/sf, w/sf, 3]."""
    if im_hr.dtype != np.float32:
        raise TypeError("degrade expects a float32 image in [0,1]")
    blur = ndimage.convolve(im_hr, kernel[:, :, None], mode="reflect")
    blur = np.clip(blur, 0.0, 1.0)
    mode = downsampler.lower()
    if mode == "direct":
        lr = blur[::sf, ::sf].astype(np.float64)                      # noise is float64, so the sum below is too
    elif mode == "bicubic":
        lr = bicubic_downscale(blur, sf)
    else:
        raise ValueError("downsampler must be 'direct' or 'bicubic'")
    lr = lr + np.random.default_rng(seed).standard_normal(size=lr.shape) * (nlevel / 255.0)
    return np.clip(lr.astype(np.float32), 0.0, 1.0)


def sisr_table(forward, data, sf: int, nlevel: float = 2.55, kernels=None, with_ssim: bool = True):
    """The PSNR-Y / SSIM-Y table of scripts/sisr_virnet_syn.py:99-170 for any ``forward(lr float32 HWC, sf) -> sr float32 HWC``:
    per dataset and per test kernel (seven, :103-116), every ground-truth image is mod-cropped, blurred, bicubically downscaled,
    noised with the seeded stream (util_sisr.py:146-177) and restored; metrics on the uint8 Y channel with border sf**2 (:150).
    LPIPS needs the pretrained AlexNet of the `lpips` package and is outside this path.  ``data`` = ["folder:ext", ...].
    Returns rows {"dataset", "kernel", "psnr_y", "ssim_y", "images", "per_image_psnr_y"}."""
    import glob
    import os
    from . import eval as veval
    kernels = test_kernels(sf) if kernels is None else kernels
    rows = []
    for spec in data:
        folder, ext = spec.rsplit(":", 1)
        files = sorted(glob.glob(os.path.join(folder, "*." + ext.lstrip("."))))
        if not files:
            continue
        for kidx, kernel in enumerate(kernels):
            psnrs, ssims = [], []
            for f in files:
                gt = modcrop(veval.imread_rgb_uint8(f), sf)
                lr = degrade(veval.img_as_float32(gt), kernel, sf, nlevel=nlevel, downsampler="bicubic")
                sr = veval.img_as_ubyte(np.clip(forward(lr, sf), 0.0, 1.0))
                psnrs.append(veval.calculate_psnr_y(sr, gt, border=sf ** 2))
                if with_ssim:
                    ssims.append(veval.calculate_ssim(sr, gt, border=sf ** 2, ycbcr=True))
            rows.append({"dataset": os.path.basename(folder.rstrip("/")), "kernel": kidx + 1, "psnr_y": float(np.mean(psnrs)),
                         "ssim_y": float(np.mean(ssims)) if ssims else float("nan"), "images": len(files), "per_image_psnr_y": psnrs})
    return rows
