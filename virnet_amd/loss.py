"""ELBO of the denoising model, host-side PyTorch (BASELINE.json north_star: "Host code stays Python on PyTorch for tensor
plumbing and the ELBO loss").  Restated from the reference's loss/ELBO_simple.py:12-53; inputs may live on the GPU, the gradients
w.r.t. ``mu`` and ``sigma_est`` are what virnet_amd.train.DenoiseFunction.backward consumes.
"""
from __future__ import annotations

from math import log, pi, sqrt
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def kl_inverse_gamma(beta_q: Tensor, alpha_p: Tensor, beta_p: Tensor) -> Tensor:
    """KL between inverse-Gamma posteriors sharing the shape alpha_p (ELBO_simple.py:12-14)."""
    return (alpha_p * (beta_p / beta_q - 1) + alpha_p * (beta_q.log() - beta_p.log())).mean()


def kl_gauss(mu_q: Tensor, mu_p: Tensor, var_p: float) -> Tensor:
    """0.5 * mean((mu_q - mu_p)^2 / var_p)  (ELBO_simple.py:16)."""
    return 0.5 * ((mu_q - mu_p) ** 2 / var_p).mean()


def likelihood(x: Tensor, mu_q: Tensor, var_q: float, alpha_q: Tensor, beta_q: Tensor) -> Tensor:
    """Expected negative log-likelihood (ELBO_simple.py:18-21)."""
    return (0.5 * (beta_q.log() - torch.digamma(alpha_q) + alpha_q / beta_q * ((x - mu_q) ** 2 + var_q))).mean() + 0.5 * log(2 * pi)


def elbo_denoising_simple(mu, sigma_est: Tensor, im_noisy: Tensor, im_gt: Tensor, eps2: float, alpha0: Tensor,
                          beta0: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """(loss, lh, kl_gauss, kl_Igamma) (ELBO_simple.py:23-53; alpha0 = 0.5*var_window**2 and beta0 = alpha0*sigma_gt come from
    train_denoising_syn.py:157,172).  ``mu`` is the restorer's output or, as the reference allows (:30-34,43-47), a LIST of outputs
    (deep supervision): the Gaussian KL and the likelihood are then averaged over the list, the variance term is shared."""
    mus = list(mu) if isinstance(mu, (list, tuple)) else [mu]
    if not mus:
        raise ValueError("elbo_denoising_simple: empty list of restorer outputs")
    beta = sigma_est * alpha0
    klig = kl_inverse_gamma(beta, alpha0 - 1, beta0)
    klg = kl_gauss(mus[0], im_gt, eps2)
    lh = likelihood(im_noisy, mus[0], eps2, alpha0 - 1, beta)
    for m in mus[1:]:
        klg = klg + kl_gauss(m, im_gt, eps2)
        lh = lh + likelihood(im_noisy, m, eps2, alpha0 - 1, beta)
    if len(mus) > 1:
        klg, lh = klg / len(mus), lh / len(mus)
    return lh + klg + klig, lh, klg, klig


# ----------------------------------------------------------------------------------------------------------------------
# SISR (loss/ELBO_simple.py:55-138, utils/util_sisr.py:26-58,127-144; train_SISR.py:207-224)
# ----------------------------------------------------------------------------------------------------------------------
def sigma2kernel(cov: Tensor, k_size: int = 21, sf: int = 3, shift: bool = False) -> Tensor:
    """[N,1,2,2] covariance -> [N,1,k,k] normalised Gaussian kernel: softmax over the grid of -0.5 z^T cov^-1 z, z = (row, col) - centre
    (util_sisr.py:26-58; a singular covariance is nudged by 1e-5 I as there)."""
    try:
        inv = torch.inverse(cov)
    except RuntimeError:
        inv = torch.inverse(cov + torch.eye(2, dtype=cov.dtype, device=cov.device).view(1, 1, 2, 2) * 1e-5)
    center = k_size // 2 + (0.5 * (sf - k_size % 2) if shift else 0)
    g = torch.arange(k_size, device=cov.device, dtype=cov.dtype) - center
    z = torch.stack(torch.meshgrid(g, g, indexing="ij"), dim=2).view(1, -1, 2, 1)          # 1 x k^2 x 2 x 1, (row, col)
    q = -0.5 * z.transpose(2, 3).matmul(inv).matmul(z).squeeze(-1).squeeze(-1)              # N x k^2
    return F.softmax(q, dim=1).view(-1, 1, k_size, k_size)


def _bicubic_matrix(n_in: int, sf: int, device, dtype) -> Tensor:
    """Cached per (size, factor, device, dtype): the matrix is a constant of the training shape, and building it on the host + the
    pageable host-to-device copy (a sync) used to sit inside every elbo_sisr call."""
    return _bicubic_matrix_cached(int(n_in), int(sf), str(device), dtype)


import functools  # noqa: E402


@functools.lru_cache(maxsize=32)
def _bicubic_matrix_cached(n_in: int, sf: int, device: str, dtype) -> Tensor:
    """[ceil(n_in/sf), n_in] weights of the antialiased cubic the reference vendors as ResizeRight (same construction as
    virnet_amd.sisr_eval._resample_axis0: stretched cubic, taps mirrored at the borders, rows normalised)."""
    from .sisr_eval import _cubic
    import math
    scale = 1.0 / sf
    n_out = math.ceil(scale * n_in)
    eps = float(np.finfo(np.float32).eps)
    support = 4.0 / scale
    pos = np.arange(n_out) / scale + (n_in - 1) / 2 - (n_out - 1) / (2 * scale)
    left = np.ceil(pos - support / 2 - eps).astype(np.int64)
    taps = left[:, None] + np.arange(math.ceil(support - eps))
    mirror = np.concatenate([np.arange(n_in), np.arange(n_in - 1, -1, -1)])
    idx = mirror[np.remainder(taps, 2 * n_in)]
    wgt = scale * _cubic(scale * (pos[:, None] - idx))
    tot = wgt.sum(1, keepdims=True)
    tot[tot == 0] = 1
    wgt = wgt / tot
    mat = np.zeros((n_out, n_in))
    np.add.at(mat, (np.repeat(np.arange(n_out), idx.shape[1]), idx.reshape(-1)), wgt.reshape(-1))
    return torch.from_numpy(mat).to(device=torch.device(device), dtype=dtype)


def blur_downsample(im_hr: Tensor, kernel: Tensor, sf: int, downsampler: str) -> Tensor:
    """Degradation model on tensors (util_sisr.py:127-144): reflect pad, one kernel per sample (cross-correlation, as F.conv3d
    computes it there), then every sf-th sample ('direct') or the antialiased bicubic resize ('bicubic')."""
    n, c, h, w = im_hr.shape
    k = kernel.shape[-1]
    pad = F.pad(im_hr, (k // 2,) * 4, mode="reflect")
    if im_hr.is_cuda and k >= 9:
        blur = _xcorr_fft(pad, kernel, h, w)
    else:
        blur = F.conv2d(pad.reshape(1, n * c, h + 2 * (k // 2), w + 2 * (k // 2)), kernel.repeat_interleave(c, 0), groups=n * c).view(n, c, h, w)
    mode = downsampler.lower()
    if mode == "direct":
        return blur[:, :, ::sf, ::sf]
    if mode == "bicubic":
        ah = _bicubic_matrix(h, sf, blur.device, blur.dtype)
        aw = _bicubic_matrix(w, sf, blur.device, blur.dtype)
        return ah @ blur @ aw.t()
    raise ValueError("downsampler must be 'direct' or 'bicubic'")


def _xcorr_fft(pad: Tensor, kernel: Tensor, h: int, w: int) -> Tensor:
    """The same per-sample cross-correlation as the grouped ``F.conv2d`` above, through the FFT: on the device the library's depthwise
    path for 21x21 kernels costs ~9 ms per training step (forward + both gradients) against <1 ms this way.  ``pad`` [N,C,h+k-1,w+k-1],
    ``kernel`` [N,1,k,k] -> [N,C,h,w]; the transform size is rounded up to a multiple of 32 (no wrap-around reaches the kept region);
    differentiable in both arguments; agrees with the direct sum to fp32 rounding (tests/test_loss.py)."""
    hs = -(-pad.shape[-2] // 32) * 32
    ws = -(-pad.shape[-1] // 32) * 32
    fx = torch.fft.rfft2(pad, s=(hs, ws))
    fk = torch.fft.rfft2(kernel, s=(hs, ws))
    return torch.fft.irfft2(fx * fk.conj(), s=(hs, ws))[..., :h, :w]


def reparameter_cov_mat(kinfo_est: Tensor, kappa0: Tensor, rho_var: float) -> Tensor:
    """Sampled 2x2 kernel covariance (ELBO_simple.py:66-80): variances ~ 1/Gamma(kappa0-1, kappa0*lambda), correlation ~ N(rho, r2),
    off-diagonal = sqrt(v1 v2).detach() * clamp(rho, -1, 1).  Draws, in order: Gamma rsample [N,2], randn [N,1]."""
    alpha_k = torch.ones_like(kinfo_est[:, :2]) * (kappa0 - 1)
    beta_k = kinfo_est[:, :2] * kappa0
    k_var = 1 / torch.distributions.gamma.Gamma(alpha_k, beta_k).rsample()
    v1, v2 = torch.chunk(k_var, 2, dim=1)
    rho_mean = kinfo_est[:, 2].unsqueeze(1)
    rho = rho_mean + sqrt(rho_var) * torch.randn_like(rho_mean)
    direction = v1.detach().sqrt() * v2.detach().sqrt() * torch.clamp(rho, min=-1, max=1)
    return torch.cat([v1, direction, direction, v2], dim=1).view(-1, 1, 2, 2)


def likelihood_sisr(x: Tensor, kernel: Tensor, sf: int, mu_q: Tensor, var_q: float, alpha_q: Tensor, beta_q: Tensor,
                    downsampler: str) -> Tensor:
    """ELBO_simple.py:55-59: one reparameterised sample z = mu + sqrt(var) eps pushed through the degradation model."""
    zz = mu_q + torch.randn_like(mu_q) * sqrt(var_q)
    zz_blur = blur_downsample(zz, kernel, sf, downsampler)
    out = 0.5 * log(2 * pi) + 0.5 * (beta_q.log() - alpha_q.digamma()) + 0.5 * alpha_q.div(beta_q) * (x - zz_blur) ** 2
    return out.mean()


def elbo_sisr(mu: Tensor, sigma_est: Tensor, kinfo_est: Tensor, im_hr: Tensor, im_lr: Tensor, sigma_prior: Tensor, alpha0: Tensor,
              kinfo_gt: Tensor, kappa0: Tensor, r2: float, eps2: float, sf: int, k_size: int, penalty_K: Sequence[float], shift: bool,
              downsampler: str) -> Tuple[Tensor, List[Tensor]]:
    """(loss, [lh, kl_rnet, kl_snet, kl_knet, kl_knet0, kl_knet1, kl_knet2, kernel]) for a single-tensor ``mu``
    (ELBO_simple.py:82-138; called as in train_SISR.py:207-224).  Stochastic: uses torch's global generator in the reference's order."""
    kl_rnet = kl_gauss(mu, im_hr, eps2)
    beta0 = sigma_prior * alpha0
    beta = sigma_est * alpha0
    kl_snet = kl_inverse_gamma(beta, alpha0 - 1, beta0)
    kl_k0 = kl_inverse_gamma(kappa0 * kinfo_est[:, 0], kappa0 - 1, kappa0 * kinfo_gt[:, 0])
    kl_k1 = kl_inverse_gamma(kappa0 * kinfo_est[:, 1], kappa0 - 1, kappa0 * kinfo_gt[:, 1])
    kl_k2 = kl_gauss(kinfo_est[:, 2], kinfo_gt[:, 2], r2) * penalty_K[0]
    kl_knet = (kl_k0 + kl_k1 + kl_k2) / 3 * penalty_K[1]
    k_cov = reparameter_cov_mat(kinfo_est, kappa0, r2)
    kernel = sigma2kernel(k_cov, k_size, sf, shift)
    lh = likelihood_sisr(im_lr, kernel, sf, mu, eps2, alpha0 - 1, beta, downsampler)
    loss = lh + kl_rnet + kl_snet + kl_knet
    return loss, [lh, kl_rnet, kl_snet, kl_knet, kl_k0, kl_k1, kl_k2, kernel]
