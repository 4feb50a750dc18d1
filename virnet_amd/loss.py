"""ELBO of the denoising model, host-side PyTorch (BASELINE.json north_star: "Host code stays Python on PyTorch for tensor
plumbing and the ELBO loss").  Restated from the reference's loss/ELBO_simple.py:12-53; inputs may live on the GPU, the gradients
w.r.t. ``mu`` and ``sigma_est`` are what virnet_amd.train.DenoiseFunction.backward consumes.
"""
from __future__ import annotations

from math import log, pi
from typing import Tuple

import torch

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


def elbo_denoising_simple(mu: Tensor, sigma_est: Tensor, im_noisy: Tensor, im_gt: Tensor, eps2: float, alpha0: Tensor,
                          beta0: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """(loss, lh, kl_gauss, kl_Igamma) for a single-tensor ``mu`` (ELBO_simple.py:23-53; alpha0 = 0.5*var_window**2 and
    beta0 = alpha0*sigma_gt come from train_denoising_syn.py:157,172)."""
    klg = kl_gauss(mu, im_gt, eps2)
    beta = sigma_est * alpha0
    klig = kl_inverse_gamma(beta, alpha0 - 1, beta0)
    lh = likelihood(im_noisy, mu, eps2, alpha0 - 1, beta)
    return lh + klg + klig, lh, klg, klig
