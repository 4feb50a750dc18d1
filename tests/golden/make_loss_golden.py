#!/usr/bin/env python
"""Golden values of the reference's ELBO (loss/ELBO_simple.py::elbo_denoising_simple) on seeded tensors.  Build container only."""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.environ.get("VIRNET_REFERENCE", "/root/reference"))
for name in ("cv2", "thop", "lpips", "skimage", "skimage.metrics", "skimage.color"):
    sys.modules.setdefault(name, types.ModuleType(name))
for a in ("img_as_ubyte", "img_as_float32", "img_as_float64"):
    setattr(sys.modules["skimage"], a, None)
sys.modules["thop"].profile = None
from loss.ELBO_simple import elbo_denoising_simple  # noqa: E402  (the reference)

g = np.random.Generator(np.random.Philox(key=[77, 1]))
shape = (2, 3, 9, 11)
mu = torch.from_numpy(g.random(shape, dtype=np.float32)).requires_grad_(True)
sigma = torch.from_numpy((g.random((2, 1, 9, 11), dtype=np.float32) * 0.05 + 1e-3)).requires_grad_(True)
noisy = torch.from_numpy(g.random(shape, dtype=np.float32))
gt = torch.from_numpy(g.random(shape, dtype=np.float32))
sigma_gt = torch.from_numpy(g.random((2, 1, 9, 11), dtype=np.float32) * 0.05 + 1e-3)
alpha0 = 0.5 * torch.tensor([7 ** 2], dtype=torch.float32)
out = elbo_denoising_simple(mu, sigma, noisy, gt, 1e-6, alpha0, alpha0 * sigma_gt)
out[0].backward()
single = dict(values=[float(v) for v in out], dmu_sum=float(mu.grad.double().sum()), dmu_absmax=float(mu.grad.abs().max()),
              dsigma_sum=float(sigma.grad.double().sum()), dsigma_absmax=float(sigma.grad.abs().max()))
# list-valued mu (ELBO_simple.py:30-34,43-47): two more restorer outputs drawn from the same stream
mu2 = torch.from_numpy(g.random(shape, dtype=np.float32)).requires_grad_(True)
mu3 = torch.from_numpy(g.random(shape, dtype=np.float32)).requires_grad_(True)
mu_l = mu.detach().clone().requires_grad_(True)
sigma_l = sigma.detach().clone().requires_grad_(True)
out_l = elbo_denoising_simple([mu_l, mu2, mu3], sigma_l, noisy, gt, 1e-6, alpha0, alpha0 * sigma_gt)
out_l[0].backward()
listed = dict(values=[float(v) for v in out_l], dmu_sums=[float(m.grad.double().sum()) for m in (mu_l, mu2, mu3)],
              dsigma_sum=float(sigma_l.grad.double().sum()))
json.dump(dict(seed=[77, 1], shape=list(shape), eps2=1e-6, var_window=7, list_case=listed, **single),
          open(os.path.join(HERE, "loss.json"), "w"), indent=1)
print([float(v) for v in out])


# ---- elbo_sisr (loss/ELBO_simple.py:82-138) as train_SISR.py:207-224 calls it; the three random draws come from torch's seeded
# global generator, so the restatement must draw in the same order to reproduce these values
from loss.ELBO_simple import elbo_sisr  # noqa: E402  (the reference)

out_sisr = {}
for down in ("Bicubic", "Direct"):
    g = np.random.Generator(np.random.Philox(key=[78, 2]))
    n, sf, hl, wl = 2, 2, 9, 11
    mu = torch.from_numpy(g.random((n, 3, hl * sf, wl * sf), dtype=np.float32)).requires_grad_(True)
    sigma = torch.from_numpy(g.random((n, 1, 1, 1), dtype=np.float32) * 0.01 + 1e-4).requires_grad_(True)
    kinfo = torch.from_numpy(np.stack([g.random(n) * 3 + 0.5, g.random(n) * 3 + 0.5, g.random(n) * 1.2 - 0.6], 1).astype(np.float32)).requires_grad_(True)
    im_hr = torch.from_numpy(g.random((n, 3, hl * sf, wl * sf), dtype=np.float32))
    im_lr = torch.from_numpy(g.random((n, 3, hl, wl), dtype=np.float32))
    prior = torch.from_numpy(g.random((n, 1, 1, 1), dtype=np.float32) * 0.01 + 1e-4)
    kgt = torch.from_numpy(np.stack([g.random(n) * 3 + 0.5, g.random(n) * 3 + 0.5, g.random(n) * 1.2 - 0.6], 1).astype(np.float32))
    alpha0 = 0.5 * torch.tensor([9 ** 2], dtype=torch.float32)
    kappa0 = torch.tensor([50.0])
    torch.manual_seed(4321)
    loss, det = elbo_sisr(mu=mu, sigma_est=sigma, kinfo_est=kinfo, im_hr=im_hr, im_lr=im_lr, sigma_prior=prior, alpha0=alpha0,
                          kinfo_gt=kgt, kappa0=kappa0, r2=1e-4, eps2=1e-5, sf=sf, k_size=9, penalty_K=[0.02, 2], shift=False,
                          downsampler=down)
    loss.backward()
    out_sisr[down] = dict(values=[float(loss)] + [float(v) for v in det[:7]], kernel_sum=float(det[7].double().sum()),
                          kernel_max=float(det[7].max()), kernel_00=[float(det[7][0, 0, 4, 4]), float(det[7][1, 0, 3, 5])],
                          dmu_sum=float(mu.grad.double().sum()), dmu_absmax=float(mu.grad.abs().max()),
                          dsigma=[float(v) for v in sigma.grad.reshape(-1)], dkinfo=[float(v) for v in kinfo.grad.reshape(-1)])
json.dump(dict(seed=[78, 2], torch_seed=4321, n=2, sf=2, lr_hw=[9, 11], k_size=9, var_window=9, kappa0=50.0, r2=1e-4, eps2=1e-5,
               penalty_K=[0.02, 2], cases=out_sisr, torch_version=torch.__version__),
          open(os.path.join(HERE, "loss_sisr.json"), "w"), indent=1)
print({k: v["values"][:3] for k, v in out_sisr.items()})
