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
json.dump(dict(seed=[77, 1], shape=list(shape), eps2=1e-6, var_window=7, values=[float(v) for v in out],
               dmu_sum=float(mu.grad.double().sum()), dmu_absmax=float(mu.grad.abs().max()),
               dsigma_sum=float(sigma.grad.double().sum()), dsigma_absmax=float(sigma.grad.abs().max())),
          open(os.path.join(HERE, "loss.json"), "w"), indent=1)
print([float(v) for v in out])
