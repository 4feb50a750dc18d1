"""Host-side ELBO (virnet_amd/loss.py) against the reference's loss/ELBO_simple.py on seeded tensors (tests/golden/loss.json)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from virnet_amd.loss import elbo_denoising_simple


def test_elbo_matches_reference():
    ref = json.load(open(os.path.join(GOLDEN, "loss.json")))
    g = np.random.Generator(np.random.Philox(key=ref["seed"]))
    shape = tuple(ref["shape"])
    mu = torch.from_numpy(g.random(shape, dtype=np.float32)).requires_grad_(True)
    sigma = torch.from_numpy((g.random((2, 1, 9, 11), dtype=np.float32) * 0.05 + 1e-3)).requires_grad_(True)
    noisy = torch.from_numpy(g.random(shape, dtype=np.float32))
    gt = torch.from_numpy(g.random(shape, dtype=np.float32))
    sigma_gt = torch.from_numpy(g.random((2, 1, 9, 11), dtype=np.float32) * 0.05 + 1e-3)
    alpha0 = 0.5 * torch.tensor([ref["var_window"] ** 2], dtype=torch.float32)
    out = elbo_denoising_simple(mu, sigma, noisy, gt, ref["eps2"], alpha0, alpha0 * sigma_gt)
    for got, want in zip(out, ref["values"]):
        assert float(got) == pytest.approx(want, rel=1e-6)
    out[0].backward()
    assert float(mu.grad.double().sum()) == pytest.approx(ref["dmu_sum"], rel=1e-5)
    assert float(mu.grad.abs().max()) == pytest.approx(ref["dmu_absmax"], rel=1e-6)
    assert float(sigma.grad.double().sum()) == pytest.approx(ref["dsigma_sum"], rel=1e-5)
    assert float(sigma.grad.abs().max()) == pytest.approx(ref["dsigma_absmax"], rel=1e-6)
