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


def test_elbo_list_valued_mu_matches_reference():
    """ELBO_simple.py:30-34,43-47: ``mu`` may be a list of restorer outputs (averaged Gaussian KL and likelihood)."""
    ref = json.load(open(os.path.join(GOLDEN, "loss.json")))
    g = np.random.Generator(np.random.Philox(key=ref["seed"]))
    shape = tuple(ref["shape"])
    mu = torch.from_numpy(g.random(shape, dtype=np.float32)).requires_grad_(True)
    sigma = torch.from_numpy((g.random((2, 1, 9, 11), dtype=np.float32) * 0.05 + 1e-3)).requires_grad_(True)
    noisy = torch.from_numpy(g.random(shape, dtype=np.float32))
    gt = torch.from_numpy(g.random(shape, dtype=np.float32))
    sigma_gt = torch.from_numpy(g.random((2, 1, 9, 11), dtype=np.float32) * 0.05 + 1e-3)
    mu2 = torch.from_numpy(g.random(shape, dtype=np.float32)).requires_grad_(True)
    mu3 = torch.from_numpy(g.random(shape, dtype=np.float32)).requires_grad_(True)
    alpha0 = 0.5 * torch.tensor([ref["var_window"] ** 2], dtype=torch.float32)
    out = elbo_denoising_simple([mu, mu2, mu3], sigma, noisy, gt, ref["eps2"], alpha0, alpha0 * sigma_gt)
    L = ref["list_case"]
    for got, want in zip(out, L["values"]):
        assert float(got) == pytest.approx(want, rel=1e-6)
    out[0].backward()
    for m, want in zip((mu, mu2, mu3), L["dmu_sums"]):
        assert float(m.grad.double().sum()) == pytest.approx(want, rel=1e-5)
    assert float(sigma.grad.double().sum()) == pytest.approx(L["dsigma_sum"], rel=1e-5)
    one = elbo_denoising_simple([mu.detach()], sigma.detach(), noisy, gt, ref["eps2"], alpha0, alpha0 * sigma_gt)
    for got, want in zip(one, ref["values"]):
        assert float(got) == pytest.approx(want, rel=1e-6)          # a one-element list = the tensor form
    with pytest.raises(ValueError):
        elbo_denoising_simple([], sigma, noisy, gt, ref["eps2"], alpha0, alpha0 * sigma_gt)


@pytest.mark.parametrize("down", ["Bicubic", "Direct"])
def test_elbo_sisr_matches_reference_golden(down):
    """virnet_amd.loss.elbo_sisr against the reference's loss/ELBO_simple.py::elbo_sisr (tests/golden/loss_sisr.json, produced by
    make_loss_golden.py from the reference itself): same seeded inputs, same torch generator seed -- the restatement must consume the
    random draws in the reference's order (Gamma rsample, randn for rho, randn_like(mu)) -- values, sampled kernel and all gradients."""
    from virnet_amd.loss import elbo_sisr
    G = json.load(open(os.path.join(GOLDEN, "loss_sisr.json")))
    if G["torch_version"] != torch.__version__:
        pytest.skip(f"golden drawn with torch {G['torch_version']}, this is {torch.__version__}: the random streams may differ")
    c = G["cases"][down]
    g = np.random.Generator(np.random.Philox(key=G["seed"]))
    n, sf, (hl, wl) = G["n"], G["sf"], G["lr_hw"]
    mu = torch.from_numpy(g.random((n, 3, hl * sf, wl * sf), dtype=np.float32)).requires_grad_(True)
    sigma = torch.from_numpy(g.random((n, 1, 1, 1), dtype=np.float32) * 0.01 + 1e-4).requires_grad_(True)
    kinfo = torch.from_numpy(np.stack([g.random(n) * 3 + 0.5, g.random(n) * 3 + 0.5, g.random(n) * 1.2 - 0.6], 1).astype(np.float32)).requires_grad_(True)
    im_hr = torch.from_numpy(g.random((n, 3, hl * sf, wl * sf), dtype=np.float32))
    im_lr = torch.from_numpy(g.random((n, 3, hl, wl), dtype=np.float32))
    prior = torch.from_numpy(g.random((n, 1, 1, 1), dtype=np.float32) * 0.01 + 1e-4)
    kgt = torch.from_numpy(np.stack([g.random(n) * 3 + 0.5, g.random(n) * 3 + 0.5, g.random(n) * 1.2 - 0.6], 1).astype(np.float32))
    alpha0 = 0.5 * torch.tensor([G["var_window"] ** 2], dtype=torch.float32)
    kappa0 = torch.tensor([G["kappa0"]])
    torch.manual_seed(G["torch_seed"])
    loss, det = elbo_sisr(mu=mu, sigma_est=sigma, kinfo_est=kinfo, im_hr=im_hr, im_lr=im_lr, sigma_prior=prior, alpha0=alpha0,
                          kinfo_gt=kgt, kappa0=kappa0, r2=G["r2"], eps2=G["eps2"], sf=sf, k_size=G["k_size"], penalty_K=G["penalty_K"],
                          shift=False, downsampler=down)
    loss.backward()
    got = [float(loss)] + [float(v) for v in det[:7]]
    assert got == pytest.approx(c["values"], rel=2e-5)
    assert float(det[7].double().sum()) == pytest.approx(c["kernel_sum"], rel=1e-6) and float(det[7].max()) == pytest.approx(c["kernel_max"], rel=1e-5)
    assert [float(det[7][0, 0, 4, 4]), float(det[7][1, 0, 3, 5])] == pytest.approx(c["kernel_00"], rel=1e-5)
    assert float(mu.grad.double().sum()) == pytest.approx(c["dmu_sum"], rel=1e-4) and float(mu.grad.abs().max()) == pytest.approx(c["dmu_absmax"], rel=1e-4)
    assert [float(v) for v in sigma.grad.reshape(-1)] == pytest.approx(c["dsigma"], rel=1e-4)
    assert [float(v) for v in kinfo.grad.reshape(-1)] == pytest.approx(c["dkinfo"], rel=2e-3, abs=1e-4)


def test_blur_fft_matches_direct_sum_cpu():
    """The FFT spelling of the degradation blur (used on the device) against the grouped F.conv2d spelling, values and both gradients."""
    import torch.nn.functional as F
    from virnet_amd.loss import _xcorr_fft
    g = torch.Generator().manual_seed(5)
    n, c, h, w, k = 3, 3, 40, 52, 21
    x = torch.rand(n, c, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    ker = torch.rand(n, 1, k, k, generator=g, dtype=torch.float64)
    ker = (ker / ker.sum((2, 3), keepdim=True)).requires_grad_(True)
    pad = F.pad(x, (k // 2,) * 4, mode="reflect")
    ref = F.conv2d(pad.reshape(1, n * c, h + k - 1, w + k - 1), ker.repeat_interleave(c, 0), groups=n * c).view(n, c, h, w)
    got = _xcorr_fft(pad, ker, h, w)
    assert float((got - ref).abs().max()) <= 1e-12
    wgt = torch.rand(n, c, h, w, generator=g, dtype=torch.float64)
    gx_ref, gk_ref = torch.autograd.grad((ref * wgt).sum(), [x, ker], retain_graph=True)
    gx, gk = torch.autograd.grad((got * wgt).sum(), [x, ker])
    assert float((gx - gx_ref).abs().max()) <= 1e-12 and float((gk - gk_ref).abs().max()) <= 1e-10


@pytest.mark.gpu
def test_blur_downsample_device_path_matches_cpu():
    from virnet_amd.loss import blur_downsample
    g = torch.Generator().manual_seed(6)
    x = torch.rand(2, 3, 64, 48, generator=g)
    ker = torch.rand(2, 1, 21, 21, generator=g)
    ker = ker / ker.sum((2, 3), keepdim=True)
    for mode in ("direct", "bicubic"):
        ref = blur_downsample(x, ker, 4, mode)
        got = blur_downsample(x.cuda(), ker.cuda(), 4, mode).cpu()
        assert float((got - ref).abs().max()) <= 2e-6
