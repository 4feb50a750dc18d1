"""Parity on weights that look TRAINED (VERDICT r03 weak #2 / next #5c).  Every other parity test runs on random-init weights, whose
activations say nothing about whether the split-fp16 / Winograd kernels stay inside fp16's range on a real checkpoint (none is shipped,
SURVEY F3).  Here the denoise-syn network is trained on the GPU with the product's own training step (virnet_amd/train.py: forward,
ELBO of loss/ELBO_simple.py:23-53, hand-written backward, clip, Adam -- train_denoising_syn.py:171-184 with configs/denoising_syn.json's
eps2 / var_window / clip values) for a few hundred steps on 128 x 128 crops of the CBSD68 fixtures, until it actually denoises; then
THAT state_dict goes through the HIP forward and the CPU oracle on full images at sigma = 50: PSNR within 0.01 dB, |mu| within 1e-3, and
the range guard must not have fired once."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_ref
from virnet_amd import engine
from virnet_amd import eval as veval
from virnet_amd.loss import elbo_denoising_simple
from virnet_amd.networks import VIRAttResUNet

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG = dict(im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input", noise_avg=False)
STEPS, BATCH, PATCH = 400, 16, 128


def test_trained_like_checkpoint_hip_vs_oracle():
    test_names = sorted(n for n in os.listdir(os.path.join(GOLDEN, "cbsd68")) if n.endswith(".png"))[:3]
    train_paths = sorted(os.path.join(GOLDEN, "cbsd68_rest", n) for n in os.listdir(os.path.join(GOLDEN, "cbsd68_rest")) if n.endswith(".png"))
    assert len(train_paths) >= 50
    imgs = [torch.from_numpy(veval.img_as_float32(veval.imread_rgb_uint8(p)).transpose(2, 0, 1).copy()) for p in train_paths]
    torch.manual_seed(1234)
    net = VIRAttResUNet(**CFG).cuda().train()              # torch's default init, as the reference's training starts from
    opt = torch.optim.Adam(net.parameters(), lr=2e-4)
    p_r = [p for n, p in net.named_parameters() if "rnet" in n.lower()]
    p_s = [p for n, p in net.named_parameters() if "snet" in n.lower()]
    alpha0 = 0.5 * torch.tensor([7.0 ** 2], device="cuda")
    g = torch.Generator().manual_seed(99)
    losses = []
    engine.guard_stats(reset=True)
    for step in range(STEPS):
        gt = torch.empty(BATCH, 3, PATCH, PATCH)
        for b in range(BATCH):
            im = imgs[int(torch.randint(len(imgs), (1,), generator=g))]
            y0 = int(torch.randint(im.shape[1] - PATCH + 1, (1,), generator=g)); x0 = int(torch.randint(im.shape[2] - PATCH + 1, (1,), generator=g))
            gt[b] = im[:, y0:y0 + PATCH, x0:x0 + PATCH]
        sig = (torch.rand(BATCH, 1, 1, 1, generator=g) * 70.0 + 5.0) / 255.0
        noise = torch.randn(gt.shape, generator=g) * sig
        # the data pipeline's variance map: local 7 x 7 mean of the squared noise (datasets: var_window = 7)
        sigma_gt = F.avg_pool2d(F.pad((noise ** 2).mean(1, keepdim=True), (3, 3, 3, 3), mode="reflect"), 7, 1).clamp_min(1e-10)
        gt, noisy, sigma_gt = gt.cuda(), (gt + noise).cuda(), sigma_gt.cuda()
        opt.zero_grad()
        mu, sigma = net(noisy)
        loss = elbo_denoising_simple(mu, sigma, noisy, gt, 1e-6, alpha0, alpha0 * sigma_gt)[0]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(p_r, 1e3)
        torch.nn.utils.clip_grad_norm_(p_s, 1e2)
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and np.mean(losses[-20:]) < 0.5 * np.mean(losses[:5]), (losses[:5], losses[-5:])

    net.eval()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    kw = {k: v for k, v in CFG.items() if k not in ("im_chn", "sigma_chn")}
    rng = np.random.default_rng(2024)
    engine.guard_stats(reset=True)
    gains = []
    for n in test_names:
        gt_u8 = veval.imread_rgb_uint8(os.path.join(GOLDEN, "cbsd68", n))
        gt = veval.img_as_float32(gt_u8)
        noisy = (gt + rng.standard_normal(gt.shape) * (50.0 / 255.0)).astype(np.float32)
        x = torch.from_numpy(noisy.transpose(2, 0, 1)[None].copy())
        with torch.no_grad():
            mu, sigma = net(x.cuda())
            mu_ref, sigma_ref = cpu_ref.virnet_denoise(sd, x, **kw)
        assert float((mu.cpu() - mu_ref).abs().max()) <= 1e-3, n
        assert float((sigma.cpu() - sigma_ref).abs().max()) <= 1e-3 * max(1.0, float(sigma_ref.abs().max())), n
        den = veval.img_as_ubyte(mu.squeeze(0).cpu().numpy().transpose(1, 2, 0))
        den_ref = veval.img_as_ubyte(mu_ref.squeeze(0).numpy().transpose(1, 2, 0))
        p, p_ref = veval.calculate_psnr(den, gt_u8), veval.calculate_psnr(den_ref, gt_u8)
        p_in = veval.calculate_psnr(veval.img_as_ubyte(noisy), gt_u8)
        assert abs(p - p_ref) <= 0.01, (n, p, p_ref)
        gains.append(p - p_in)
    # the checkpoint really denoises (sigma = 50 input is ~14.9 dB): its dynamics are a trained network's, not an initialisation's
    assert min(gains) >= 6.0, gains
    st = engine.guard_stats()
    assert st["forwards"] == len(test_names) and st["reruns"] == 0, st      # the activations stayed inside fp16's window
    print(f"trained-like checkpoint: loss {np.mean(losses[:5]):.3f} -> {np.mean(losses[-20:]):.3f}, PSNR gains over the noisy input {gains}, guard {st}")
