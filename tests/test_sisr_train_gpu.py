"""SISR training step on the HIP path (reference train_SISR.py:207-224): the grad-mode forward against the inference
forward, every parameter gradient against torch autograd through the CPU oracle, and the reference's loop shape (elbo_sisr, gradient
clipping per sub-network, Adam)."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref
from virnet_amd.networks import VIRAttResUNetSR
from virnet_amd.utils.synth import synth_images, synth_state_dict

pytestmark = pytest.mark.gpu

SMALL = dict(im_chn=3, sigma_chn=1, kernel_chn=3, n_feat=[64, 96], dep_S=3, dep_K=2, noise_cond=True, kernel_cond=True, n_resblocks=1,
             extra_mode="Both", noise_avg=True)
# scripts/sisr_virnet_syn.py:53-63 / configs/sisr_x4.json
FULL = dict(im_chn=3, sigma_chn=1, kernel_chn=3, n_feat=[96, 160, 224], dep_S=5, dep_K=8, noise_cond=True, kernel_cond=True,
            n_resblocks=2, extra_mode="Both", noise_avg=True)


def build(cfg, seed=5):
    net = VIRAttResUNetSR(**cfg)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=seed)
    net.load_state_dict(sd, strict=True)
    return net.cuda().train(), sd


def surrogate_loss(mu, kinfo, sigma, gt):
    """Deterministic stand-in touching all three outputs the way elbo_sisr does (squared error on mu, log / ratio terms on the
    kernel descriptor and the variance) -- the ELBO itself samples, and CPU / GPU generators differ."""
    return ((mu - gt) ** 2).mean() * 50 + (kinfo[:, :2].log() ** 2).mean() + (kinfo[:, 2] ** 2).mean() * 3 + (sigma.log() ** 2).mean() * 0.01 \
        + (1.0 / sigma.clamp_min(1e-6)).mean() * 1e-4


@pytest.mark.parametrize("cfg,shape,sf", [(SMALL, (2, 3, 12, 20), 2), (SMALL, (1, 3, 9, 7), 3), (FULL, (2, 3, 16, 16), 4),
                                           (dict(SMALL, extra_mode="Down"), (2, 3, 8, 8), 4), (dict(SMALL, extra_mode="Input", kernel_cond=False), (1, 3, 10, 10), 2)])
def test_sisr_gradients_match_autograd_oracle(cfg, shape, sf):
    net, sd = build(cfg)
    x = synth_images(*shape)
    gt = synth_images(shape[0], 3, shape[2] * sf, shape[3] * sf, seed=2)
    # grad-mode forward == inference forward (the training path is the unfused spelling of the same arithmetic)
    with torch.no_grad():
        mu_i, k_i, s_i = net(x.cuda(), sf)
    mu, kinfo, sigma = net(x.cuda(), sf)
    assert mu.requires_grad and kinfo.requires_grad and sigma.requires_grad
    assert float((mu - mu_i).abs().max()) <= 2e-5 and float((kinfo - k_i).abs().max()) <= 2e-5
    assert float(((sigma - s_i).abs() / s_i).max()) <= 2e-5
    loss = surrogate_loss(mu, kinfo, sigma, gt.cuda())
    loss.backward()
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kw = {k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn", "kernel_chn")}
    mu_r, k_r, s_r = cpu_ref.virnet_sisr(ref, x, sf, **kw)
    loss_r = surrogate_loss(mu_r, k_r, s_r, gt)
    loss_r.backward()
    assert abs(float(loss.detach()) - float(loss_r.detach())) <= 1e-4 * abs(float(loss_r.detach()))
    # as in test_backward_gpu: LeakyReLU kink flips are sparse -> median at fp32 noise, worst element bounded, few tensors touched
    errs = []
    missing = [n for n, p in net.named_parameters() if p.grad is None]
    assert not missing, missing
    for name, p in net.named_parameters():
        g, gr = p.grad.cpu(), ref[name].grad
        assert g.shape == gr.shape, name
        scale = max(float(gr.abs().max()), 1e-12)
        err, med = float((g - gr).abs().max()) / scale, float((g - gr).abs().median()) / scale
        errs.append(err)
        assert med <= 5e-5 and err <= 2e-2, (name, med, err, scale)
    errs = np.asarray(errs)
    assert float(np.mean(errs > 1e-4)) <= 0.34 and float(np.median(errs)) <= 1e-4, (float(np.mean(errs > 1e-4)), float(np.median(errs)))


@pytest.mark.parametrize("factor", [1e6, 1e-7])
def test_sisr_backward_does_not_depend_on_the_loss_scale(factor):
    """The SISR step's backward rescales the three incoming gradients by one power of two at the network boundary and scales every
    parameter gradient back (train_sisr._Boundary): the gradients of `factor * loss` are factor times those of `loss`, for a factor that
    would flush the split-fp16 GEMMs' operands to zero and for one that would overflow them."""
    net, _ = build(SMALL)
    x = synth_images(2, 3, 12, 20).cuda()
    gt = synth_images(2, 3, 24, 40, seed=2).cuda()

    def grads(f):
        for p in net.parameters():
            p.grad = None
        mu, kinfo, sigma = net(x, 2)
        (surrogate_loss(mu, kinfo, sigma, gt) * f).backward()
        return {k: p.grad.double().clone() for k, p in net.named_parameters()}

    base, scaled = grads(1.0), grads(factor)
    for k in base:
        ref, got = base[k], scaled[k] / factor
        assert bool(torch.isfinite(got).all()), k
        assert float((got - ref).abs().max()) <= 1e-4 * max(float(ref.abs().max()), 1e-30), (k, factor)


def test_sisr_unscale_lives_in_the_graph():
    """The unscale is an autograd node of the forward it belongs to (train_sisr._Gate), not a hook on the leaves: a
    deep-copied module, a module with frozen sub-networks and two forwards folded into ONE backward (each with its own power of two)
    all give the gradients of the scaled loss."""
    import copy
    net, _ = build(SMALL)
    x1, x2 = synth_images(2, 3, 12, 20).cuda(), synth_images(2, 3, 12, 20, seed=5).cuda()
    gt1, gt2 = synth_images(2, 3, 24, 40, seed=2).cuda(), synth_images(2, 3, 24, 40, seed=6).cuda()

    def grads(m, terms):
        for p in m.parameters():
            p.grad = None
        loss = 0.0
        for xx, gg, f in terms:
            mu, kinfo, sigma = m(xx, 2)
            loss = loss + surrogate_loss(mu, kinfo, sigma, gg) * f
        loss.backward()
        return {k: (None if p.grad is None else p.grad.double().clone()) for k, p in m.named_parameters()}

    def close(got, ref, what):
        for k in ref:
            if ref[k] is None:
                assert got[k] is None, (what, k)
                continue
            assert bool(torch.isfinite(got[k]).all()), (what, k)
            assert float((got[k] - ref[k]).abs().max()) <= 1e-4 * max(float(ref[k].abs().max()), 1e-30), (what, k)

    g1, g2 = grads(net, [(x1, gt1, 1.0)]), grads(net, [(x2, gt2, 1.0)])
    assert not any(h for p in net.parameters() for h in (getattr(p, "_backward_hooks", None) or {}).values())   # no leaf hooks
    # (1) a deep copy made AFTER the original has run
    twin = copy.deepcopy(net)
    close({k: v / 1e6 for k, v in grads(twin, [(x1, gt1, 1e6)]).items()}, g1, "deepcopy")
    # (2) two forwards, one backward, factors 2^43 apart: each sub-graph is unscaled by its own factor
    both = grads(net, [(x1, gt1, 1e6), (x2, gt2, 1e-7)])
    close(both, {k: 1e6 * g1[k] + 1e-7 * g2[k] for k in g1}, "two forwards")
    both = grads(net, [(x1, gt1, 1e-7), (x2, gt2, 1e6)])
    close(both, {k: 1e-7 * g1[k] + 1e6 * g2[k] for k in g1}, "two forwards, swapped")
    # (3) frozen sub-networks (fine-tuning): no gradient for them, the rest unchanged
    for p in net.SNet.parameters():
        p.requires_grad_(False)
    frozen = grads(net, [(x1, gt1, 1e6)])
    assert all(v is None for k, v in frozen.items() if k.startswith("SNet."))
    close({k: (None if v is None else v / 1e6) for k, v in frozen.items() if not k.startswith("SNet.")},
          {k: v for k, v in g1.items() if not k.startswith("SNet.")}, "frozen SNet")


def test_sisr_training_loop_shape():
    """train_SISR.py:207-224: elbo_sisr on the three outputs, backward, per-sub-network gradient clipping, Adam; the loss goes down
    and the packed weights follow the parameter updates (next forward differs)."""
    from virnet_amd.loss import elbo_sisr
    net, _ = build(SMALL, seed=6)
    opt = torch.optim.Adam(net.parameters(), lr=2e-4)
    sf, n = 2, 2
    im_lr = synth_images(n, 3, 16, 16).cuda()
    im_hr = synth_images(n, 3, 32, 32, seed=3).cuda()
    kinfo_gt = torch.tensor([[1.2, 0.8, 0.1], [2.0, 1.5, -0.3]], device="cuda")
    nlevel = torch.full((n, 1, 1, 1), 2e-3, device="cuda")
    alpha0 = 0.5 * torch.tensor([9.0 ** 2], device="cuda")
    kappa0 = torch.tensor([50.0], device="cuda")
    groups = {key: [p for nm, p in net.named_parameters() if key in nm.lower()] for key in ("rnet", "snet", "knet")}
    assert sum(len(v) for v in groups.values()) == len(list(net.parameters()))
    torch.manual_seed(0)
    losses = []
    for _ in range(5):
        opt.zero_grad()
        mu, kinfo_est, sigma_est = net(im_lr, sf)
        loss, detail = elbo_sisr(mu=mu, sigma_est=sigma_est, kinfo_est=kinfo_est, im_hr=im_hr, im_lr=im_lr, sigma_prior=nlevel, alpha0=alpha0,
                                 kinfo_gt=kinfo_gt, kappa0=kappa0, r2=1e-4, eps2=1e-5, sf=sf, k_size=9, penalty_K=[0.02, 2], shift=False,
                                 downsampler="Bicubic")
        loss.backward()
        torch.nn.utils.clip_grad_norm_(groups["rnet"], 5e2)
        torch.nn.utils.clip_grad_norm_(groups["snet"], 1e2)
        torch.nn.utils.clip_grad_norm_(groups["knet"], 5e2)
        opt.step()
        assert torch.isfinite(loss) and detail[7].shape == (n, 1, 9, 9)
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses


def test_sisr_training_rejects_what_is_not_built():
    net2, _ = build(SMALL)
    with pytest.raises(RuntimeError, match="input image"):
        net2(synth_images(1, 3, 8, 8).cuda().requires_grad_(True), 2)


def _grad_parity(net, ref, name_filter=lambda k: True):
    errs = []
    missing = [n for n, p in net.named_parameters() if p.grad is None]
    assert not missing, missing
    for name, p in net.named_parameters():
        g, gr = p.grad.cpu(), ref[name].grad
        assert g.shape == gr.shape, name
        scale = max(float(gr.abs().max()), 1e-12)
        err, med = float((g - gr).abs().max()) / scale, float((g - gr).abs().median()) / scale
        errs.append(err)
        assert med <= 5e-5 and err <= 2e-2, (name, med, err, scale)
    errs = np.asarray(errs)
    assert float(np.mean(errs > 1e-4)) <= 0.34 and float(np.median(errs)) <= 1e-4, (float(np.mean(errs > 1e-4)), float(np.median(errs)))


@pytest.mark.parametrize("cfg,shape,sf", [(dict(SMALL, noise_avg=False), (2, 3, 12, 20), 2),
                                           (dict(SMALL, noise_avg=False, extra_mode="Down"), (1, 3, 9, 7), 3),
                                           (dict(SMALL, noise_avg=False, extra_mode="Input"), (2, 3, 8, 8), 4),
                                           (dict(SMALL, noise_avg=False, kernel_cond=False), (1, 3, 10, 14), 2)])
def test_sisr_per_pixel_conditioning_gradients_match_autograd_oracle(cfg, shape, sf):
    """VIRNet.py:94 (noise_avg=False: the nearest-upsampled sqrt-variance MAP conditions the head and the SFT layers; the JPEG variant
    of train_SISR.py:87): grad-mode forward == inference forward, every parameter gradient against autograd through the CPU oracle."""
    net, sd = build(cfg)
    x = synth_images(*shape)
    gt = synth_images(shape[0], 3, shape[2] * sf, shape[3] * sf, seed=2)
    with torch.no_grad():
        mu_i, k_i, s_i = net(x.cuda(), sf)
    mu, kinfo, sigma = net(x.cuda(), sf)
    assert mu.requires_grad and tuple(sigma.shape) == (shape[0], 1, shape[2], shape[3])
    assert float((mu - mu_i).abs().max()) <= 2e-5 and float(((sigma - s_i).abs() / s_i).max()) <= 2e-5
    loss = surrogate_loss(mu, kinfo, sigma, gt.cuda())
    loss.backward()
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kw = {k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn", "kernel_chn")}
    mu_r, k_r, s_r = cpu_ref.virnet_sisr(ref, x, sf, **kw)
    loss_r = surrogate_loss(mu_r, k_r, s_r, gt)
    loss_r.backward()
    assert abs(float(loss.detach()) - float(loss_r.detach())) <= 1e-4 * abs(float(loss_r.detach()))
    _grad_parity(net, ref)


DN = dict(im_chn=3, sigma_chn=1, n_feat=[64, 96], dep_S=3, n_resblocks=1, noise_cond=True, extra_mode="Both", noise_avg=False)


@pytest.mark.parametrize("cfg,shape", [(DN, (2, 3, 17, 22)), (dict(DN, extra_mode="Down", sigma_chn=3), (1, 3, 12, 12)),
                                        (dict(DN, noise_cond=False, noise_avg=True, extra_mode="Null"), (2, 3, 10, 9))])
def test_denoiser_sft_conditioning_trains_through_the_layer_nodes(cfg, shape):
    """VIRAttResUNet with extra_mode Down / Both (SFT layers fed by the per-pixel sqrt-variance map, AttResUNet.py:54-58,158,168) or a
    pooled variance: the configurations outside the fused step (train.py) -- every gradient against autograd through the CPU oracle."""
    from virnet_amd.networks import VIRAttResUNet
    net = VIRAttResUNet(**cfg)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=8)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().train()
    x = synth_images(*shape)
    gt = synth_images(*shape, seed=4)
    with torch.no_grad():
        mu_i, s_i = net(x.cuda())
    mu, sigma = net(x.cuda())
    assert mu.requires_grad and sigma.requires_grad
    assert float((mu - mu_i).abs().max()) <= 2e-5 and float(((sigma - s_i).abs() / s_i).max()) <= 2e-5

    def lossf(m, s, g):
        return ((m - g) ** 2).mean() * 50 + (s.log() ** 2).mean() * 0.01 + (1.0 / s.clamp_min(1e-6)).mean() * 1e-4

    lossf(mu, sigma, gt.cuda()).backward()
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kw = {k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn")}
    mu_r, s_r = cpu_ref.virnet_denoise(ref, x, **kw)
    lossf(mu_r, s_r, gt).backward()
    _grad_parity(net, ref)


@pytest.mark.parametrize("n,h,w,c,with_res", [(2, 9, 13, 96, True), (3, 7, 5, 160, False), (1, 33, 40, 224, True), (2, 4, 4, 8, False)])
def test_sft_backward_kernel_vs_autograd(n, h, w, c, with_res):
    """virnet_sft_backward: gradients of a = lrelu(x*mul + add, 0.2) w.r.t. x, mul, add (AttResUNet.py:54-58) from dL/da, + the skip."""
    from virnet_amd import ops
    g = torch.Generator().manual_seed(40 + c)
    x = (torch.rand(n, h, w, c, generator=g) - 0.5).requires_grad_(True)
    mul = (0.3 + torch.rand(n, c, generator=g)).requires_grad_(True)
    add = (torch.rand(n, c, generator=g) - 0.5).requires_grad_(True)
    da = torch.rand(n, h, w, c, generator=g) - 0.5
    res = torch.rand(n, h, w, c, generator=g) - 0.5 if with_res else None
    a = torch.nn.functional.leaky_relu(x * mul[:, None, None, :] + add[:, None, None, :], 0.2)
    a.backward(da)
    dx, dmul, dadd = ops.sft_backward(da.cuda(), x.detach().cuda(), mul.detach().cuda(), add.detach().cuda(), slope=0.2,
                                      res=None if res is None else res.cuda())
    ref_dx = x.grad + (res if with_res else 0)
    assert float((dx.cpu() - ref_dx).abs().max()) <= 1e-6
    assert float((dmul.cpu() - mul.grad).abs().max()) <= 2e-5 * max(1.0, float(mul.grad.abs().max()))
    assert float((dadd.cpu() - add.grad).abs().max()) <= 2e-5 * max(1.0, float(add.grad.abs().max()))
