"""SISR degradation harness (virnet_amd/sisr_eval.py) against outputs of the reference's own helpers
(tests/golden/make_sisr_harness_golden.py), and -- on the GPU -- BASELINE.json's Set5 x4 PSNR-parity line: the LR input built as
scripts/sisr_virnet_syn.py builds it, HIP path vs CPU oracle within 0.01 dB PSNR-Y."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from virnet_amd import eval as veval
from virnet_amd import sisr_eval as se

G = np.load(os.path.join(GOLDEN, "sisr_harness.npz"))
GT = veval.imread_rgb_uint8(os.path.join(GOLDEN, "set5", "butterfly_GT.bmp"))


@pytest.mark.parametrize("sf", [2, 3, 4])
def test_seven_kernels_match_reference(sf):
    ks = se.test_kernels(sf)
    assert len(ks) == 7
    for k, ref in zip(ks, G[f"kernels_sf{sf}"]):
        assert k.shape == (21, 21) and abs(k.sum() - 1.0) < 1e-12
        np.testing.assert_allclose(k, ref, rtol=1e-10, atol=1e-16)


def test_shifted_kernel_and_info():
    k, info = se.anisotropic_gaussian_kernel(21, 4, 1.2, 5.0, 0.3, True)
    np.testing.assert_allclose(k, G["shifted_kernel"], rtol=1e-10, atol=1e-16)
    np.testing.assert_allclose(info, G["shifted_info"], rtol=1e-12)


@pytest.mark.parametrize("sf,kidx,mode", [(4, 0, "bicubic"), (4, 4, "bicubic"), (3, 5, "bicubic"), (2, 6, "direct"), (2, 1, "bicubic")])
def test_degrade_matches_reference(sf, kidx, mode):
    im = veval.img_as_float32(se.modcrop(GT, sf))
    lr = se.degrade(im, se.test_kernels(sf)[kidx], sf, nlevel=2.55, downsampler=mode)
    ref = G[f"lr_sf{sf}_k{kidx}_{mode}"]
    assert lr.dtype == np.float32 and lr.shape == ref.shape == (GT.shape[0] // sf, GT.shape[1] // sf, 3)
    assert lr.min() >= 0.0 and lr.max() <= 1.0
    np.testing.assert_allclose(lr, ref, rtol=0, atol=1.2e-7)                       # one fp32 ulp at 1.0
    assert np.array_equal(veval.img_as_ubyte(lr), veval.img_as_ubyte(ref))


def test_bicubic_downscale_properties():
    c = np.full((24, 36, 3), 0.37, dtype=np.float32)
    np.testing.assert_allclose(se.bicubic_downscale(c, 4), 0.37, atol=1e-7)        # taps sum to one, also at mirrored borders
    ramp = np.tile(np.arange(48, dtype=np.float32)[:, None, None], (1, 8, 1))
    d = se.bicubic_downscale(ramp, 4)[:, 0, 0]
    np.testing.assert_allclose(d[2:-2], np.arange(12)[2:-2] * 4 + 1.5, atol=1e-9)   # sample centres: i*sf + (sf-1)/2
    assert se.bicubic_downscale(np.zeros((10, 9, 1), np.float32), 4).shape == (3, 3, 1)   # ceil


def test_degrade_rejects_bad_input():
    with pytest.raises(TypeError):
        se.degrade(np.zeros((8, 8, 3)), se.test_kernels(2)[0], 2)
    with pytest.raises(ValueError):
        se.degrade(np.zeros((8, 8, 3), np.float32), se.test_kernels(2)[0], 2, downsampler="nearest")


@pytest.mark.gpu
def test_set5_x4_all_images_psnr_parity():
    """BASELINE.json: 'PSNR within 0.01 dB on ... Set5' -- all five Set5 images (modcropped, 228x344 ... 512x512), x4, the first and
    the sixth test kernel, nlevel 2.55, bicubic: per-image Y-PSNR of the HIP forward vs the CPU oracle and their mean over the set."""
    from oracle import cpu_ref
    from virnet_amd.networks.VIRNet import VIRAttResUNetSR
    from virnet_amd.utils.synth import synth_state_dict
    cfg = dict(im_chn=3, sigma_chn=1, kernel_chn=3, n_feat=[96, 160, 224], dep_S=5, dep_K=8, noise_cond=True, kernel_cond=True,
               n_resblocks=2, extra_mode="Both", noise_avg=True)                   # scripts/sisr_virnet_syn.py:53-63
    net = VIRAttResUNetSR(**cfg)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=5)
    net.load_state_dict(sd)
    net = net.cuda()
    kw = {k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn", "kernel_chn")}
    names = sorted(n for n in os.listdir(os.path.join(GOLDEN, "set5")) if n.endswith(".bmp"))
    assert len(names) == 5
    sf = 4
    to_u8 = lambda t: veval.img_as_ubyte(t.clamp(0.0, 1.0).squeeze(0).numpy().transpose(1, 2, 0))
    for kidx in (0, 5):
        pa_all, pb_all = [], []
        for name in names:
            gt = se.modcrop(veval.imread_rgb_uint8(os.path.join(GOLDEN, "set5", name)), sf)
            lr = se.degrade(veval.img_as_float32(gt), se.test_kernels(sf)[kidx], sf, nlevel=2.55, downsampler="bicubic")
            x = torch.from_numpy(lr.transpose(2, 0, 1)[None].copy())
            with torch.no_grad():
                mu_ref, _, _ = cpu_ref.virnet_sisr(sd, x, sf, **kw)
                mu, _, _ = net(x.cuda(), sf)
            assert (mu.cpu() - mu_ref).abs().max().item() <= 1e-3, name
            a, b = to_u8(mu.cpu()), to_u8(mu_ref)
            pa, pb = veval.calculate_psnr_y(a, gt, border=sf ** 2), veval.calculate_psnr_y(b, gt, border=sf ** 2)
            assert abs(pa - pb) <= 0.01, (name, kidx, pa, pb)
            pa_all.append(pa)
            pb_all.append(pb)
        assert abs(float(np.mean(pa_all)) - float(np.mean(pb_all))) <= 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("sf,kidx", [(4, 0), (4, 5), (2, 3)])
def test_set5_psnr_parity_hip_vs_oracle(sf, kidx):
    from oracle import cpu_ref
    from virnet_amd.networks.VIRNet import VIRAttResUNetSR
    from virnet_amd.utils.synth import synth_state_dict
    cfg = dict(im_chn=3, sigma_chn=1, kernel_chn=3, n_feat=[96, 160, 224], dep_S=5, dep_K=8, noise_cond=True, kernel_cond=True,
               n_resblocks=2, extra_mode="Both", noise_avg=True)                   # scripts/sisr_virnet_syn.py:53-63
    net = VIRAttResUNetSR(**cfg)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=5)
    net.load_state_dict(sd)
    gt = se.modcrop(GT, sf)
    lr = se.degrade(veval.img_as_float32(gt), se.test_kernels(sf)[kidx], sf, nlevel=2.55, downsampler="bicubic")
    x = torch.from_numpy(lr.transpose(2, 0, 1)[None].copy())
    with torch.no_grad():
        mu_ref, kinfo_ref, sigma_ref = cpu_ref.virnet_sisr(sd, x, sf, **{k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn", "kernel_chn")})
        mu, kinfo, sigma = net.cuda()(x.cuda(), sf)
    assert (mu.cpu() - mu_ref).abs().max().item() <= 1e-3
    assert (kinfo.cpu() - kinfo_ref).abs().max().item() <= 1e-3 * max(1.0, kinfo_ref.abs().max().item())
    to_u8 = lambda t: veval.img_as_ubyte(t.clamp(0.0, 1.0).squeeze(0).numpy().transpose(1, 2, 0))
    a, b = to_u8(mu.cpu()), to_u8(mu_ref)
    pa, pb = veval.calculate_psnr_y(a, gt, border=sf ** 2), veval.calculate_psnr_y(b, gt, border=sf ** 2)
    assert abs(pa - pb) <= 0.01, (pa, pb)
    assert (a.astype(np.int16) - b.astype(np.int16)).__abs__().max() <= 1
