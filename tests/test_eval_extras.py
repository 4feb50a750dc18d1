"""Evaluation-side pieces around the hot path: the 8-way flip self-ensemble of the SIDD/DND scripts, the SISR
table of scripts/sisr_virnet_syn.py, tiled inference (the reference's forward_chop counterpart)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from virnet_amd import eval as veval
from virnet_amd import sisr_eval as se
from virnet_amd.utils.tiling import forward_tiled


def test_dihedral_group_matches_the_reference_enumeration():
    """utils/util_image.py:391-436: 0 identity, 1 flipud, 2 rot90, 3 rot90+flipud, 4 rot180, 5 rot180+flipud, 6 rot270, 7 rot270+flipud."""
    im = np.arange(2 * 3 * 1, dtype=np.float32).reshape(2, 3, 1)
    expect = [im, np.flipud(im), np.rot90(im), np.flipud(np.rot90(im)), np.rot90(im, 2), np.flipud(np.rot90(im, 2)),
              np.rot90(im, 3), np.flipud(np.rot90(im, 3))]
    seen = set()
    for mode in range(8):
        t = veval.dihedral(im, mode)
        assert np.array_equal(t, expect[mode])
        assert np.array_equal(veval.dihedral_inverse(t, mode), im)
        seen.add(t.tobytes() + bytes(t.shape))
    assert len(seen) == 8                                   # eight distinct transforms
    with pytest.raises(ValueError):
        veval.dihedral(im, 8)


def test_flip_ensemble_averages_back_transformed_outputs():
    rng = np.random.default_rng(0)
    im = rng.random((6, 9, 3), dtype=np.float32)
    assert np.allclose(veval.flip_ensemble(lambda a: a, im), im, atol=1e-6)
    # an orientation-dependent "restorer" (adds a left-to-right ramp): the ensemble symmetrises the ramp away
    def ramp(a):
        return a + np.linspace(-1, 1, a.shape[1], dtype=np.float32)[None, :, None]
    out = veval.flip_ensemble(ramp, im)
    assert np.allclose(out, im, atol=1e-6)


def test_sisr_table_plumbing_cpu():
    """scripts/sisr_virnet_syn.py:99-170 flow with a nearest-neighbour 'network': rows per kernel, Y-channel PSNR with border sf^2."""
    def forward(lr, sf):
        return np.repeat(np.repeat(lr, sf, axis=0), sf, axis=1)
    rows = se.sisr_table(forward, [os.path.join(GOLDEN, "set5") + ":bmp"], 4, kernels=se.test_kernels(4)[:2], with_ssim=False)
    assert [r["kernel"] for r in rows] == [1, 2] and all(r["images"] == 5 for r in rows)
    assert all(15.0 < r["psnr_y"] < 30.0 for r in rows) and rows[0]["psnr_y"] != rows[1]["psnr_y"]
    assert se.sisr_table(forward, [os.path.join(GOLDEN, "nowhere") + ":bmp"], 4) == []


def test_forward_tiled_equals_full_when_overlap_covers_the_receptive_field():
    """A 3-layer 3x3 conv 'network' (receptive-field radius 3): tiles with overlap >= 3 reproduce the untiled result exactly in every
    kept pixel; a x2 'SISR' variant checks the scaled stitching."""
    torch.manual_seed(0)
    ws = [torch.randn(4, 3, 3, 3) * 0.2, torch.randn(4, 4, 3, 3) * 0.2, torch.randn(3, 4, 3, 3) * 0.2]

    def net(t):
        for w in ws:
            t = F.conv2d(t, w, padding=1)
        return t

    x = torch.randn(1, 3, 70, 93)
    full = net(x)
    calls = []

    def counted(t):
        calls.append(tuple(t.shape))
        return net(t)
    tiled = forward_tiled(counted, x, tile=32, overlap=4, batch=3)
    assert tiled.shape == full.shape and torch.allclose(tiled, full, atol=1e-5)
    assert len(calls) > 1 and all(c[2:] == (32, 32) for c in calls) and max(c[0] for c in calls) == 3
    up = lambda t: F.interpolate(net(t), scale_factor=2, mode="nearest")      # noqa: E731
    assert torch.allclose(forward_tiled(up, x, tile=32, overlap=4, scale=2), up(x), atol=1e-5)
    assert torch.equal(forward_tiled(net, x, tile=128), full)                 # image fits one tile: the plain forward
    with pytest.raises(ValueError):
        forward_tiled(net, x, tile=8, overlap=4)
    with pytest.raises(ValueError):
        forward_tiled(net, torch.zeros(2, 3, 64, 64))


@pytest.mark.gpu
def test_forward_tiled_on_the_hip_path(manifest):
    """Tiled denoising through the drop-in module: tiles batched through one forward; with a tile covering the image the result is the
    plain forward bit for bit, with real tiling the kept pixels stay close to it (the U-Net's receptive field exceeds the overlap)."""
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_images, synth_state_dict
    cfg = dict(manifest["configs"]["syn"]); cfg.pop("kind")
    net = VIRAttResUNet(**cfg)
    net.load_state_dict(synth_state_dict({k: tuple(s) for k, s in manifest["shapes"]["syn"].items()}), strict=True)
    net = net.cuda().eval()
    x = synth_images(1, 3, 200, 264).cuda()
    fwd = lambda t: net(t)[0]                                                  # noqa: E731
    with torch.no_grad():
        full = fwd(x)
        assert torch.equal(forward_tiled(fwd, x, tile=512), full)
        tiled = forward_tiled(fwd, x, tile=128, overlap=32, batch=4)
    assert tiled.shape == full.shape and torch.isfinite(tiled).all()
    rel = float((tiled - full).abs().mean() / full.abs().mean())
    assert rel < 0.05, rel


@pytest.mark.gpu
def test_forward_tiled_sisr_keeps_the_conditioning_global():
    """VIRAttResUNetSR pools SNet / KNet over the whole input: tiling the MODULE would give every tile its own sigma / kernel estimate
    (visibly different from the untiled result); forward_tiled_sisr estimates both once and tiles only RNet -- the conditioning is the
    full forward's bit for bit, a tile covering the image reproduces the module exactly, real tiling stays close to it and much closer
    than the naive per-tile module call."""
    from virnet_amd.networks import VIRAttResUNetSR
    from virnet_amd.utils.synth import synth_images, synth_state_dict
    from virnet_amd.utils.tiling import forward_tiled_sisr
    cfg = dict(im_chn=3, sigma_chn=1, kernel_chn=3, n_feat=[96, 160, 224], dep_S=5, dep_K=8, noise_cond=True, kernel_cond=True,
               n_resblocks=2, extra_mode="Both", noise_avg=True)
    net = VIRAttResUNetSR(**cfg)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=5))
    net = net.cuda().eval()
    x = synth_images(1, 3, 96, 128).cuda()
    x[:, :, :, 64:] *= 0.3                                  # two halves with different statistics: per-tile pooling would differ
    with torch.no_grad():
        mu, kinfo, sigma = net(x, 2)
        mu1, k1, s1 = forward_tiled_sisr(net, x, 2, tile=256)
        assert torch.equal(mu1, mu) and torch.equal(k1, kinfo) and torch.equal(s1, sigma)
        mu_t, k_t, s_t = forward_tiled_sisr(net, x, 2, tile=64, overlap=16, batch=4)
        assert torch.equal(k_t, kinfo) and torch.equal(s_t, sigma)
        naive = forward_tiled(lambda t: net(t, 2)[0], x, tile=64, overlap=16, scale=2, batch=1)
    err_t = float((mu_t - mu).abs().mean() / mu.abs().mean())
    err_naive = float((naive - mu).abs().mean() / mu.abs().mean())
    assert err_t < 0.05 and err_t < 0.5 * err_naive, (err_t, err_naive)


@pytest.mark.gpu
def test_flip_ensemble_on_the_hip_path(manifest):
    """denoising_virnet_real_sidd.py:120-136 with the drop-in module: eight forwards on transformed inputs (two image orientations),
    HIP ensemble vs the same ensemble through the CPU oracle."""
    from oracle import cpu_ref
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_state_dict
    cfg = dict(manifest["configs"]["syn"]); cfg.pop("kind")
    sd = synth_state_dict({k: tuple(s) for k, s in manifest["shapes"]["syn"].items()})
    net = VIRAttResUNet(**cfg)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    kw = {k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn")}
    im = np.random.default_rng(3).random((36, 52, 3), dtype=np.float32)
    to_x = lambda a: torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)[None]))        # noqa: E731
    with torch.no_grad():
        hip = veval.flip_ensemble(lambda a: net(to_x(a).cuda())[0].squeeze(0).cpu().numpy().transpose(1, 2, 0), im)
        ref = veval.flip_ensemble(lambda a: cpu_ref.virnet_denoise(sd, to_x(a), **kw)[0].squeeze(0).numpy().transpose(1, 2, 0), im)
    assert hip.shape == im.shape and float(np.abs(hip - ref).max()) <= 1e-4
