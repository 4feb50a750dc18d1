"""PSNR parity on the FULL evaluation sets against numbers produced by the reference itself (tests/golden/fullset.json, written by
tests/golden/make_fullset_golden.py from /root/reference's own network classes and helpers):
  * all 68 CBSD68 images at iid sigma = 50 (scripts/denoising_virnet_syn.py:96-135; the sigma = 15 / 25 draws replayed first),
  * the 5 Set5 images x all 7 test kernels at x4 (scripts/sisr_virnet_syn.py:85-170).
BASELINE.json: "PSNR within 0.01 dB" -- per image and in the set means.  Only HIP forwards run here (68 + 35 of them)."""
import json
import os

import numpy as np
import pytest
import torch

from virnet_amd import eval as veval
from virnet_amd import sisr_eval as se
from virnet_amd.utils.synth import synth_state_dict

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "fullset.json")) as f:
    FULL = json.load(f)
with open(os.path.join(GOLDEN, "harness.json")) as f:
    H = json.load(f)


def test_cbsd68_all_68_images_sigma50_vs_reference_psnr():
    from virnet_amd.networks import VIRAttResUNet
    g = FULL["cbsd68_sigma50"]
    assert len(g["images"]) == 68
    net = VIRAttResUNet(**g["config"])
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}), strict=True)
    net = net.cuda()
    shapes = [tuple(s) for s in H["cbsd68_shapes"]]
    names = H["cbsd68_names"]
    def path(n):          # cbsd68/ = the 12 images of the oracle-based tests, cbsd68_rest/ = the other 56
        p = os.path.join(GOLDEN, "cbsd68", n)
        return p if os.path.exists(p) else os.path.join(GOLDEN, "cbsd68_rest", n)
    images = {i: veval.imread_rgb_uint8(path(n)) for i, n in enumerate(names)}
    ref = {r["index"]: r for r in g["images"]}
    got = []
    for idx, gt, noisy in veval.noisy_inputs(images, shapes, 50):
        x = torch.from_numpy(noisy.transpose(2, 0, 1)[np.newaxis].copy())
        with torch.no_grad():
            mu, sigma = net(x.cuda())
        den = veval.img_as_ubyte(mu.squeeze(0).cpu().numpy().transpose(1, 2, 0))
        p = veval.calculate_psnr(den, gt)
        r = ref[idx]
        assert names[idx] == r["name"]
        assert abs(p - r["psnr"]) <= 0.01, (r["name"], p, r["psnr"])
        assert abs(float(mu.double().mean()) - r["mu_mean"]) <= 1e-4 and abs(float(mu.abs().max()) - r["mu_absmax"]) <= 1e-3
        assert abs(float(sigma.double().mean()) - r["sigma_mean"]) <= 1e-4 * max(1.0, abs(r["sigma_mean"]))
        got.append(p)
    assert len(got) == 68
    assert abs(float(np.mean(got)) - g["mean_psnr"]) <= 0.01, (np.mean(got), g["mean_psnr"])


def test_set5_x4_all_seven_kernels_vs_reference_psnr():
    from virnet_amd.networks.VIRNet import VIRAttResUNetSR
    g = FULL["set5_x4"]
    assert len(g["images"]) == 35
    net = VIRAttResUNetSR(**g["config"])
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=g["seed"]))
    net = net.cuda().eval()
    sf = g["sf"]
    kernels = se.test_kernels(sf)
    per_kernel = {k: [] for k in range(7)}
    for r in g["images"]:
        gt = se.modcrop(veval.imread_rgb_uint8(os.path.join(GOLDEN, "set5", r["name"])), sf)
        lr = se.degrade(veval.img_as_float32(gt), kernels[r["kernel"]], sf, nlevel=g["nlevel"], downsampler="bicubic")
        assert abs(float(np.asarray(lr, np.float64).sum()) - r["lr_sum"]) <= 1e-3 * max(1.0, abs(r["lr_sum"])) * 1e-2     # same degradation as the reference's
        x = torch.from_numpy(lr.transpose(2, 0, 1)[None].copy())
        with torch.no_grad():
            mu, kinfo, sigma = net(x.cuda(), sf)
        sr = veval.img_as_ubyte(mu.clamp(0.0, 1.0).squeeze(0).cpu().numpy().transpose(1, 2, 0))
        p = veval.calculate_psnr_y(sr, gt, border=sf ** 2)
        assert abs(p - r["psnr_y"]) <= 0.01, (r["name"], r["kernel"], p, r["psnr_y"])
        assert np.allclose(kinfo.cpu().numpy().reshape(-1), np.asarray(r["kinfo"]), rtol=1e-4, atol=1e-5)
        assert abs(float(sigma.reshape(-1)[0]) - r["sigma"]) <= 1e-4 * max(1.0, abs(r["sigma"]))
        per_kernel[r["kernel"]].append(p)
    for k in range(7):
        assert len(per_kernel[k]) == 5
        assert abs(float(np.mean(per_kernel[k])) - g["mean_psnr_y_per_kernel"][k]) <= 0.01


def test_mcmaster_all_18_images_sigma50_vs_reference_psnr():
    """The script's SECOND dataset (scripts/denoising_virnet_syn.py:93): its sigma = 50 case sits behind the 3 x 68 CBSD68 draws and its own
    sigma = 15 / 25 draws of the shared generator -- replayed from the shape lists.  500 x 500 images (not a multiple of the tile sizes)."""
    from virnet_amd.networks import VIRAttResUNet
    with open(os.path.join(GOLDEN, "mcmaster.json")) as f:
        g = json.load(f)
    assert len(g["images"]) == 18
    net = VIRAttResUNet(**g["config"])
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}), strict=True)
    net = net.cuda()
    shapes = [tuple(s) for s in g["shapes"]]
    images = {i: veval.imread_rgb_uint8(os.path.join(GOLDEN, "mcmaster", n)) for i, n in enumerate(g["names"])}
    ref = {r["index"]: r for r in g["images"]}
    got = []
    for idx, gt, noisy in veval.noisy_inputs(images, shapes, 50, before=[[tuple(s) for s in g["cbsd68_shapes"]]]):
        x = torch.from_numpy(noisy.transpose(2, 0, 1)[np.newaxis].copy())
        with torch.no_grad():
            mu, sigma = net(x.cuda())
        den = veval.img_as_ubyte(mu.squeeze(0).cpu().numpy().transpose(1, 2, 0))
        p = veval.calculate_psnr(den, gt)
        r = ref[idx]
        assert g["names"][idx] == r["name"]
        assert abs(p - r["psnr"]) <= 0.01, (r["name"], p, r["psnr"])
        assert abs(float(mu.double().mean()) - r["mu_mean"]) <= 1e-4 and abs(float(mu.abs().max()) - r["mu_absmax"]) <= 1e-3
        got.append(p)
    assert len(got) == 18
    assert abs(float(np.mean(got)) - g["mean_psnr"]) <= 0.01, (np.mean(got), g["mean_psnr"])
