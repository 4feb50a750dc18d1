"""Evaluation-harness counterpart (virnet_amd/eval.py) against values produced by the reference's own helpers
(tests/golden/make_harness_golden.py): the shared-rng iid noise stream and uint8 PSNR.  Plus, on the GPU, the PSNR-parity line
of BASELINE.json: CBSD68 sigma=50 inputs built exactly as the reference script builds them, HIP path vs CPU oracle <= 0.01 dB."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from virnet_amd import eval as veval

H = json.load(open(os.path.join(GOLDEN, "harness.json")))
FIX = {n: veval.imread_rgb_uint8(os.path.join(GOLDEN, "cbsd68", n)) for n in H["noise_sigma50"]}


def test_psnr_matches_reference_helper():
    g = np.random.default_rng(H["psnr"]["seed"])
    a = g.integers(0, 256, size=tuple(H["psnr"]["shape"]), dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + g.integers(-9, 10, size=a.shape), 0, 255).astype(np.uint8)
    assert veval.calculate_psnr(a, b) == pytest.approx(H["psnr"]["border0"], abs=1e-12)
    assert veval.calculate_psnr(a, b, border=4) == pytest.approx(H["psnr"]["border4"], abs=1e-12)
    assert veval.calculate_psnr(a, a) == float("inf")
    with pytest.raises(ValueError):
        veval.calculate_psnr(a, a[:-1])


def test_image_conversions():
    u = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2)
    f = veval.img_as_float32(u)
    assert f.dtype == np.float32 and f.max() == 1.0 and np.array_equal(veval.img_as_ubyte(f), u)
    assert veval.img_as_ubyte(np.array([-0.2, 0.5 / 255, 1.5 / 255, 1.7])).tolist() == [0, 0, 2, 255]   # clip, round half to even


def test_sigma50_noise_is_the_scripts_stream():
    shapes = [tuple(s) for s in H["cbsd68_shapes"]]
    assert len(shapes) == 68 and H["cbsd68_names"] == sorted(H["cbsd68_names"])
    images = {v["index"]: FIX[n] for n, v in H["noise_sigma50"].items()}
    got = veval.noisy_inputs(images, shapes, 50)
    assert len(got) == len(images)
    by_index = {v["index"]: v for v in H["noise_sigma50"].values()}
    for idx, gt, noisy in got:
        noise = noisy - veval.img_as_float32(gt)
        ref = by_index[idx]
        assert noisy.dtype == np.float32 and noisy.min() < 0.0 and noisy.max() > 1.0        # not clipped
        np.testing.assert_allclose((noisy.astype(np.float64) - veval.img_as_float32(gt)).reshape(-1)[:8], ref["first8"], atol=2e-7)
        assert abs(float(noise.astype(np.float64).sum()) - ref["sum"]) < 0.5


def test_niid_maps_match_reference_helpers():
    rng = np.random.default_rng(veval.NOISE_SEED)
    maps = [veval.peaks(256), veval.sincos_kernel(), veval.gauss_kernel_mix(256, 256, rng)]
    for m, ref in zip(maps, H["niid"]["stats"]):
        assert m.shape == (256, 256)
        assert float(m.min()) == pytest.approx(ref["min"], rel=1e-6, abs=1e-12) and float(m.max()) == pytest.approx(ref["max"], rel=1e-6)
        assert float(np.asarray(m, dtype=np.float64).sum()) == pytest.approx(ref["sum"], rel=1e-6)
        np.testing.assert_allclose([m[17, 200], m[255, 0], m[128, 64]], ref["probe"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(rng.standard_normal(size=4), H["niid"]["next_normals"], rtol=0, atol=0)   # same stream position
    sig = veval.niid_sigma_maps(np.random.default_rng(veval.NOISE_SEED))
    assert all(abs(s.min() - 10 / 255) < 1e-9 and abs(s.max() - 75 / 255) < 1e-9 for s in sig)
    r = veval.resize_nearest_exact(np.arange(12.0).reshape(3, 4), 6, 6)
    assert r.shape == (6, 6) and r[0, 0] == 0 and r[5, 5] == 11 and r[2, 3] == 4 * 1 + 2


def test_y_channel_and_ssim():
    g = np.random.default_rng(H["psnr"]["seed"])
    a = g.integers(0, 256, size=tuple(H["psnr"]["shape"]), dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + g.integers(-9, 10, size=a.shape), 0, 255).astype(np.uint8)
    y = veval.rgb2y_uint8(a)
    assert int(y.astype(np.int64).sum()) == H["ycbcr"]["sum"] and y.reshape(-1)[:6].tolist() == H["ycbcr"]["first"]
    assert veval.calculate_psnr_y(a, b, border=4) == pytest.approx(H["ycbcr"]["psnr_y"], abs=1e-12)
    # SSIM is restated from its definition (the reference's needs cv2): definitional properties only
    assert veval.calculate_ssim(a, a) == pytest.approx(1.0, abs=1e-12)
    s_ab, s_ac = veval.calculate_ssim(a, b), veval.calculate_ssim(a, 255 - a)
    assert 0.5 < s_ab < 1.0 and s_ac < s_ab and veval.calculate_ssim(a, b) == veval.calculate_ssim(b, a)
    assert veval.calculate_ssim(a, b, border=3, ycbcr=True) <= 1.0
    # an independent restatement of utils/util_image.py:17-37 on another library: cv2.getGaussianKernel(11, 1.5) is exp(-(i - 5)^2 / (2 * 1.5^2))
    # normalised, cv2.filter2D is a correlation whose [5:-5, 5:-5] crop is scipy's 'valid' correlation
    from scipy.signal import correlate2d
    k = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2)); k /= k.sum()
    win = np.outer(k, k)

    def ssim_scipy(x, y):
        x, y = x.astype(np.float64), y.astype(np.float64)
        f = lambda t: correlate2d(t, win, mode="valid")        # noqa: E731
        m1, m2 = f(x), f(y)
        s1, s2, s12 = f(x * x) - m1 ** 2, f(y * y) - m2 ** 2, f(x * y) - m1 * m2
        c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
        return float((((2 * m1 * m2 + c1) * (2 * s12 + c2)) / ((m1 ** 2 + m2 ** 2 + c1) * (s1 + s2 + c2))).mean())
    want = float(np.mean([ssim_scipy(a[:, :, i], b[:, :, i]) for i in range(3)]))
    assert veval.calculate_ssim(a, b) == pytest.approx(want, abs=1e-10)
    assert veval.calculate_ssim(a, b, border=3, ycbcr=True) == pytest.approx(ssim_scipy(veval.rgb2y_uint8(a)[3:-3, 3:-3], veval.rgb2y_uint8(b)[3:-3, 3:-3]), abs=1e-10)


@pytest.mark.gpu
def test_psnr_parity_cbsd68_sigma50(manifest):
    """BASELINE.json: 'PSNR within 0.01 dB on CBSD68' -- twelve CBSD68 images (both orientations, 481x321 / 321x481: reflect pad +
    crop), sigma=50 noise from the replayed stream, denoise-syn network on identical synthetic weights: HIP path vs CPU oracle, per
    image and in the mean over the set (what the paper table reports)."""
    from oracle import cpu_ref
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_state_dict
    cfg = dict(manifest["configs"]["syn"]); cfg.pop("kind")
    net = VIRAttResUNet(**cfg)
    sd = synth_state_dict({k: tuple(s) for k, s in manifest["shapes"]["syn"].items()})
    net.load_state_dict(sd, strict=True)
    net = net.cuda()                                 # NB the script never calls .eval() (no BN/dropout): same numerics
    kw = {k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn")}
    shapes = [tuple(s) for s in H["cbsd68_shapes"]]
    images = {v["index"]: FIX[n] for n, v in H["noise_sigma50"].items()}
    assert len(images) >= 10 and {tuple(FIX[n].shape[:2]) for n in H["noise_sigma50"]} == {(321, 481), (481, 321)}
    ps, ps_ref = [], []
    for idx, gt, noisy in veval.noisy_inputs(images, shapes, 50):
        x = torch.from_numpy(noisy.transpose(2, 0, 1)[np.newaxis].copy())
        with torch.no_grad():
            mu, _ = net(x.cuda())
            mu_ref, _ = cpu_ref.virnet_denoise(sd, x, **kw)
        den = veval.img_as_ubyte(mu.squeeze(0).cpu().numpy().transpose(1, 2, 0))
        den_ref = veval.img_as_ubyte(mu_ref.squeeze(0).numpy().transpose(1, 2, 0))
        p, p_ref = veval.calculate_psnr(den, gt), veval.calculate_psnr(den_ref, gt)
        assert abs(p - p_ref) <= 0.01, (idx, p, p_ref)
        assert float((mu.cpu() - mu_ref).abs().max()) <= 1e-3
        assert int(np.abs(den.astype(int) - den_ref.astype(int)).max()) <= 1      # identical up to rounding ties
        ps.append(p)
        ps_ref.append(p_ref)
    assert abs(float(np.mean(ps)) - float(np.mean(ps_ref))) <= 0.01, (np.mean(ps), np.mean(ps_ref))


@pytest.mark.gpu
def test_niid_table_cbsd68_hip_vs_oracle(manifest):
    """BASELINE configs[0] (denoising_virnet_syn.py, niid noise on CBSD68): the evaluation-table code path the CLI
    tools/denoising_syn_eval.py runs (virnet_amd.eval.denoise_table: one shared rng, three variance maps, unclipped noise, uint8
    PSNR) over the twelve CBSD68 fixture images, HIP forward vs CPU oracle forward on identical synthetic weights: every per-image
    PSNR within 0.01 dB.  (The product has no CPU path, so configs[0]'s 'CPU plumbing' run is honoured on the GPU box.)"""
    from oracle import cpu_ref
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_state_dict
    cfg = dict(manifest["configs"]["syn"]); cfg.pop("kind")
    net = VIRAttResUNet(**cfg)
    sd = synth_state_dict({k: tuple(s) for k, s in manifest["shapes"]["syn"].items()})
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    kw = {k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn")}

    def to_x(noisy):
        return torch.from_numpy(np.ascontiguousarray(noisy.transpose(2, 0, 1)[np.newaxis]))

    def fwd_hip(noisy):
        with torch.no_grad():
            return net(to_x(noisy).cuda())[0].squeeze(0).cpu().numpy().transpose(1, 2, 0)

    def fwd_ref(noisy):
        with torch.no_grad():
            return cpu_ref.virnet_denoise(sd, to_x(noisy), **kw)[0].squeeze(0).numpy().transpose(1, 2, 0)

    data = [os.path.join(GOLDEN, "cbsd68") + ":png"]
    rows = veval.denoise_table(fwd_hip, data, "niid", with_ssim=False)
    rows_ref = veval.denoise_table(fwd_ref, data, "niid", with_ssim=False)
    assert [r["case"] for r in rows] == [1, 2, 3] and all(r["images"] == 12 for r in rows)
    for r, rr in zip(rows, rows_ref):
        for p, pr in zip(r["per_image_psnr"], rr["per_image_psnr"]):
            assert abs(p - pr) <= 0.01, (r["case"], p, pr)
        assert abs(r["psnr"] - rr["psnr"]) <= 0.01


def test_denoise_table_plumbing_cpu():
    """The table code itself (no GPU): identity 'denoiser' over the fixtures reproduces the noisy-image PSNR for both noise types,
    the rng stream is shared across cases (case 2 differs from a fresh-rng case 2), rows are ordered like the script's."""
    data = [os.path.join(GOLDEN, "cbsd68") + ":png"]
    rows = veval.denoise_table(lambda noisy: noisy, data, "iid", with_ssim=False)
    assert [r["case"] for r in rows] == [15, 25, 50]
    assert rows[0]["psnr"] > rows[1]["psnr"] > rows[2]["psnr"]
    assert abs(rows[2]["psnr"] - 14.9) < 1.0                       # sigma 50 on uint8: 20 log10(255/50) = 14.15 dB before clipping gains
    rows_n = veval.denoise_table(lambda noisy: noisy, data, "niid", with_ssim=False)
    assert [r["case"] for r in rows_n] == [1, 2, 3] and all(10.0 < r["psnr"] < 30.0 for r in rows_n)
    with pytest.raises(ValueError):
        veval.denoise_table(lambda n: n, data, "poisson")


def test_mcmaster_noise_stream_continues_behind_cbsd68():
    """scripts/denoising_virnet_syn.py:93,95,130: ONE generator for both datasets -- McMaster's sigma = 50 noise (probes written by the
    reference's own generator, tests/golden/make_mcmaster_golden.py) is reproduced by replaying the CBSD68 draws from their shapes."""
    import json
    with open(os.path.join(GOLDEN, "mcmaster.json")) as f:
        g = json.load(f)
    shapes = [tuple(s) for s in g["shapes"]]
    probes = {r["index"]: r["noise_probe"] for r in g["images"]}
    seen = 0
    for s, idx, noise in veval.iid_noise_stream(shapes, before=[[tuple(x) for x in g["cbsd68_shapes"]]]):
        if s != 50:
            continue
        p = probes[idx]
        assert np.array_equal(noise.reshape(-1)[:4], np.asarray(p["first"], np.float32)), idx
        assert abs(float(noise.astype(np.float64).sum()) - p["sum"]) <= 1e-6 * max(1.0, abs(p["sum"]))
        seen += 1
    assert seen == 18
