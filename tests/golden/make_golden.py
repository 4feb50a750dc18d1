#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Runs only in the build container (needs /root/reference; never on the GPU
box).  It imports ``networks.VIRNet`` from the reference with the one missing
dependency (``thop``, ``utils/util_net.py:7``) stubbed, builds the script
configurations (``scripts/testing_demo.py:21-66``), loads deterministic weights
from ``virnet_amd.utils.synth`` with ``load_state_dict(strict=True)``, runs
small seeded inputs on CPU and stores inputs + outputs as ``.npz``.

Only numeric arrays are written: no reference source travels.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VIRNET_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

thop = types.ModuleType("thop")
thop.profile = lambda *a, **k: (0, 0)
sys.modules["thop"] = thop

from networks.VIRNet import VIRAttResUNet, VIRAttResUNetSR  # noqa: E402  (the reference)
from networks.KNet import KernelNet  # noqa: E402
from networks.DnCNN import DnCNN  # noqa: E402
from networks.AttResUNet import AttResBlock  # noqa: E402

from virnet_amd.utils.synth import synth_images, synth_state_dict  # noqa: E402

CONFIGS = {
    # scripts/denoising_virnet_syn.py:62-71
    "syn": dict(kind="denoise", im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=3,
                noise_cond=True, extra_mode="Input", noise_avg=False),
    # scripts/testing_demo.py:37-46
    "real": dict(kind="denoise", im_chn=3, sigma_chn=3, n_feat=[96, 160, 224, 288], dep_S=8, n_resblocks=3,
                 noise_cond=True, extra_mode="Input", noise_avg=False),
    # scripts/testing_demo.py:52-63
    "sisr": dict(kind="sisr", im_chn=3, sigma_chn=1, kernel_chn=3, n_feat=[96, 160, 224], dep_S=5, dep_K=8,
                 n_resblocks=2, noise_cond=True, kernel_cond=True, extra_mode="Both", noise_avg=True),
    # train_SISR.py:87 (add_jpeg) -> spatially varying sigma map: SFT sees per-pixel extra maps
    "sisr_varsig": dict(kind="sisr", im_chn=3, sigma_chn=1, kernel_chn=3, n_feat=[64, 96], dep_S=5, dep_K=2,
                        n_resblocks=1, noise_cond=True, kernel_cond=True, extra_mode="Down", noise_avg=False),
    # grey-scale, no conditioning: extra_mode='Null' path (AttResUNet.py:116-117,156-157)
    "small_null": dict(kind="denoise", im_chn=1, sigma_chn=1, n_feat=[64, 128], dep_S=3, n_resblocks=1,
                       noise_cond=False, extra_mode="Null", noise_avg=False),
}

CASES = [
    # (tag, config, input shape, sf)
    ("syn_a", "syn", (1, 3, 40, 52), None),
    ("syn_b", "syn", (2, 3, 37, 45), None),      # odd -> reflect pad + crop (m = 4)
    ("real_a", "real", (1, 3, 40, 52), None),    # m = 8 -> pads W 52 -> 56
    ("real_b", "real", (2, 3, 37, 45), None),
    ("sisr_x4", "sisr", (1, 3, 16, 20), 4),
    ("sisr_x2", "sisr", (2, 3, 13, 15), 2),      # 26 x 30 -> pad to 28 x 32
    ("sisr_x3", "sisr", (1, 3, 9, 11), 3),       # 27 x 33 -> pad to 28 x 36
    ("sisr_varsig_x2", "sisr_varsig", (2, 3, 11, 14), 2),
    ("small_null_a", "small_null", (2, 1, 17, 22), None),
]


def build(cfg):
    kw = {k: v for k, v in cfg.items() if k != "kind"}
    net = (VIRAttResUNet if cfg["kind"] == "denoise" else VIRAttResUNetSR)(**kw)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synth_state_dict(shapes, seed=1234), strict=True)
    return net.eval(), shapes


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    nets, manifest = {}, {"configs": CONFIGS, "shapes": {}, "cases": {}}
    for name, cfg in CONFIGS.items():
        nets[name], shapes = build(cfg)
        manifest["shapes"][name] = {k: list(s) for k, s in shapes.items()}
        print(f"{name}: {len(shapes)} tensors, {sum(int(np.prod(s)) for s in shapes.values())} params")
    for tag, cname, shape, sf in CASES:
        x = synth_images(*shape)
        if sf is None:
            mu, sigma = nets[cname](x)
            out = dict(x=x.numpy(), mu=mu.numpy(), sigma=sigma.numpy())
        else:
            mu, kinfo, sigma = nets[cname](x, sf)
            out = dict(x=x.numpy(), mu=mu.numpy(), kinfo=kinfo.numpy(), sigma=sigma.numpy())
        np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **out)
        manifest["cases"][tag] = dict(config=cname, shape=list(shape), sf=sf)
        print(f"{tag}: |mu|max={float(mu.abs().max()):.3f} sigma[{float(sigma.min()):.3e},{float(sigma.max()):.3e}]")

    # ---- sub-network vectors (each pins one oracle function) ----
    sub = {}
    # KNet alone (networks/KNet.py:41-59)
    knet = KernelNet(3, 3, num_blocks=3).eval()
    kshapes = {k: tuple(v.shape) for k, v in knet.state_dict().items()}
    ksd = synth_state_dict({("KNet." + k): s for k, s in kshapes.items()}, seed=7)
    knet.load_state_dict({k[5:]: v for k, v in ksd.items()})
    xk = synth_images(2, 3, 21, 30)
    sub["knet_x"], sub["knet_out"] = xk.numpy(), knet(xk).numpy()
    manifest["shapes"]["sub_knet"] = {("KNet." + k): list(s) for k, s in kshapes.items()}
    # DnCNN with GAP (networks/DnCNN.py:30-31)
    snet = DnCNN(3, 2, dep=4, noise_avg=True).eval()
    sshapes = {k: tuple(v.shape) for k, v in snet.state_dict().items()}
    ssd = synth_state_dict({("SNet." + k): s for k, s in sshapes.items()}, seed=9)
    snet.load_state_dict({k[5:]: v for k, v in ssd.items()})
    xs = synth_images(2, 3, 15, 18)
    sub["snet_x"], sub["snet_out"] = xs.numpy(), snet(xs).numpy()
    manifest["shapes"]["sub_snet"] = {("SNet." + k): list(s) for k, s in sshapes.items()}
    # AttResBlock with SFT on spatially varying extra maps (networks/AttResUNet.py:34-60)
    blk = AttResBlock(nf=64, extra_chn=4).eval()
    bshapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    bsd = synth_state_dict({("blk." + k): s for k, s in bshapes.items()}, seed=11)
    blk.load_state_dict({k[4:]: v for k, v in bsd.items()})
    xb = synth_images(2, 64, 12, 20) - 0.5
    eb = synth_images(2, 4, 12, 20, seed=5)
    sub["blk_x"], sub["blk_extra"], sub["blk_out"] = xb.numpy(), eb.numpy(), blk(xb, eb).numpy()
    manifest["shapes"]["sub_blk"] = {("blk." + k): list(s) for k, s in bshapes.items()}
    np.savez_compressed(os.path.join(HERE, "subnets.npz"), **sub)

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", HERE)


if __name__ == "__main__":
    main()
