"""'Runs unmodified' (BASELINE.json north_star): with compat/ ahead on the import path, the reference scripts' own import lines and
constructor calls (scripts/testing_demo.py:23-63, restated here -- nothing is read from the reference at run time) resolve to the
MI355X modules, and checkpoints load through the scripts' strict / `module.`-stripping fallback (testing_demo.py:66-70)."""
import importlib
import os
import subprocess
import sys
from collections import OrderedDict

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(REPO, "compat")

# the three load_model() branches of scripts/testing_demo.py:22-63, keyword for keyword
CALLS = {
    "denoising-syn": ("VIRAttResUNet", dict(im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True,
                                            extra_mode="Input", noise_avg=False), 82),
    "denoising-real": ("VIRAttResUNet", dict(im_chn=3, sigma_chn=3, n_feat=[96, 160, 224, 288], dep_S=8, n_resblocks=3, noise_cond=True,
                                             extra_mode='Input', noise_avg=False), 116),
    "sisr": ("VIRAttResUNetSR", dict(im_chn=3, sigma_chn=1, dep_S=5, dep_K=8, n_feat=[96, 160, 224], n_resblocks=2, extra_mode='Both',
                                     noise_avg=True, noise_cond=True, kernel_cond=True), 225),
}


@pytest.fixture()
def shim_first(monkeypatch):
    """compat/ first on sys.path and no `networks` module cached: what `PYTHONPATH=compat python scripts/...` sees."""
    for name in [m for m in sys.modules if m == "networks" or m.startswith("networks.")]:
        monkeypatch.delitem(sys.modules, name)
    monkeypatch.syspath_prepend(COMPAT)
    yield
    for name in [m for m in sys.modules if m == "networks" or m.startswith("networks.")]:
        sys.modules.pop(name, None)


@pytest.mark.parametrize("task", sorted(CALLS))
def test_reference_import_lines_and_constructors_resolve_to_this_package(shim_first, task):
    cls_name, kwargs, nkeys = CALLS[task]
    mod = importlib.import_module("networks.VIRNet")                      # `from networks.VIRNet import ...`
    assert os.path.realpath(mod.__file__).startswith(os.path.realpath(COMPAT))
    cls = getattr(mod, cls_name)
    import virnet_amd.networks as ours
    assert cls is getattr(ours, cls_name)
    net = cls(**kwargs)
    sd = net.state_dict()
    assert len(sd) == nkeys                                               # 82 / 116 / 225 tensors
    # a checkpoint saved from DistributedDataParallel carries `module.`: strict load fails, the stripped one succeeds (:66-70)
    ckpt = OrderedDict(("module." + k, torch.full_like(v, 0.25)) for k, v in sd.items())
    with pytest.raises(RuntimeError):
        net.load_state_dict(ckpt, strict=True)
    net.load_state_dict(OrderedDict({key[7:]: value for key, value in ckpt.items()}), strict=True)
    assert all(float(p.detach().min()) == 0.25 == float(p.detach().max()) for p in net.parameters())
    net.eval()
    assert hasattr(net, "SNet") and hasattr(net, "RNet") and (task != "sisr" or hasattr(net, "KNet"))


def test_shim_shadows_only_networks(shim_first):
    """`utils`, `loss`, `datasets` must keep resolving to the reference checkout: the shim ships no such packages."""
    assert sorted(d for d in os.listdir(COMPAT) if os.path.isdir(os.path.join(COMPAT, d)) and not d.startswith("__")) == ["networks"]
    for sub in ("AttResUNet", "DnCNN", "KNet"):
        assert importlib.import_module("networks." + sub).__file__.startswith(COMPAT)


def test_shim_in_a_fresh_interpreter():
    """PYTHONPATH=compat, cwd elsewhere: the import the scripts perform works from a clean process."""
    code = ("import sys; from networks.VIRNet import VIRAttResUNet, VIRAttResUNetSR; import virnet_amd.networks as n; "
            "assert VIRAttResUNet is n.VIRAttResUNet and VIRAttResUNetSR is n.VIRAttResUNetSR; print('ok')")
    env = dict(os.environ, PYTHONPATH=COMPAT)
    out = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-1500:]


@pytest.mark.gpu
def test_testing_demo_flow_through_the_shim(shim_first):
    """process_image() of scripts/testing_demo.py:77-97 restated: numpy HWC in, `.cuda()` net, no_grad forward, in-place clamp."""
    import numpy as np
    from networks.VIRNet import VIRAttResUNet, VIRAttResUNetSR
    from virnet_amd.utils.synth import synth_state_dict
    for task, sf in (("denoising-syn", None), ("sisr", 4)):
        cls_name, kwargs, _ = CALLS[task]
        net = (VIRAttResUNet if cls_name == "VIRAttResUNet" else VIRAttResUNetSR)(**kwargs).cuda()
        net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}), strict=True)
        net.eval()
        im_lq = np.random.default_rng(0).random((40, 52, 3), dtype=np.float32)
        inputs = torch.from_numpy(im_lq.transpose([2, 0, 1])).type(torch.float32).cuda().unsqueeze(0)
        with torch.no_grad():
            im_pred = net(inputs)[0] if sf is None else net(inputs, sf)[0]
        out = im_pred.clamp_(0.0, 1.0).cpu().squeeze(0).numpy().transpose([1, 2, 0])
        assert out.shape == ((40, 52, 3) if sf is None else (160, 208, 3)) and out.min() >= 0.0 and out.max() <= 1.0
