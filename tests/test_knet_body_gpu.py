"""The persistent KernelNet body (csrc/knet_body.hip): all RB_Layers (KNet.py:28-39) in one launch, one workgroup per
image, against the CPU oracle, the reference-generated golden and the per-layer path it replaces."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref
from virnet_amd import engine, ops
from virnet_amd.networks.KNet import KernelNet
from virnet_amd.utils.synth import synth_images, synth_state_dict
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _knet(blocks, seed=7):
    knet = KernelNet(3, 3, num_blocks=blocks)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in knet.state_dict().items()}, seed=seed)
    knet.load_state_dict(sd)
    return knet.cuda().eval(), sd


@pytest.mark.parametrize("blocks,shape", [(8, (16, 3, 64, 64)), (8, (1, 3, 64, 64)), (3, (2, 3, 21, 30)), (8, (3, 3, 40, 52)), (2, (2, 3, 4, 61)),
                                          (9, (2, 3, 33, 64))])
def test_persistent_body_vs_oracle_and_per_layer_path(monkeypatch, blocks, shape):
    """Full maps (LR 64 x 64 -> 16 x 16: the bench shape), ragged maps (6 x 8, 10 x 13, 1 x 16, 9 x 16: pixels outside the map must act as
    zero padding and stay out of the channel means), one image and many, 8 layers (one launch) and 9 (two)."""
    knet, sd = _knet(blocks)
    x = synth_images(*shape, seed=3)
    with torch.no_grad():
        ref = cpu_ref.kernel_net({"k." + k: v for k, v in sd.items()}, "k.", x, blocks)
        monkeypatch.setenv("VIRNET_KNET_PERSISTENT", "1")
        got = knet(x.cuda())
        again = knet(x.cuda())
        monkeypatch.setenv("VIRNET_KNET_PERSISTENT", "0")
        layered = knet(x.cuda())
    assert got.shape == ref.shape
    assert float((got.cpu() - ref).abs().max()) <= 1e-5, float((got.cpu() - ref).abs().max())
    assert float((got - layered).abs().max()) <= 1e-5
    assert torch.equal(got, again)                                    # fixed-order channel sums: bitwise reproducible


def test_persistent_body_feature_map(monkeypatch):
    """The body's output MAP (not just the pooled descriptor) against the oracle's RB_Layers, element for element."""
    blocks = 8
    knet, sd = _knet(blocks, seed=11)
    x = synth_images(3, 3, 64, 48, seed=5)
    psd = {"k." + k: v for k, v in sd.items()}
    with torch.no_grad():
        cur = torch.nn.functional.conv2d(x, sd["head.weight"], None, stride=4, padding=4)
        for i in range(blocks):
            cur = cpu_ref.rb_layer(psd, f"k.body.{i}.", cur)
        head = ops.conv_head_s4(x.cuda(), knet.head.weight)
        got = ops.knet_body(head, [(rb.body["0"].packed(), rb.body["2"].packed(), rb.body["3"].body["0"].weight, rb.body["3"].body["0"].bias,
                                    rb.body["3"].body["2"].weight, rb.body["3"].body["2"].bias) for rb in knet.body])
    err = float((got.permute(0, 3, 1, 2).cpu() - cur).abs().max())
    assert err <= 2e-5 * max(1.0, float(cur.abs().max())), err


def test_golden_knet_runs_the_persistent_body(manifest, monkeypatch):
    """tests/golden/subnets.npz (produced by the reference's KernelNet) through the persistent body: <= 1e-5."""
    g = load_golden("subnets")
    knet = KernelNet(3, 3, num_blocks=3)
    ksd = synth_state_dict({k: tuple(s) for k, s in manifest["shapes"]["sub_knet"].items()}, seed=7)
    knet.load_state_dict({k[5:]: v for k, v in ksd.items()})
    calls = []
    real = ops.knet_body
    monkeypatch.setattr(ops, "knet_body", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        k = knet.cuda()(torch.from_numpy(g["knet_x"]).cuda())
    assert calls == [1]
    assert float((k.cpu() - torch.from_numpy(g["knet_out"])).abs().max()) <= 1e-5


def test_large_maps_keep_the_per_layer_path(monkeypatch):
    knet, sd = _knet(2)
    x = synth_images(1, 3, 72, 64, seed=9)                            # 18 x 16 map: does not fit one workgroup
    calls = []
    real = ops.knet_body
    monkeypatch.setattr(ops, "knet_body", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        got = knet(x.cuda())
        ref = cpu_ref.kernel_net({"k." + k: v for k, v in sd.items()}, "k.", x, 2)
    assert not calls and float((got.cpu() - ref).abs().max()) <= 1e-5
    with pytest.raises(RuntimeError, match="16 x 16"):
        ops.knet_body(torch.zeros(1, 18, 16, 64, device="cuda"), [(knet.body[0].body["0"].packed(), knet.body[0].body["2"].packed(),
                                                                   knet.body[0].body["3"].body["0"].weight, knet.body[0].body["3"].body["0"].bias,
                                                                   knet.body[0].body["3"].body["2"].weight, knet.body[0].body["3"].body["2"].bias)])
