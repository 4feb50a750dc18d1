"""CPU-side checks: the C-ABI library loads and exports what include/virnet_hip.h declares, the host modules mirror the
reference's state_dict, and the product path refuses to run without a ROCm device (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import REPO
from virnet_amd import _native, ops
from virnet_amd.networks import VIRAttResUNet, VIRAttResUNetSR
from virnet_amd.networks.AttResUNet import AttResUNet
from virnet_amd.utils.synth import synth_state_dict


def _header_functions():
    src = open(os.path.join(REPO, "include", "virnet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(virnet_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    declared = _header_functions()
    assert len(declared) >= 12
    bound = {name for name, _, _ in _native.SYMBOLS}
    assert set(declared) == bound, set(declared) ^ bound
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.virnet_abi_version() == _native.ABI_VERSION


def test_abi_struct_sizes_match_header_layout():
    # pointer-first structs: 8 pointers + ints/floats, no hidden padding surprises
    assert ctypes.sizeof(_native.ConvDesc) == 11 * 8 + 15 * 4 + 5 * 4  # 11 pointers, 15 ints, 5 floats
    assert ctypes.sizeof(_native.WgradDesc) == 6 * 8 + 11 * 4 + 4 + 0
    assert ctypes.sizeof(_native.PackDesc) == 4 * 8 + 14 * 4
    assert ctypes.sizeof(_native.SftWeights) == 8 * 8 + 4 * 4
    assert ctypes.sizeof(_native.ConvPlan) == 12


def test_plan_and_error_reporting_without_gpu():
    lib = _native.load()
    for cout, nrep in [(64, 2), (96, 3), (128, 4), (160, 5), (192, 3), (224, 7), (288, 3), (3, 1), (1, 1)]:
        plan = ops.get_plan(3, 1, 96, cout)
        assert (plan.nrep, plan.n_pad, plan.cin_pad) == (nrep, (cout + 31) // 32 * 32, 96)
    assert ops.get_plan(3, 1, 4, 96).cin_pad == 16
    plan = _native.ConvPlan()
    assert lib.virnet_conv_get_plan(5, 1, 3, 64, ctypes.byref(plan)) != 0
    assert b"unsupported ks=5" in lib.virnet_last_error()
    assert lib.virnet_conv_mfma(None, None) != 0 and b"NULL" in lib.virnet_last_error()


@pytest.mark.parametrize("cname", ["syn", "real", "sisr", "sisr_varsig", "small_null"])
def test_state_dict_matches_reference(manifest, cname):
    cfg = dict(manifest["configs"][cname])
    kind = cfg.pop("kind")
    net = (VIRAttResUNet if kind == "denoise" else VIRAttResUNetSR)(**cfg)
    ours = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert ours == manifest["shapes"][cname]
    assert all(v.dtype == torch.float32 for v in net.state_dict().values())
    assert not list(net.buffers())
    # DDP checkpoints carry a 'module.' prefix that loaders strip (scripts/testing_demo.py:69-72)
    sd = synth_state_dict({k: tuple(s) for k, s in manifest["shapes"][cname].items()})
    with pytest.raises(RuntimeError):
        net.load_state_dict({"module." + k: v for k, v in sd.items()}, strict=True)
    net.load_state_dict({k[7:]: v for k, v in {"module." + k: v for k, v in sd.items()}.items()}, strict=True)
    assert torch.equal(net.state_dict()["RNet.tail.weight"], sd["RNet.tail.weight"])
    assert hasattr(net, "SNet") and hasattr(net, "RNet") and (kind == "denoise" or hasattr(net, "KNet"))


def test_snet_init_is_orthogonal_with_zero_bias():
    net = VIRAttResUNet(3, sigma_chn=1, n_feat=[64, 128], dep_S=4, n_resblocks=1)
    w = net.SNet.mid_layer["0"].weight.detach().reshape(64, -1)
    gain = torch.nn.init.calculate_gain("leaky_relu", 0.25)
    assert torch.allclose(w @ w.t(), torch.eye(64) * gain ** 2, atol=1e-4)   # networks/DnCNN.py:46-52
    assert float(net.SNet.conv_last.bias.abs().max()) == 0.0


def test_constructor_contract():
    with pytest.raises(AssertionError):
        AttResUNet(extra_mode="sideways")          # networks/AttResUNet.py:113-114
    with pytest.raises(AssertionError):
        AttResUNet(n_feat=64)                      # networks/AttResUNet.py:110
    with pytest.raises(ValueError, match="blocks of 32"):
        AttResUNet(n_feat=[60, 120])
    net = AttResUNet(n_feat=(64, 128), extra_mode="BOTH")
    assert net.extra_mode == "both" and net.depth == 2
    assert isinstance(net.down_path[-1].downsampler, torch.nn.Identity)
    assert tuple(net.up_path[0].upsampler.weight.shape) == (128, 64, 2, 2)


def test_no_cpu_fallback():
    net = VIRAttResUNet(3, sigma_chn=1, n_feat=[64, 128], dep_S=3, n_resblocks=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 3, 8, 8))
    with pytest.raises(RuntimeError, match="parameters only"):
        net.RNet.head(torch.zeros(1, 4, 8, 8))
    import virnet_amd.engine as eng
    src = open(eng.__file__).read() + open(ops.__file__).read()
    assert "oracle" not in src and "F.conv2d" not in src and "conv2d(" not in src


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libvirnet_hip.so")
    with pytest.raises(_native.NativeLibraryError, match="no CPU fallback"):
        _native.load()


def test_wx4_shape_rule(monkeypatch):
    """Which launches take the Winograd-along-x kernel (ops.wx4_shape_ok): enough output channels, tiles reasonably filled, and a launch
    that fills the chip -- the headline batch on every level, a single 256 x 256 image only on the level where the two kernels are level."""
    for k in ("VIRNET_WX4_MIN_COUT", "VIRNET_WX4_MIN_TILES", "VIRNET_WX4_MIN_FILL", "VIRNET_WX4_MIN_WGS", "VIRNET_WX4_MIN_SLAB_WGS"):
        monkeypatch.delenv(k, raising=False)
    assert all(ops.wx4_shape_ok(32, s, s, c) for s, c in ((256, 96), (128, 192), (64, 288), (256, 64)))      # bench shape: all levels + SNet
    assert ops.wx4_shape_ok(64, 32, 32, 288) and ops.wx4_shape_ok(16, 64, 64, 224)                            # configs[1] level 2, SISR level 2
    assert ops.wx4_shape_ok(1, 256, 256, 96) and not ops.wx4_shape_ok(1, 64, 64, 288) and not ops.wx4_shape_ok(1, 32, 32, 288)
    # launches below a round of 96-channel workgroups still take it (8-row tiles, fewer slabs per workgroup) while single-slab work fills
    # the chip and the channel count is whole 96-blocks (SISR's 160 / 224 stay on the direct kernel's one-launch single-slab form)
    assert ops.wx4_shape_ok(1, 128, 128, 192) and ops.wx4_shape_ok(1, 128, 128, 96) and not ops.wx4_shape_ok(1, 128, 128, 160)
    monkeypatch.setenv("VIRNET_WX4_MIN_SLAB_WGS", "1000000000")
    assert not ops.wx4_shape_ok(1, 128, 128, 192)
    monkeypatch.delenv("VIRNET_WX4_MIN_SLAB_WGS")
    assert not ops.wx4_shape_ok(32, 256, 256, 32)                                                             # thin layers stay on conv_f16
    assert not ops.wx4_shape_ok(64, 17, 33, 96)                                                               # 2 x 2 tiles for 561 pixels: fill 0.27
    monkeypatch.setenv("VIRNET_WX4_MIN_WGS", "0")                                                             # form independent of the launch size
    assert ops.wx4_shape_ok(1, 128, 128, 192) and ops.wx4_shape_ok(1, 16, 32, 64)
