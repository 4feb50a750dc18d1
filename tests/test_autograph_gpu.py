"""`net(x)` itself at replay latency: the modules' inference forward captures a hipGraph per (shape, device,
args) after two eager calls and replays it afterwards (virnet_amd/graph.py::auto_forward) -- the call pattern of the reference's
scripts (scripts/testing_demo.py:77-97, scripts/denoising_virnet_syn.py:133-134: one image per call, a handful of shapes per set).

What must hold: same bits as the eager forward, fresh output tensors on every call, stale weights never replayed, the range guard
still repairs an overflowing input, the bypass conditions really bypass."""
import threading
import warnings

import pytest
import torch

from virnet_amd import engine, graph, ops
from virnet_amd.networks import VIRAttResUNet, VIRAttResUNetSR
from virnet_amd.utils.synth import synth_images, synth_state_dict

pytestmark = pytest.mark.gpu
CFG = dict(n_feat=[96, 192, 288], dep_S=5, n_resblocks=2, noise_cond=True, extra_mode="Input", noise_avg=False)
SISR = dict(n_feat=[96, 160, 224], dep_S=5, dep_K=8, n_resblocks=2, noise_cond=True, kernel_cond=True, extra_mode="Both", noise_avg=True)


def _net(seed=0, sisr=False):
    net = VIRAttResUNetSR(im_chn=3, sigma_chn=1, kernel_chn=3, **SISR) if sisr else VIRAttResUNet(im_chn=3, sigma_chn=1, **CFG)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=seed), strict=True)
    return net.cuda().eval()


@pytest.fixture(autouse=True)
def _clean_env(monkeypatch):
    for k in ("VIRNET_AUTOGRAPH", "VIRNET_AUTOGRAPH_MAX_PIXELS", "VIRNET_AUTOGRAPH_MAX_GRAPHS", "VIRNET_GUARD_CHECK", "VIRNET_RANGE_GUARD",
              "VIRNET_CONV_FORM"):
        monkeypatch.delenv(k, raising=False)


def test_cbsd68_style_loop_two_alternating_shapes_replays_with_fresh_outputs():
    """The CBSD68 loop: portrait and landscape images alternate; every image is a new tensor, results are consumed (mutated) at once."""
    net = _net()
    shapes = [(1, 3, 48, 32), (1, 3, 32, 48)]
    with torch.no_grad():
        keep = []
        for i in range(12):
            x = synth_images(*shapes[i % 2], seed=100 + i).cuda()
            with graph.no_autograph():
                ref = [t.clone() for t in net(x)]
            mu, sigma = net(x)
            assert torch.equal(mu, ref[0]) and torch.equal(sigma, ref[1]), i
            keep.append((mu.clone(), mu, ref[0]))
            mu.clamp_(0.0, 1.0)                                  # scripts/testing_demo.py:95: callers write into the result
        st = graph.auto_stats(net)
        assert st["graphs"] == 2 and st["replays"] == 12 - 2 * graph.AUTO_AFTER and st["reruns"] == 0, st
        assert len({k[1].data_ptr() for k in keep}) == len(keep)     # every call returned its own tensor
        for before, got, ref in keep:
            assert torch.equal(before, ref)
            assert torch.equal(got, ref.clamp(0.0, 1.0))         # ... and later replays did not overwrite earlier (clamped) results


def test_parameter_updates_are_never_replayed_stale(monkeypatch):
    net = _net()
    x = synth_images(1, 3, 40, 40).cuda()
    with torch.no_grad():
        for _ in range(4):
            a = net(x)[0]
        assert graph.auto_stats(net)["replays"] >= 1
        net.RNet.tail.weight.mul_(1.5)                           # in-place write: _version moves
        with graph.no_autograph():
            ref = net(x)[0].clone()
        assert not torch.equal(ref, a)
        assert torch.equal(net(x)[0], ref)
        for _ in range(3):
            assert torch.equal(net(x)[0], ref)                   # (re-captured with the new weights)
        # Module.to(): new storage, same version -- ConvParam._apply bumps the registration epoch
        net.cpu(); net.cuda()
        with graph.no_autograph():
            ref2 = net(x)[0].clone()
        assert torch.equal(net(x)[0], ref2) and torch.equal(ref2, ref)
        # a write into Module._parameters fires no hook at all: caught by the periodic full validation ...
        for _ in range(3):
            net(x)
        monkeypatch.setattr(graph, "FULL_CHECK_EVERY", 1)
        tail = net.RNet.tail
        tail._parameters["weight"] = torch.nn.Parameter(tail.weight.detach() * 0.5)
        with graph.no_autograph():
            ref3 = net(x)[0].clone()
        assert not torch.equal(ref3, ref2)
        assert torch.equal(net(x)[0], ref3)
        # ... or at once after bump_epoch()
        monkeypatch.setattr(graph, "FULL_CHECK_EVERY", 1 << 30)
        for _ in range(3):
            net(x)
        tail._parameters["weight"] = torch.nn.Parameter(tail.weight.detach() * 2.0)
        graph.bump_epoch()
        assert torch.equal(net(x)[0], ref2)


def test_overflowing_input_is_repaired_through_the_replayed_graph():
    net = _net()
    x = synth_images(1, 3, 64, 64).cuda()
    xh = x.clone()
    xh[0, :, 20:24, 20:24] = 3.0e4
    with torch.no_grad():
        with ops.forward_scope(form=engine.FP32_FORM):
            ref_hot = [t.clone() for t in engine._denoise_forward(net, xh)]
        with graph.no_autograph():
            ref_ok = [t.clone() for t in net(x)]
        for _ in range(3):
            net(x)
        assert graph.auto_stats(net)["replays"] == 1
        with pytest.warns(RuntimeWarning, match="fp16's range"):
            mu, sig = net(xh)                                    # same shape: replayed, flag up, repeated with the fp32 kernels
        assert torch.equal(mu, ref_hot[0]) and torch.equal(sig, ref_hot[1])
        assert graph.auto_stats(net)["reruns"] == 1
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            mu2, _ = net(x)                                      # the flag is cleared inside the graph: a clean input is clean again
        assert torch.equal(mu2, ref_ok[0])


def test_sisr_forward_replays_and_matches_eager():
    net = _net(sisr=True)
    x = synth_images(1, 3, 24, 28).cuda()
    with torch.no_grad():
        with graph.no_autograph():
            ref = [t.clone() for t in net(x, 4)]
        for i in range(5):
            out = net(x, 4)
            assert all(torch.equal(a, b) for a, b in zip(out, ref)), i
        assert graph.auto_stats(net)["replays"] == 3
        out2 = net(x, 2)                                         # another sf = another key: eager again
        assert out2[0].shape[-1] == 56 and graph.auto_stats(net)["graphs"] == 1


def test_bypass_conditions(monkeypatch):
    net = _net()
    x = synth_images(1, 3, 32, 32).cuda()
    with torch.no_grad():
        monkeypatch.setenv("VIRNET_AUTOGRAPH", "0")
        for _ in range(4):
            net(x)
        assert graph.auto_stats(net) == {"replays": 0, "graphs": 0, "reruns": 0}
        monkeypatch.delenv("VIRNET_AUTOGRAPH")
        monkeypatch.setenv("VIRNET_AUTOGRAPH_MAX_PIXELS", "1000")        # 32 x 32 = 1024 output pixels: too large for the bound
        for _ in range(4):
            net(x)
        assert graph.auto_stats(net)["graphs"] == 0
        monkeypatch.delenv("VIRNET_AUTOGRAPH_MAX_PIXELS")
        monkeypatch.setenv("VIRNET_GUARD_CHECK", "deferred")             # the deferred guard repairs in place later: eager only
        for _ in range(4):
            net(x)
        engine.guard_poll()
        assert graph.auto_stats(net)["graphs"] == 0
        monkeypatch.delenv("VIRNET_GUARD_CHECK")
        timer = ops.LaunchTimer()
        ops.set_launch_timer(timer)
        try:
            for _ in range(4):
                net(x)
        finally:
            ops.set_launch_timer(None)
        assert graph.auto_stats(net)["graphs"] == 0 and len(timer.records) > 0
        # the LRU bound
        monkeypatch.setenv("VIRNET_AUTOGRAPH_MAX_GRAPHS", "2")
        net2 = _net()
        for hw in ((32, 32), (32, 48), (48, 32)):
            xx = synth_images(1, 3, *hw).cuda()
            for _ in range(3):
                net2(xx)
        assert graph.auto_stats(net2)["graphs"] == 2
        with pytest.raises(ValueError, match="channels"):
            net(torch.zeros(1, 1, 32, 32, device="cuda"))                 # the eager path's input errors are unchanged


def test_worker_threads_run_eagerly_beside_the_main_thread_s_graphs():
    """Automatic capture is a MAIN-THREAD feature: two host threads that each capture, replay and drop graphs abort this torch / HIP build
    (a graph destroyed from another thread than its capturer's, a pinned allocation beside another thread's capture:
    tools/probes/capture_concurrency.py).  Worker threads get the eager forward -- same bits -- while the main thread keeps replaying."""
    net = _net()
    x = synth_images(1, 3, 40, 32).cuda()
    with torch.no_grad(), graph.no_autograph():
        ref = net(x)[0].clone()
    errs, stats = [], []

    def work():
        try:
            st = torch.cuda.Stream()
            with torch.no_grad(), torch.cuda.stream(st):
                for _ in range(6):
                    got = net(x)[0]
                    st.synchronize()
                    assert torch.equal(got, ref)
                stats.append(graph.auto_stats(net))
        except Exception as e:                                   # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work) for _ in range(2)]
    for t in ts:
        t.start()
    with torch.no_grad():
        for _ in range(6):                                       # the main thread captures and replays meanwhile
            assert torch.equal(net(x)[0], ref)
    for t in ts:
        t.join()
    assert not errs, errs
    assert all(s == {"replays": 0, "graphs": 0, "reruns": 0} for s in stats), stats
    assert graph.auto_stats(net)["graphs"] == 1 and graph.auto_stats(net)["replays"] == 4


def test_a_knob_flipped_between_two_calls_of_one_shape_is_not_replayed_from_the_old_capture(monkeypatch):
    """The captured launches bake in the knobs of capture time: the environment is part of the key (tests/test_e2e_gpu.py's
    test_full_size_properties flips VIRNET_DETERMINISTIC between calls of one shape and found this)."""
    net = _net()
    x = synth_images(1, 3, 128, 128).cuda()
    with torch.no_grad():
        for _ in range(4):
            a = net(x)[0]
        assert graph.auto_stats(net)["replays"] == 2
        monkeypatch.setenv("VIRNET_CONV_FORM", "f16x3")           # another kernel family: different low bits
        with graph.no_autograph():
            ref = net(x)[0].clone()
        assert not torch.equal(ref, a)
        for _ in range(4):
            assert torch.equal(net(x)[0], ref)
        assert graph.auto_stats(net)["graphs"] == 2
        monkeypatch.delenv("VIRNET_CONV_FORM")
        assert torch.equal(net(x)[0], a)                          # the first capture is still there for the first environment
