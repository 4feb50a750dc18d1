"""GPU tests of the range guard's plumbing (ADVICE r03 engine.py:187): the replayed-graph path, the
per-thread flag / form override, the per-form packing cache and the tiled SISR path."""
import os
import threading
import warnings

import pytest
import torch

from virnet_amd import engine, ops
from virnet_amd.graph import RangeOverflow
from virnet_amd.networks import VIRAttResUNet, VIRAttResUNetSR
from virnet_amd.utils.synth import synth_images, synth_state_dict

pytestmark = pytest.mark.gpu

CFG = dict(n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input", noise_avg=False)


def _net(seed=0):
    net = VIRAttResUNet(im_chn=3, sigma_chn=1, **CFG)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=seed), strict=True)
    return net.cuda().eval()


def _hot(x):
    x = x.clone()
    x[0, :, 20:24, 20:24] = 3.0e4              # the head conv amplifies this beyond fp16's range inside RNet
    return x


@pytest.fixture(autouse=True)
def _forms(monkeypatch):
    for k in ("VIRNET_WX4_MIN_TILES", "VIRNET_WX4_MIN_COUT", "VIRNET_WX4_MIN_FILL", "VIRNET_WINOGRAD", "VIRNET_RANGE_GUARD"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("VIRNET_CONV_FORM", "wx4")
    monkeypatch.setenv("VIRNET_DETERMINISTIC", "1")      # (single small images: keep them on the Winograd form, one tile height)
    monkeypatch.setenv("VIRNET_AUTOGRAPH", "0")          # (this file is about the EAGER forward's guard and the explicit net.graphed(); the automatic
                                                         #  replay of repeated shapes has its own file, tests/test_autograph_gpu.py)


def test_graph_replay_is_range_guarded_sync():
    net = _net()
    x = synth_images(1, 3, 64, 64).cuda()
    xh = _hot(x)
    with torch.no_grad():
        with ops.forward_scope(form=engine.FP32_FORM):
            ref_hot = [t.clone() for t in engine._denoise_forward(net, xh)]
        ref_ok = [t.clone() for t in net(x)]
        g = net.graphed()
        a = [t.clone() for t in g(x)]
        assert torch.equal(a[0], ref_ok[0]) and torch.equal(a[1], ref_ok[1]) and g.reruns == 0
        with pytest.warns(RuntimeWarning, match="fp16's range"):
            b = [t.clone() for t in g(xh)]
        assert g.reruns == 1
        assert bool(torch.isfinite(b[0]).all()) and torch.equal(b[0], ref_hot[0]) and torch.equal(b[1], ref_hot[1])
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            c = [t.clone() for t in g(x)]                   # the flag is cleared inside the graph: the next replay is clean
        assert torch.equal(c[0], ref_ok[0])


def test_graph_replay_deferred_check_poisons_and_raises():
    net = _net()
    x = synth_images(1, 3, 64, 64).cuda()
    with torch.no_grad():
        ref_ok = net(x)[0].clone()
        g = net.graphed(check="deferred")
        assert torch.equal(g(x)[0], ref_ok)
        mu = g(_hot(x))[0].clone()
        assert bool(torch.isnan(mu).all())                   # loud on the device, before the host has looked
        with pytest.raises(RangeOverflow):
            g(x)
        assert torch.equal(g(x)[0], ref_ok)                  # the graph itself is intact
        g.poll()


def test_graph_follows_parameter_updates():
    net = _net(seed=0)
    other = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=7)
    x = synth_images(1, 3, 48, 40).cuda()
    with torch.no_grad():
        g = net.graphed()
        a = g(x)[0].clone()
        net.load_state_dict(other, strict=True)               # in-place copies: parameter versions move
        eager = net(x)[0].clone()
        b = g(x)[0].clone()
    assert not torch.equal(a, eager)
    assert torch.equal(b, eager)


def test_guard_rerun_keeps_both_packings_and_the_environment():
    net = _net()
    x = synth_images(1, 3, 64, 64).cuda()
    conv = net.RNet.head
    env_before = dict(os.environ)
    with torch.no_grad():
        net(x)
        pk = conv.packed()
        engine.guard_stats(reset=True)
        with pytest.warns(RuntimeWarning, match="fp16's range"):
            net(_hot(x))
        assert engine.guard_stats() == {"forwards": 1, "reruns": 1}
        assert dict(os.environ) == env_before
        assert set(conv._packs) == {"wx4", engine.FP32_FORM}
        assert conv.packed() is pk                            # the split-fp16 image survived the fp32 re-run: no repack
        net(x)
        assert conv.packed() is pk


def test_guard_is_per_thread():
    """Thread A keeps tripping the guard while thread B runs clean forwards on its own stream: B's results stay bit-identical to a
    single-threaded run, B never sees a re-run, and os.environ is never touched."""
    net = _net()
    x = synth_images(1, 3, 64, 64).cuda()
    xh = _hot(x)
    with torch.no_grad():
        ref = net(x)[0].clone()
    env_before = dict(os.environ)
    errs, res_b = [], []
    engine.guard_stats(reset=True)

    def hot():
        try:
            with torch.no_grad(), torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(6):
                    mu = net(xh)[0]
                    assert bool(torch.isfinite(mu).all())
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    def clean():
        try:
            with torch.no_grad(), torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(12):
                    res_b.append(net(x)[0].clone())
                    assert os.environ.get("VIRNET_CONV_FORM") == "wx4"
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ta, tb = threading.Thread(target=hot), threading.Thread(target=clean)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ta.start(); tb.start(); ta.join(); tb.join()
    torch.cuda.synchronize()
    assert not errs, errs
    assert engine.guard_stats() == {"forwards": 18, "reruns": 6}      # (warnings.catch_warnings is process-global: count instead)
    assert all(torch.equal(r, ref) for r in res_b)
    assert dict(os.environ) == env_before


def test_tiled_sisr_is_range_guarded():
    from virnet_amd.utils.tiling import forward_tiled_sisr
    cfg = dict(n_feat=[96, 160, 224], dep_S=5, dep_K=8, noise_cond=True, kernel_cond=True, n_resblocks=2, extra_mode="Both", noise_avg=True)
    net = VIRAttResUNetSR(im_chn=3, sigma_chn=1, kernel_chn=3, **cfg)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}), strict=True)
    net = net.cuda().eval()
    x = synth_images(1, 3, 40, 40).cuda()
    xh = x.clone()
    xh[0, :, 10:14, 10:14] = 3.0e4
    with torch.no_grad():
        with ops.forward_scope(form=engine.FP32_FORM):
            ref = engine._sisr_forward(net, xh, 2)[0].clone()
        engine.guard_stats(reset=True)
        with pytest.warns(RuntimeWarning, match="fp16's range"):
            mu, _, _ = forward_tiled_sisr(net, xh, 2, tile=256)
        assert engine.guard_stats()["reruns"] == 1
    assert bool(torch.isfinite(mu).all())
    assert float((mu - ref).abs().max()) <= 1e-3 * max(1.0, float(ref.abs().max()))


def test_guard_works_in_sequentially_created_threads():
    """ADVICE r04 (ops.py range flag): Python recycles thread idents, the library's registration is a C thread_local -- a thread created
    after another one has exited must still register ITS flag.  Every thread of the sequence trips the guard and gets the fp32 result."""
    net = _net()
    x = synth_images(1, 3, 64, 64).cuda()
    xh = _hot(x)
    with torch.no_grad():
        with ops.forward_scope(form=engine.FP32_FORM):
            ref_hot = engine._denoise_forward(net, xh)[0].clone()
    errs, idents = [], []
    engine.guard_stats(reset=True)

    def work():
        try:
            idents.append(threading.get_ident())
            with torch.no_grad(), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                mu = net(xh)[0]
                assert bool(torch.isfinite(mu).all()) and torch.equal(mu, ref_hot)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    for _ in range(5):
        t = threading.Thread(target=work)
        t.start(); t.join()
    assert not errs, errs
    assert engine.guard_stats() == {"forwards": 5, "reruns": 5}, (engine.guard_stats(), idents)


def test_eager_deferred_guard_poisons_then_repairs_in_place(monkeypatch):
    """VIRNET_GUARD_CHECK=deferred (VERDICT r04 next #5b): no host wait on the launch path -- an overflowed forward's outputs are NaN on the
    device at once, and the thread's next guarded forward (or guard_poll) repeats it with the fp32 kernels INTO the same tensors."""
    monkeypatch.setenv("VIRNET_GUARD_CHECK", "deferred")
    net = _net()
    x = synth_images(1, 3, 64, 64).cuda()
    xh = _hot(x)
    with torch.no_grad():
        with ops.forward_scope(form=engine.FP32_FORM):
            ref_hot = [t.clone() for t in engine._denoise_forward(net, xh)]
        monkeypatch.setenv("VIRNET_GUARD_CHECK", "sync")
        ref_ok = [t.clone() for t in net(x)]
        monkeypatch.setenv("VIRNET_GUARD_CHECK", "deferred")
        engine.guard_poll()
        engine.guard_stats(reset=True)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            a = net(x)
            engine.guard_poll()                                # clean forward: nothing to report
        assert torch.equal(a[0], ref_ok[0]) and torch.equal(a[1], ref_ok[1])
        mu, sig = net(xh)
        assert bool(torch.isnan(mu.clone()).all())             # loud before the host has looked (stream-ordered read of the poisoned tensor)
        with pytest.warns(engine.RangeOverflowRepaired, match="fp16's range"):
            engine.guard_poll()
        assert torch.equal(mu, ref_hot[0]) and torch.equal(sig, ref_hot[1])      # repaired in place
        assert engine.guard_stats() == {"forwards": 2, "reruns": 1}
        # the next forward's entry settles a pending overflow by itself, and clean forwards in between stay bit-identical
        # (the warning may already come from the end of net(xh) itself: its own settle pass looks at every flag copy that HAS landed, and on a
        # fast box the 64 x 64 forward is done before the host gets there -- r06: one full-suite run in five)
        with pytest.warns(engine.RangeOverflowRepaired):
            mu2, _ = net(xh)
            torch.cuda.synchronize()                            # (so that the flag copy has landed when the next forward looks)
            b = net(x)
        engine.guard_poll()
        assert torch.equal(mu2, ref_hot[0]) and torch.equal(b[0], ref_ok[0])
        # at most MAX_PENDING forwards stay unchecked
        for _ in range(6):
            net(x)
        assert len(engine._pending_list()) <= engine.MAX_PENDING
        engine.guard_poll()
        assert not engine._pending_list()


def test_deferred_repair_happens_before_the_warning_and_on_the_forward_s_stream(monkeypatch):
    """(b) with a warning filter that RAISES the exception escapes from the settling call, but the overflowed forward's
    outputs are already repaired; (c) the repair runs on the stream the forward was enqueued on, not on whatever is current at
    settle time; (a) the pinned buffers are per thread."""
    monkeypatch.setenv("VIRNET_GUARD_CHECK", "deferred")
    net = _net()
    x = synth_images(1, 3, 64, 64).cuda()
    xh = _hot(x)
    with torch.no_grad():
        with ops.forward_scope(form=engine.FP32_FORM):
            ref_hot = [t.clone() for t in engine._denoise_forward(net, xh)]
        engine.guard_poll()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            mu, sig = net(xh)
        assert engine._pending_list()[-1][5] == side
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            with pytest.raises(engine.RangeOverflowRepaired):
                engine.guard_poll()                            # settles on the DEFAULT stream's thread context
        side.synchronize()
        assert torch.equal(mu, ref_hot[0]) and torch.equal(sig, ref_hot[1])
        assert not engine._pending_list()
    assert isinstance(engine._pinned_pool(), list) and not hasattr(engine, "_PINNED_POOL")
