"""Host logic of virnet_amd/graph.py and the knob snapshot of virnet_amd/ops.py that needs no device."""
import glob
import os
import re

import torch
from torch import nn

from conftest import REPO
from virnet_amd import graph, ops


def test_every_knob_read_through_env_is_in_the_forward_scope_snapshot():
    """inside a forward_scope `_env()` only sees the snapshot of `_KNOBS` -- a knob read through `_env` but missing there
    silently takes its default in every engine forward (VIRNET_ENTRY_FORM did)."""
    names = set()
    for path in glob.glob(os.path.join(REPO, "virnet_amd", "**", "*.py"), recursive=True):
        with open(path) as f:
            names |= set(re.findall(r"_env\(\s*\"(VIRNET_[A-Z0-9_]+)\"", f.read()))
    assert names, "no knob found: the pattern is stale"
    missing = sorted(n for n in names if n not in ops._KNOBS)
    assert not missing, f"read through ops._env() but not in ops._KNOBS: {missing}"
    with ops.forward_scope():
        os.environ["VIRNET_ENTRY_FORM"] = "f16"
        try:
            assert ops._env("VIRNET_ENTRY_FORM", "rows") == "rows"       # the snapshot was taken at scope entry
        finally:
            del os.environ["VIRNET_ENTRY_FORM"]
    os.environ["VIRNET_ENTRY_FORM"] = "f16"
    try:
        with ops.forward_scope():
            assert ops._env("VIRNET_ENTRY_FORM", "rows") == "f16"
    finally:
        del os.environ["VIRNET_ENTRY_FORM"]


class _Holder(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Parameter(torch.zeros(3))
        self.sub = nn.Linear(2, 2)


def test_fingerprint_sees_versions_every_call_and_identity_on_epoch_or_cadence(monkeypatch):
    m = _Holder()
    gf = graph.GraphedForward(lambda x: x, params=m.parameters)
    s0 = gf._fingerprint()
    assert gf._fingerprint() == s0
    with torch.no_grad():
        m.a.add_(1.0)                                    # in-place write: version
    s1 = gf._fingerprint()
    assert s1 != s0
    m.sub.weight = nn.Parameter(torch.ones(2, 2))        # registration hook -> epoch -> re-walk
    s2 = gf._fingerprint()
    assert s2 != s1
    # a write behind every hook's back: invisible between full checks ...
    monkeypatch.setattr(graph, "FULL_CHECK_EVERY", 1 << 30)
    m._parameters["a"] = nn.Parameter(torch.zeros(3))
    with torch.no_grad():
        m.a.add_(0)                                      # (give the new object the old one's version: worst case)
        m.a.add_(0)
    stale = gf._fingerprint()
    # ... caught by bump_epoch() at once
    graph.bump_epoch()
    assert gf._fingerprint() != stale
    # ... and by the cadence without any bump
    m._parameters["a"] = nn.Parameter(torch.zeros(3))
    monkeypatch.setattr(graph, "FULL_CHECK_EVERY", 1)
    assert gf._fingerprint()[0] != stale[0]


def test_convparam_apply_bumps_the_epoch():
    from virnet_amd.networks.params import ConvParam
    cp = ConvParam(16, 32, 3)
    e0 = graph._EPOCH[0]
    cp.double()
    cp.float()
    assert graph._EPOCH[0] >= e0 + 2


def test_auto_forward_bypasses_for_cpu_tensors_and_inside_no_autograph():
    calls = []

    def fn(mod, x, *a):
        calls.append((x.shape, a))
        return x

    m = _Holder()
    x = torch.zeros(1, 3, 8, 8)
    for _ in range(4):
        assert graph.auto_forward(m, fn, x) is x          # CPU tensor: straight to fn (whose own checks raise the no-CPU-path error)
    assert len(calls) == 4 and graph.auto_stats(m) == {"replays": 0, "graphs": 0, "reruns": 0}
    with graph.no_autograph():
        graph.auto_forward(m, fn, x, 4, scale=4)
    assert calls[-1][1] == (4,)


def test_every_product_knob_is_listed_in_knobs_md():
    """tools/knobs.md is the one table of the VIRNET_* knobs and launch-shape constants: every environment
    variable the product reads -- Python (`_env`, os.environ) or library (`getenv`) -- has a row there."""
    names = set()
    pats = (r"_env\(\s*\"(VIRNET_[A-Z0-9_]+)\"", r"environ(?:\.get)?\s*[\[\(]\s*\"(VIRNET_[A-Z0-9_]+)\"", r"getenv\(\s*\"(VIRNET_[A-Z0-9_]+)\"")
    files = glob.glob(os.path.join(REPO, "virnet_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(REPO, "virnet_amd", "csrc", "*.*")) + [os.path.join(REPO, "bench.py")]
    for path in files:
        if path.endswith((".py", ".hip", ".cpp", ".h")):
            with open(path) as f:
                src = f.read()
            for p in pats:
                names |= set(re.findall(p, src))
    assert len(names) >= 40
    with open(os.path.join(REPO, "tools", "knobs.md")) as f:
        doc = f.read()
    listed = set(re.findall(r"`(VIRNET_[A-Z0-9_]+)`", doc)) | {"VIRNET_FORCE" + s for s in ("_NREP", "_NW")}      # (one row: `VIRNET_FORCE_MREP` / `_NREP` / `_NW`)
    missing = sorted(names - listed)
    assert not missing, f"read by the product but not in tools/knobs.md: {missing}"
