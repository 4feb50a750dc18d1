"""hipGraph replay of a whole forward for the launch-bound small-batch path.

The reference scripts run ONE image per call (scripts/denoising_virnet_syn.py:133-134, scripts/testing_demo.py:87-93): ~45 kernel
launches whose host cost (~3 ms of Python + ctypes per forward) exceeds the GPU time below ~256x256.  Capturing the launches once
per input shape and replaying them removes that cost.  torch supplies the capture machinery
(`torch.cuda.CUDAGraph` == hipGraph on ROCm); every kernel in it is ours, launched on the capturing stream through the C ABI.

Two things a captured graph must not do silently, and does not:
  * RANGE GUARD.  engine._range_guarded cannot read the device flag while capturing, so the captured region itself starts by clearing
    the thread's range flag and ends with virnet_poison_on_flag on every output: an out-of-range forward comes back as NaN, never as a
    plausible image.  ``check="sync"`` (default) additionally reads the flag after the replay -- one device->host read, the wait the
    caller would pay anyway to consume the outputs -- and repeats THAT input eagerly with the fp32 kernels into the output buffers, like
    the eager path; ``check="deferred"`` copies the flag to pinned memory behind the replay and looks at it at the start of the NEXT
    call (or in ``poll()``): no host wait on the launch path, the overflowed call's outputs are NaN and the next call raises.
  * STALE WEIGHTS.  Packed weights are baked into the captured launches; the graphs are dropped when any parameter's storage or
    ``_version`` changes (load_state_dict, an optimizer step, .to()).  Writes through ``.data`` are invisible to that check, as they are
    to ConvParam.packed(): call ``reset()`` after one.
"""
from __future__ import annotations

import operator
import os
import threading
import warnings
import weakref
from collections import OrderedDict
from typing import Callable, Iterable, Optional, Tuple

import torch

from . import _native as nat
from . import ops


# Global registration epoch: bumped whenever ANY module registers a parameter or a sub-module (nn.Module.__setattr__ with a Parameter /
# Module, register_parameter, add_module).  A GraphedForward walks its module tree again only when the epoch has moved; between such
# events the parameter OBJECTS are the cached ones and the fingerprint is their (storage pointer, version) pairs.
# ConvParam._apply -- the only class of this package that owns parameters -- bumps it too: Module.to() / .cuda() / .cpu() replace
# ``param.data`` (new storage, same ``_version``), which the per-call version check below cannot see.
_EPOCH = [0]
FULL_CHECK_EVERY = 64       # replays between two full (storage pointer + version + identity) validations of the cached parameter list
_VERSIONS = operator.attrgetter("_version")


def _bump(*_a, **_k):
    _EPOCH[0] += 1


def bump_epoch() -> None:
    """Tell every GraphedForward that parameters may have been replaced behind the registration hooks' back (writes into
    ``Module._parameters``, ``torch.func.functional_call``-style swaps, ``param.data = ...``)."""
    _EPOCH[0] += 1


_HOOKS = []


def _install_hooks() -> None:
    """The two global registration hooks, installed when the FIRST GraphedForward is built (at import every nn.Module
    construction of the process paid the callback, whether or not a graph existed)."""
    if not _HOOKS:
        _HOOKS.append(torch.nn.modules.module.register_module_parameter_registration_hook(_bump))
        _HOOKS.append(torch.nn.modules.module.register_module_module_registration_hook(_bump))


_CAPTURE_LOCK = nat.capture_lock       # captures, every destruction of a captured graph, pinned allocations (see _native.capture_lock)


def _env_stamp() -> int:
    """One hash over the process environment.  The captured launches bake in what every VIRNET_* knob said at capture time (kernel form,
    tile height, store policy ... -- read by Python per forward and by the library per launch, tools/knobs.md), so the environment is part
    of a graph's key: a test or a caller that flips a knob between two calls of the same shape gets a fresh capture, not the old form.
    (The whole environment, not only VIRNET_*: 2 us for ~80 variables, against 9 us for filtering them.)"""
    data = getattr(os.environ, "_data", None)
    return hash(tuple(data.items())) if data is not None else hash(tuple(sorted(os.environ.items())))


# A captured graph must never be DESTROYED while any stream of the process is capturing: HIP answers hipGraphExecDestroy with "operation not
# permitted when stream is capturing" even from another thread in thread-local capture mode, torch raises that from ~CUDAGraph, and an
# exception in a destructor ends the process (tools/probes/capture_concurrency.py reproduces it; a garbage-collection pass at the wrong
# moment is enough).  So graphs are destroyed deliberately -- their last references dropped under _CAPTURE_LOCK, which every capture of this module holds --
# and graphs whose owner simply went away (GraphedForward.__del__, possibly inside someone's capture) are parked here until the next safe point.
_GRAVEYARD: list = []


def _release(holder=None) -> None:
    """Destroy the captured graphs of `holder` (a GraphedForward's key -> entry dict; emptied here) and whatever waits in the graveyard --
    by dropping the last references while _CAPTURE_LOCK is held, i.e. not while a capture of this module runs.  (Not CUDAGraph.reset():
    in this torch build a reset() graph fails its destructor's generator-state check -- "The graph should be registered to the state".)"""
    with _CAPTURE_LOCK:
        if holder is not None:
            holder.clear()
        del _GRAVEYARD[:]


class RangeOverflow(RuntimeError):
    """Raised by a ``check="deferred"`` GraphedForward when the PREVIOUS replay staged an operand outside fp16's range."""


class GraphedForward:
    """Callable that replays `fn(static_input, *args)` from a captured graph; one graph per (shape, device, args).

    ``auto_after`` = k > 0 (the modules' own ``forward``, see ``auto_forward``): the first k calls of a key run ``fn`` eagerly and
    only then is the key captured.  ``fresh`` : return CLONES of the graph's output buffers (what ``nn.Module.forward`` callers
    expect: scripts/testing_demo.py:95 clamps the result in place).  ``max_graphs``: least-recently-used bound on the captured graphs
    (each holds its intermediates in a private pool)."""

    def __init__(self, fn: Callable, warmup: int = 2, params: Optional[Callable[[], Iterable[torch.Tensor]]] = None,
                 check: str = "sync", auto_after: int = 0, fresh: bool = False, max_graphs: int = 0):
        if check not in ("sync", "deferred", "off"):
            raise ValueError(f"check={check!r}: expected 'sync', 'deferred' or 'off'")
        _install_hooks()
        self.fn, self.warmup, self.check = fn, warmup, check
        self.auto_after, self.fresh, self.max_graphs = auto_after, fresh, max_graphs
        self._params = params
        self._stamp = None
        self._plist = None             # (registration epoch, parameter objects, their (id, storage pointer) pairs) -- see _EPOCH
        self._calls = 0
        self._graphs: "OrderedDict[Tuple, Tuple]" = OrderedDict()
        self._seen: dict = {}          # auto mode: key -> eager calls so far
        self._pending = None           # (pinned flag copy, event) of the last deferred replay
        self.reruns = 0                # replays repeated with the fp32 kernels (check="sync")
        self.replays = 0

    # ---- parameter fingerprint.  Per call: the `_version` of every cached parameter object (an optimizer step, load_state_dict, any
    # in-place write move it) -- ~0.09 us per parameter.  When the registration epoch has moved (a parameter / sub-module was registered
    # anywhere, Module.to() / .cuda() ran through ConvParam._apply, bump_epoch() was called) and every FULL_CHECK_EVERY-th call anyway:
    # the module tree is walked again and identity + storage pointer of every parameter compared as well -- that catches what no hook
    # sees (direct writes into Module._parameters, stateless swaps, __delattr__).
    def _walk(self):
        plist = list(self._params())
        return (_EPOCH[0], plist, tuple((id(p), p.data_ptr()) for p in plist))

    def _fingerprint(self):
        if self._params is None:
            return None
        self._calls += 1
        if self._plist is None or self._plist[0] != _EPOCH[0] or self._calls % FULL_CHECK_EVERY == 0:
            self._plist = self._walk()
        return (self._plist[2], tuple(map(_VERSIONS, self._plist[1])))

    def _guard_flag(self, device) -> Optional[torch.Tensor]:
        if self.check == "off" or not (ops._f16_family() and ops.range_guard_enabled()):
            return None
        return ops.range_flag(device)

    def _capture(self, x: torch.Tensor, args: tuple):
        static_x = x.clone()
        flag = self._guard_flag(x.device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):              # packs weights, sets kernel attributes, fills the allocator pool
                self.fn(static_x, *args)
        torch.cuda.current_stream().wait_stream(side)
        # (thread_local: launches and device allocations of OTHER host threads -- each with its own graphs, see auto_forward -- do not
        # invalidate this capture; what HIP does not tolerate beside a capture -- another capture, a pinned allocation, a graph's destruction --
        # is serialised by the lock, which is held from the pinned allocation to the end of the capture)
        with _CAPTURE_LOCK:
            _sweep_registry()
            _release()
            pinned = torch.zeros(1, dtype=torch.int32).pin_memory() if flag is not None else None
            graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
                if flag is not None:
                    flag.zero_()
                out = self.fn(static_x, *args)
                out = out if isinstance(out, tuple) else (out,)
                if flag is not None:
                    for o in out:
                        ops.poison_on_flag(flag, o)
                    if pinned is not None and self.check == "sync":
                        pinned.copy_(flag, non_blocking=True)  # the flag travels to pinned memory as the graph's last node: the sync check
                                                               # waits for an event and reads host memory (no separate device -> host read)
        return graph, static_x, out, flag, pinned

    def poll(self) -> None:
        """check="deferred": wait for the last replay's flag copy and raise RangeOverflow if it was up."""
        if self._pending is None:
            return
        pinned, ev = self._pending
        self._pending = None
        ev.synchronize()
        if int(pinned.item()):
            raise RangeOverflow("the previous graph replay staged an operand outside fp16's range: its outputs are NaN-filled; "
                                "run that input through the eager forward (which repeats it with the fp32 kernels)")

    def __call__(self, x: torch.Tensor, *args):
        """Returns the graph's OUTPUT BUFFERS (overwritten by the next call) -- clone what must outlive it -- or, with ``fresh``, clones."""
        if self.check == "deferred":
            self.poll()
        stamp = self._fingerprint()
        if stamp != self._stamp:
            _release(self._graphs)                  # parameters changed: the packed weights baked into the launches are stale
            self._seen.clear()
            self._stamp = stamp
        key = (tuple(x.shape), x.device.index, args, _env_stamp())
        hit = self._graphs.get(key)
        if hit is None:
            if self.auto_after > 0:
                seen = self._seen.get(key, 0)
                if seen < self.auto_after:
                    if len(self._seen) > 4096:
                        self._seen.clear()
                    self._seen[key] = seen + 1
                    return self.fn(x, *args)
            with torch.cuda.device(x.device):
                hit = self._graphs[key] = self._capture(x, args)
            self._seen.pop(key, None)
            while self.max_graphs > 0 and len(self._graphs) > self.max_graphs:
                with _CAPTURE_LOCK:
                    self._graphs.popitem(last=False)
        elif self.max_graphs > 0:
            self._graphs.move_to_end(key)
        graph, static_x, out, flag, pinned = hit
        static_x.copy_(x)
        graph.replay()
        self.replays += 1
        done = None
        if flag is not None and self.check == "sync":
            done = torch.cuda.Event()
            done.record()                                    # behind the graph (whose last node copied the flag to pinned memory)
        if self.fresh:                                       # fresh tensors, ONE copy launch (enqueued behind the replay; the host does not wait for it)
            ret = tuple(torch.empty_like(o) for o in out)
            torch._foreach_copy_(list(ret), list(out))
        else:
            ret = out
        if flag is not None:
            if self.check == "deferred":
                pinned.copy_(flag, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pending = (pinned, ev)
                return ret if len(ret) > 1 else ret[0]
            done.synchronize()
            if int(pinned[0]):
                warnings.warn("VIRNet HIP path: an activation left fp16's range in a replayed graph; this input was repeated eagerly "
                              "with the fp32 kernels", RuntimeWarning, stacklevel=2)
                self.reruns += 1
                from . import engine
                with torch.no_grad(), torch.cuda.device(x.device), ops.forward_scope(form=engine.FP32_FORM):
                    res = self.fn(static_x, *args)
                for o, r in zip(ret, res if isinstance(res, tuple) else (res,)):
                    o.copy_(r)
        return ret if len(ret) > 1 else ret[0]

    def reset(self) -> None:
        """Drop captured graphs (after a parameter write through ``.data``, which the version check cannot see)."""
        _release(self._graphs)
        self._seen.clear()
        self._pending = None

    def __del__(self):
        # (no lock, no HIP call here: this may run inside a garbage-collection pass in the middle of a capture)
        try:
            _GRAVEYARD.extend(self._graphs.values())      # (whole entries: the graph AND the buffers of its private pool)
        except Exception:          # noqa: BLE001  (interpreter shutdown)
            pass


# ----------------------------------------------------------------------------------------------------------------------------------
# net(x) itself at replay latency.  The reference scripts call the module one image at a time
# (scripts/testing_demo.py:87-93, scripts/denoising_virnet_syn.py:133-134): ~45 launches whose host enqueue costs as much as the kernels.
# The modules' inference forward therefore goes through auto_forward: per (module, host thread) a GraphedForward in auto mode -- the
# first AUTO_AFTER calls of a (shape, device, args) key run eagerly, the next one captures, later ones replay; outputs are fresh tensors;
# the range guard is the graph's sync check (flag read after the replay, fp32 repeat of that input); parameters are fingerprinted on
# every call.  Bypassed (plain eager forward): VIRNET_AUTOGRAPH=0, more than VIRNET_AUTOGRAPH_MAX_PIXELS output pixels per call (default
# 2^19: larger calls are not launch-bound and their graphs would pin GBs of intermediates), the deferred guard mode, a launch timer, a
# capture already running (the caller's own graph), inputs the eager path is going to reject anyway.
AUTO_AFTER = 2


class no_autograph:
    """``with graph.no_autograph():`` -- the modules' forwards run eagerly inside the block (the calling thread only)."""

    def __enter__(self):
        nat.tls.autograph_off = getattr(nat.tls, "autograph_off", 0) + 1
        return self

    def __exit__(self, *exc):
        nat.tls.autograph_off -= 1
        return False


# Every automatic GraphedForward is also held by this process-wide registry, so that its graphs are never destroyed as a side effect of a
# host thread ending (its threading.local goes away inside the dying thread) or of a module being collected (whichever thread runs the GC):
# a hipGraph and its private pool torn down while ANOTHER thread is capturing aborts the process (seen once in ~3 runs of the two-thread test).
# Entries of dead threads / dead modules are swept -- graphs dropped -- by the next capture, under the capture lock.
_REGISTRY: list = []          # (weakref to the module, the owning threading.Thread, the GraphedForward)
_REGISTRY_LOCK = threading.Lock()


def _sweep_registry() -> None:
    """Called with _CAPTURE_LOCK held: drop the graphs of owners that are gone."""
    with _REGISTRY_LOCK:
        dead = [e for e in _REGISTRY if e[0]() is None or not e[1].is_alive()]
        for e in dead:
            _REGISTRY.remove(e)
    for _, _, gf in dead:
        gf.reset()


def _auto_state(module, fn) -> GraphedForward:
    per_thread = getattr(nat.tls, "autograph", None)
    if per_thread is None:
        per_thread = nat.tls.autograph = weakref.WeakKeyDictionary()
    gf = per_thread.get(module)
    if gf is None:
        ref = weakref.ref(module)                      # (no strong reference from the thread's table back to the module)
        gf = per_thread[module] = GraphedForward(lambda x, *a: fn(ref(), x, *a), warmup=1, params=lambda: ref().parameters(),
                                                 check="sync", auto_after=AUTO_AFTER, fresh=True,
                                                 max_graphs=int(ops._env("VIRNET_AUTOGRAPH_MAX_GRAPHS", "8")))
        with _REGISTRY_LOCK:
            _REGISTRY.append((ref, threading.current_thread(), gf))
    return gf


def auto_forward(module, fn: Callable, x: torch.Tensor, *args, scale: int = 1):
    """``fn(module, x, *args)`` -- eagerly, or from the (module, thread)'s captured graph once the same call has been seen AUTO_AFTER
    times.  ``scale``: output pixels per input pixel side (the SISR forward's sf) for the size bound."""
    # MAIN THREAD ONLY.  Two host threads that each capture, replay and drop graphs bring this torch / HIP build down: a captured graph destroyed
    # from another thread than its capturer's fails ~CUDAGraph's generator-state check, a pinned allocation or a graph's destruction beside
    # another thread's capture invalidates that capture -- and both end in an exception thrown from a destructor, i.e. process abort
    # (tools/probes/capture_concurrency.py; it aborted one full-suite run in three).  Worker threads therefore run the eager forward; the
    # capture lock and the graveyard below keep the main thread's graphs away from what the other threads do.
    if (threading.current_thread() is not threading.main_thread()
            or getattr(nat.tls, "autograph_off", 0) or ops._env("VIRNET_AUTOGRAPH", "1") == "0" or ops._TIMER is not None
            or not isinstance(x, torch.Tensor) or not x.is_cuda or x.dim() != 4 or x.dtype != torch.float32
            or x.shape[0] * x.shape[2] * x.shape[3] * scale * scale > int(ops._env("VIRNET_AUTOGRAPH_MAX_PIXELS", str(1 << 19)))
            or ops._env("VIRNET_GUARD_CHECK", "sync") != "sync" or torch.cuda.is_current_stream_capturing()):
        return fn(module, x, *args)
    return _auto_state(module, fn)(x, *args)


def auto_stats(module) -> dict:
    """Replays / captured graphs / fp32 repeats of the calling thread's automatic graph of ``module`` (tests, tools/bench_latency.py)."""
    per_thread = getattr(nat.tls, "autograph", None)
    gf = None if per_thread is None else per_thread.get(module)
    return {"replays": 0, "graphs": 0, "reruns": 0} if gf is None else {"replays": gf.replays, "graphs": len(gf._graphs), "reruns": gf.reruns}
