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

import warnings
from typing import Callable, Dict, Iterable, Optional, Tuple

import torch

from . import ops


# Global registration epoch: bumped whenever ANY module registers a parameter or a sub-module (nn.Module.__setattr__ with a Parameter /
# Module, register_parameter, add_module).  A GraphedForward walks its module tree again only when the epoch has moved; between such
# events the parameter OBJECTS are the cached ones and the fingerprint is their (storage pointer, version) pairs.
_EPOCH = [0]


def _bump(*_a, **_k):
    _EPOCH[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_bump)
torch.nn.modules.module.register_module_module_registration_hook(_bump)


class RangeOverflow(RuntimeError):
    """Raised by a ``check="deferred"`` GraphedForward when the PREVIOUS replay staged an operand outside fp16's range."""


class GraphedForward:
    """Callable that replays `fn(static_input, *args)` from a captured graph; one graph per (shape, args)."""

    def __init__(self, fn: Callable, warmup: int = 2, params: Optional[Callable[[], Iterable[torch.Tensor]]] = None,
                 check: str = "sync"):
        if check not in ("sync", "deferred", "off"):
            raise ValueError(f"check={check!r}: expected 'sync', 'deferred' or 'off'")
        self.fn, self.warmup, self.check = fn, warmup, check
        self._params = params
        self._stamp = None
        self._plist = None             # (registration epoch, parameter objects) -- see _EPOCH
        self._graphs: Dict[Tuple, Tuple] = {}
        self._pending = None           # (pinned flag copy, event) of the last deferred replay
        self.reruns = 0                # replays repeated with the fp32 kernels (check="sync")

    # ---- parameter fingerprint: one hash over every parameter's (storage pointer, version) -- a swapped Parameter, a re-assigned
    # sub-module, load_state_dict, an optimizer step and .to() all move it (the round-4 form summed the versions and looked
    # at the first pointer only, so a fresh tensor with the same version in a later slot went unnoticed)
    def _fingerprint(self):
        if self._params is None:
            return None
        if self._plist is None or self._plist[0] != _EPOCH[0]:
            self._plist = (_EPOCH[0], list(self._params()))
        return hash(tuple((p.data_ptr(), p._version) for p in self._plist[1]))

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
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            if flag is not None:
                flag.zero_()
            out = self.fn(static_x, *args)
            out = out if isinstance(out, tuple) else (out,)
            if flag is not None:
                for o in out:
                    ops.poison_on_flag(flag, o)
        pinned = torch.zeros(1, dtype=torch.int32).pin_memory() if flag is not None else None
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
        """Returns the graph's OUTPUT BUFFERS (overwritten by the next call): clone what must outlive it."""
        if self.check == "deferred":
            self.poll()
        stamp = self._fingerprint()
        if stamp != self._stamp:
            self._graphs.clear()                      # parameters changed: the packed weights baked into the launches are stale
            self._stamp = stamp
        key = (tuple(x.shape), x.device.index, args)
        if key not in self._graphs:
            self._graphs[key] = self._capture(x, args)
        graph, static_x, out, flag, pinned = self._graphs[key]
        static_x.copy_(x)
        graph.replay()
        if flag is not None:
            if self.check == "deferred":
                pinned.copy_(flag, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pending = (pinned, ev)
            elif bool(flag.item()):
                warnings.warn("VIRNet HIP path: an activation left fp16's range in a replayed graph; this input was repeated eagerly "
                              "with the fp32 kernels", RuntimeWarning, stacklevel=2)
                self.reruns += 1
                from . import engine
                with torch.no_grad(), ops.forward_scope(form=engine.FP32_FORM):
                    res = self.fn(static_x, *args)
                for o, r in zip(out, res if isinstance(res, tuple) else (res,)):
                    o.copy_(r)
        return out if len(out) > 1 else out[0]

    def reset(self) -> None:
        """Drop captured graphs (after a parameter write through ``.data``, which the version check cannot see)."""
        self._graphs.clear()
        self._pending = None
