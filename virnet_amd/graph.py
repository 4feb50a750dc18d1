"""hipGraph replay of a whole forward for the launch-bound small-batch path.

The reference scripts run ONE image per call (scripts/denoising_virnet_syn.py:133-134, scripts/testing_demo.py:87-93): ~45 kernel
launches whose host cost (~3 ms of Python + ctypes per forward) exceeds the GPU time below ~256x256.  Capturing the launches once
per input shape and replaying them removes that cost.  torch supplies the capture machinery
(`torch.cuda.CUDAGraph` == hipGraph on ROCm); every kernel in it is ours, launched on the capturing stream through the C ABI.
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch


class GraphedForward:
    """Callable that replays `fn(static_input, *args)` from a captured graph; one graph per (shape, args)."""

    def __init__(self, fn: Callable, warmup: int = 2):
        self.fn, self.warmup = fn, warmup
        self._graphs: Dict[Tuple, Tuple[torch.cuda.CUDAGraph, torch.Tensor, tuple]] = {}

    def _capture(self, x: torch.Tensor, args: tuple):
        static_x = x.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):              # packs weights, sets kernel attributes, fills the allocator pool
                self.fn(static_x, *args)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            out = self.fn(static_x, *args)
        return graph, static_x, out if isinstance(out, tuple) else (out,)

    def __call__(self, x: torch.Tensor, *args):
        """Returns the graph's OUTPUT BUFFERS (overwritten by the next call): clone what must outlive it."""
        key = (tuple(x.shape), x.device.index, args)
        if key not in self._graphs:
            self._graphs[key] = self._capture(x, args)
        graph, static_x, out = self._graphs[key]
        static_x.copy_(x)
        graph.replay()
        return out if len(out) > 1 else out[0]

    def reset(self) -> None:
        """Drop captured graphs (call after changing parameters: packed weights are baked into the captured launches)."""
        self._graphs.clear()
