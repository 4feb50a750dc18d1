#!/usr/bin/env python
"""Micro-benchmark of the weight-gradient path (re-layout passes + GEMM + reduce) on the denoise-syn shapes at the training batch
(32 x 128x128).  VIRNET_WGRAD_FORM=f32 times the round-1 fp32 kernel, VIRNET_CONV_FORM=bf16 the one-product variant, BENCH_ZEROS=1 the
same launches on all-zero operands (power probe: nothing toggles in the multipliers)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virnet_amd import ops  # noqa: E402

for name, (n, h, w, c) in {"l0": (32, 128, 128, 96), "l1": (32, 64, 64, 192), "l2": (32, 32, 32, 288)}.items():
    x = torch.rand(n, h, w, c, device="cuda") - 0.5
    dy = torch.rand(n, h, w, c, device="cuda") - 0.5
    if os.environ.get("BENCH_ZEROS") == "1":                # power probe: all-zero operands toggle nothing in the matrix pipe
        x.zero_(); dy.zero_()
    for _ in range(3):
        ops.conv_wgrad(x, dy, (c, c, 3, 3), in_slope=0.2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(15):
        e0.record(); ops.conv_wgrad(x, dy, (c, c, 3, 3), in_slope=0.2); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    fl = 2.0 * n * h * w * c * c * 9
    print(f"{os.path.basename(os.environ.get('VIRNET_HIP_LIB', 'default')):28s} wgrad {name}: {ts[len(ts) // 2]:.3f} ms  {fl / ts[len(ts) // 2] / 1e9:.1f} TFLOP/s", flush=True)
