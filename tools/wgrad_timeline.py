#!/usr/bin/env python
"""Where a conv_wgrad_f16 workgroup's cycles go, from a -DVIRNET_F16_TIMING build (tools/build_variant.sh timing -DVIRNET_F16_TIMING):

    VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_timing.so python tools/wgrad_timeline.py [--shape l0]

Wave 0 of every workgroup sums s_memtime cycles spent (a) waiting for operands + the step barrier, (b) issuing LDS-DMA requests,
(c) in the MFMA phase, and logs its start / end."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virnet_amd import _native as nat, ops  # noqa: E402

SHAPES = {"l0": (32, 128, 128, 96), "l1": (32, 64, 64, 192), "l2": (32, 32, 32, 288)}
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="l0")
args = ap.parse_args()
n, h, w, c = SHAPES[args.shape]
lib = nat.load()
lib.virnet_debug_wgrad_timing_buffer.argtypes = [C.c_void_p]
x = torch.rand(n, h, w, c, device="cuda") - 0.5
dy = torch.rand(n, h, w, c, device="cuda") - 0.5
for _ in range(3):
    ops.conv_wgrad(x, dy, (c, c, 3, 3), in_slope=0.2)
log = torch.zeros(4096, 8, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
lib.virnet_debug_wgrad_timing_buffer(log.data_ptr())
ops.conv_wgrad(x, dy, (c, c, 3, 3), in_slope=0.2)
torch.cuda.synchronize()
lib.virnet_debug_wgrad_timing_buffer(None)
t = log.cpu().numpy()
t = t[t[:, 0] != 0]
life = t[:, 4] - t[:, 0]
print(f"{args.shape}: {len(t)} workgroups, steps/run {t[:, 5].mean():.1f}; s_memtime ticks (100 MHz): life {life.mean():.0f} "
      f"(min {life.min()} max {life.max()}), wait+barrier {t[:, 1].mean():.0f}, issue {t[:, 2].mean():.0f}, compute {t[:, 3].mean():.0f}; "
      f"per step: wait {t[:, 1].sum() / t[:, 5].sum():.1f} issue {t[:, 2].sum() / t[:, 5].sum():.1f} compute {t[:, 3].sum() / t[:, 5].sum():.1f}")

import collections
pat = collections.Counter()
for row in t:
    simds = [((int(row[6]) >> (4 * i)) & 0xf) - 1 for i in range(12) if (int(row[6]) >> (4 * i)) & 0xf]
    pat[tuple(simds)] += 1
print("SIMD of waves 0..5 (most common):", pat.most_common(6))
cu = collections.defaultdict(list)
for row in t:
    hw = int(row[7]) & 0xffff
    key = (int(row[7]) >> 16, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)
    simds = [((int(row[6]) >> (4 * i)) & 0xf) - 1 for i in range(12) if (int(row[6]) >> (4 * i)) & 0xf]
    cu[key].append(simds)
occ = collections.Counter()
for k, v in cu.items():
    c = collections.Counter(x for s in v for x in s)
    occ[tuple(sorted(c.values()))] += 1
print("waves per SIMD on a CU (sorted) -> #CUs:", occ.most_common(6))
