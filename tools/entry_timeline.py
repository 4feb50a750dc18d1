#!/usr/bin/env python
"""Where a persistent workgroup of conv_entry_kernel spends its cycles (a -DVIRNET_F16_TIMING build of the one unit):
    tools/build_one.sh etiming conv_entry -DVIRNET_F16_TIMING
    VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_etiming.so python tools/entry_timeline.py
Thread 0 accumulates s_memtime deltas per tile: gather + MFMAs | barrier + wait for the next tile's pixels + landing them | LDS turn-around +
store issue | closing barrier."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VIRNET_CONV_FORM", "wx4")
from virnet_amd import _native as nat, ops  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402

lib = nat.load()
lib.virnet_debug_entry_timing_buffer.argtypes = [C.c_void_p]
N = 32
x_in = torch.rand(N, 3, 256, 256, device="cuda")
sig = torch.rand(N, 1, 256, 256, device="cuda") + 0.01
head = ConvParam(4, 96, 3).cuda()
c1 = ConvParam(3, 64, 3).cuda()
with ops.forward_scope():
    for name, fn in (("head 4->96 (+sqrt map)", lambda: ops.conv_entry(x_in, head.packed(), 256, 256, map_=sig, map_sqrt=True)),
                     ("SNet 3->64 lrelu", lambda: ops.conv_entry(x_in, c1.packed(), 256, 256, want_act=True, slope=0.25))):
        log = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.virnet_debug_entry_timing_buffer(log.data_ptr())
        fn()
        torch.cuda.synchronize()
        lib.virnet_debug_entry_timing_buffer(None)
        st = log.cpu().numpy().reshape(-1, 8)
        st = st[st[:, 5] != 0]
        tiles = st[:, 5]
        per = lambda c: np.median(st[:, c] / tiles)
        life = st[:, 6] - st[:, 0]
        print(f"{name}: {len(st)} persistent workgroups x {np.median(tiles):.0f} tiles; median cycles per tile: gather + MFMA {per(1):.0f} | barrier + wait + land {per(2):.0f} | "
              f"turn-around + store issue {per(3):.0f} | closing barrier {per(4):.0f} | workgroup lifetime / tiles {np.median(life / tiles):.0f}; launch span {(st[:, 6].max() - st[:, 0].min())} cycles")
