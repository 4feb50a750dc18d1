#!/usr/bin/env python
"""Where a workgroup of conv_exit_kernel spends its time (a -DVIRNET_F16_TIMING build of the one unit:
    tools/build_one.sh xtiming conv_exit -DVIRNET_F16_TIMING
    VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_xtiming.so python tools/exit_timeline.py
Thread 0 stamps s_memtime at entry / weights in LDS / its wave's blocks done / barrier passed / stores acknowledged, plus HW_ID, XCC_ID."""
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
lib.virnet_debug_exit_timing_buffer.argtypes = [C.c_void_p]
for name, n, h, w, c, cout, op in [("tail 96->3 +x_in", 32, 256, 256, 96, 3, nat.NCHW_ADD), ("SNet last 64->1", 32, 256, 256, 64, 1, nat.NCHW_EXPCLAMP)]:
    cp = ConvParam(c, cout, 3).cuda()
    x = torch.rand(n, h, w, c, device="cuda") - 0.5
    res = torch.rand(n, cout, h, w, device="cuda")
    pw = cp.packed()
    kw = dict(op=op, res=res if op == nat.NCHW_ADD else None, clamp=(-20.0, 4.0))
    nwg = n * (h // 8) * (w // 32)
    log = torch.zeros((nwg + 64) * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.conv_f16_nchw(x, pw, (h, w), **kw)
    torch.cuda.synchronize()
    lib.virnet_debug_exit_timing_buffer(log.data_ptr())
    ops.conv_f16_nchw(x, pw, (h, w), **kw)
    torch.cuda.synchronize()
    lib.virnet_debug_exit_timing_buffer(None)
    st = log.cpu().numpy().reshape(-1, 8)
    st = st[st[:, 0] != 0]
    d = [st[:, i + 1] - st[:, i] for i in range(4)]
    tot = st[:, 4] - st[:, 0]
    print(f"{name}: {len(st)} workgroups; median s_memtime ticks (100 MHz) per phase: weights {np.median(d[0]):.0f} | blocks {np.median(d[1]):.0f} | barrier {np.median(d[2]):.0f} | "
          f"shift-add + stores {np.median(d[3]):.0f} | workgroup {np.median(tot):.0f} (p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f})")
    span = st[:, 4].max() - st[:, 0].min()
    hw, xcc = st[:, 5], st[:, 6] & 0xF
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    ncu = len(np.unique(cu))
    print(f"   launch span {span} ticks = {span / 100:.1f} us; sum of workgroup lifetimes / (span x CUs seen {ncu}) = {tot.sum() / (span * ncu):.2f} workgroups resident per CU on average")
    # start-to-start interval of consecutive workgroups on one CU
    gaps = []
    for cid in np.unique(cu)[:64]:
        s = np.sort(st[cu == cid][:, 0])
        gaps += list(np.diff(s))
    print(f"   per CU: median interval between workgroup starts {np.median(gaps):.0f} ticks; workgroups per CU {len(st) / ncu:.1f}")
