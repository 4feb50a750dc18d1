#!/usr/bin/env python
"""Per-phase cycle counts of the Winograd kernel (needs a -DWINO_TIMING variant build selected by VIRNET_HIP_LIB).

    VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_TIMING.so python tools/wino_timing.py [shape]
Phases per chunk (wave 0 of each workgroup): t0 = barrier exit -> after MFMA 0..3 + fragment/DMA/transform-read pieces,
t1 = MFMA 4..7 + transform writes, t2 = MFMA 8,9, t3 = pixel stores + MFMA 10,11 + fragment reads, t4 = MFMA 12..15, t5 = barrier."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virnet_amd import ops  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402
from bench_conv import SHAPES  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "l1"
n, h, w, c = SHAPES[name]
cp = ConvParam(c, c, 3).cuda()
x = torch.rand(n, h, w, c, device="cuda") - 0.5
pw = cp.packed()
nwg = 8 * 200000
tbuf = torch.zeros(nwg // 8 * 16 * 2, dtype=torch.float32, device="cuda")
for _ in range(3):
    tbuf.zero_()
    ops.conv_mfma(x, pw, in_slope=0.2, want_raw=False, want_act=True, add=tbuf)
torch.cuda.synchronize()
t = tbuf.view(torch.int64).view(-1, 16).cpu()
t = t[t[:, 9] > 0]
nch = c // 4
for role in (2, 1):
    for wv in range(1, 9):
        r = t[(t[:, 9] == role) & (t[:, 11] == wv)].double()
        if len(r) == 0:
            continue
        m = r.mean(0)
        per = [m[i].item() / nch for i in (5, 0, 1, 2, 3, 4, 10)]
        print(f"{name} CB={role} wave {wv - 1}: {len(r)} wgs; per chunk: barrier {per[0]:.0f} | q1+pieces {per[1]:.0f} | q2+xfw {per[2]:.0f} | "
              f"m8,9 {per[3]:.0f} | store+m10,11+frag {per[4]:.0f} | q4 {per[5]:.0f} | waitcnt {per[6]:.0f} | sum {sum(per):.0f} ; prologue {m[8].item():.0f} "
              f"loop {m[6].item():.0f} epilogue {m[7].item():.0f} | epilogue: last MFMAs+loads {m[12].item():.0f} transform+exchange write {m[13].item():.0f} barrier {m[14].item():.0f} combine+stores {m[15].item():.0f}")
