#!/usr/bin/env python
"""Where knet_body_kernel's single workgroup spends its time (a -DVIRNET_F16_TIMING build of the one unit):
    tools/build_one.sh ktiming knet_body -DVIRNET_F16_TIMING
    VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_ktiming.so python tools/knet_timeline.py
Thread 0 accumulates s_memtime deltas: inside the stages (fragment reads + MFMAs + the next pieces' issue) | s_waitcnt + s_barrier at the stage
ends | everything between the convolutions (split to LDS, scale / bias, CALayer)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_net  # noqa: E402
from virnet_amd import _native as nat, engine  # noqa: E402
from virnet_amd.utils.synth import synth_images  # noqa: E402

dev = torch.device("cuda", 0)
net, sd = build_net(dev, "sisr")
net.load_state_dict(sd, strict=True)
net = net.to(dev).eval()
lib = nat.load()
lib.virnet_debug_knet_timing_buffer.argtypes = [C.c_void_p]
for n in (1, 16):
    x = synth_images(n, 3, 64, 64).to(dev)
    log = torch.zeros(64 * 8, dtype=torch.int64, device=dev)
    with torch.no_grad():
        for _ in range(3):
            engine.knet_forward(net.KNet, x)
        torch.cuda.synchronize()
        lib.virnet_debug_knet_timing_buffer(log.data_ptr())
        engine.knet_forward(net.KNet, x)
        torch.cuda.synchronize()
        lib.virnet_debug_knet_timing_buffer(None)
    st = log.cpu().numpy().reshape(-1, 8)[:n]
    tot = st[:, 4] - st[:, 0]
    nst = 24 * len(net.KNet.body)
    print(f"n={n}: ticks per workgroup: total {tot.mean():.0f}; inside the {nst} stages {st[:, 1].mean():.0f} ({st[:, 1].mean() / nst:.0f} per stage) | "
          f"wait + barrier {st[:, 2].mean():.0f} ({st[:, 2].mean() / nst:.0f} per stage) | between the convs {st[:, 3].mean():.0f} ({st[:, 3].mean() / len(net.KNet.body):.0f} per layer)")
