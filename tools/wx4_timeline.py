#!/usr/bin/env python
"""Where a workgroup of conv_wx4_kernel spends its time (a -DVIRNET_F16_TIMING build: tools/build_variant.sh timing -DVIRNET_F16_TIMING):
    VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_timing.so python tools/wx4_timeline.py [--shape l0] [--mode pre]
Wave 0 stamps s_memtime at start / after the prologue / after the K loop / at exit; waves 0 and 1 also sum, per stage type ji = 0,1,2,
the cycles until they reach the end-of-stage waits, the cycles in `s_waitcnt vmcnt(0)` and the cycles in the barrier."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VIRNET_CONV_FORM"] = "wx4"
from virnet_amd import _native as nat, ops  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402
from bench_conv import SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="l0")
    ap.add_argument("--mode", default="pre")
    ap.add_argument("--loaded", type=int, default=0, help="launches in front of the stamped one, stamps on (the socket at its power cap: ~300)")
    args = ap.parse_args()
    n, h, w, c = SHAPES[args.shape]
    lib = nat.load()
    lib.virnet_debug_timing_buffer.argtypes = [C.c_void_p]
    cp = ConvParam(c, c, 3).cuda()
    x = torch.rand(n, h, w, c, device="cuda") - 0.5
    res = torch.rand(n, h, w, c, device="cuda") - 0.5
    kw = {"res": dict(res=res, want_raw=True), "pre": dict(in_slope=0.2, want_raw=False, want_act=True)}[args.mode]
    pw = cp.packed()
    ntiles = n * ((h + 15) // 16) * ((w + 31) // 32)
    ncb = (c + 95) // 96
    nwg = ntiles * ncb
    log = torch.zeros(nwg * 8 + (nwg + 64) * 8 * 32 + 4096, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.conv_mfma(x, pw, **kw)
    torch.cuda.synchronize()
    lib.virnet_debug_timing_buffer(log.data_ptr())
    for _ in range(args.loaded):
        ops.conv_mfma(x, pw, **kw)
    ops.conv_mfma(x, pw, **kw)
    torch.cuda.synchronize()
    lib.virnet_debug_timing_buffer(None)
    t = log.cpu().numpy()
    st = t[:nwg * 8].reshape(-1, 8)
    st = st[st[:, 0] != 0]
    pro, kl, ep = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
    nst = (c // 16)
    print(f"{len(st)} workgroups; per workgroup (median cycles): prologue {np.median(pro):.0f}, K loop {np.median(kl):.0f} "
          f"({np.median(kl) / (3 * nst):.0f} per stage), epilogue {np.median(ep):.0f}")
    if st[:, 4].any():
        print(f"prologue split (median cycles after the start stamp): requests issued {np.median(st[:, 4] - st[:, 0]):.0f}, all arrived + first position staged "
              f"{np.median(st[:, 5] - st[:, 0]):.0f}, V written + barrier {np.median(pro):.0f}")
    if st[:, 6].any():
        print(f"epilogue split (median cycles after the K loop): first exchange written + barrier {np.median(st[:, 6] - st[:, 2]):.0f}, "
              f"slab 0 stored {np.median(st[:, 7] - st[:, 2]):.0f}, exit {np.median(ep):.0f}")
    # are the CUs' epilogues in step?  K-loop end times modulo the median tile period, 8 bins (flat = desynchronised)
    period = float(np.median(st[:, 3] - st[:, 0]))
    ph = ((st[:, 2] - st[:, 0].min()) % period) / period
    hist = np.histogram(ph, bins=8, range=(0, 1))[0]
    print(f"tile period {period:.0f} cycles; K-loop end phase histogram (8 bins): {hist.tolist()}; span of all tiles {st[:, 3].max() - st[:, 0].min()} cycles")
    acc = t[nwg * 8:nwg * 8 + nwg * 8 * 32].reshape(-1, 8, 32).astype(np.float64) / nst
    acc = acc[acc[:, 0, 0] != 0]
    print("cycles per MFMA group (3 MFMAs + the slots in front of them), g = 0..8, then tail + waits + barrier; median over workgroups")
    for wv in (0, 4, 1, 5):
        for ji in range(3):
            v = [np.median(acc[:, wv, ji * 10 + g]) for g in range(10)]
            print(f"  wave {wv} stage {ji}: " + " ".join(f"{x:5.0f}" for x in v[:9]) + f"  | {v[9]:5.0f}   sum {sum(v):6.0f}")


if __name__ == "__main__":
    main()
