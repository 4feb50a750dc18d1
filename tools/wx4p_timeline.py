#!/usr/bin/env python
"""Where a persistent workgroup of conv_wx4p_kernel (csrc/conv_f16_wx4p.hip) spends its time, per item (a -DVIRNET_F16_TIMING build:
tools/build_variant.sh timing -DVIRNET_F16_TIMING):
    VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_timing.so python tools/wx4p_timeline.py [--shape l0] [--mode pre]
Thread 0 stamps s_memtime at an item's K-loop start, at the start of its last chunk, at the K loop's end and at the epilogue's end."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VIRNET_CONV_FORM"] = "wx4"
os.environ["VIRNET_WX4_PERSIST"] = "1"
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
    log = torch.zeros(256 * 64 * 4 + 4096, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.conv_mfma(x, pw, **kw)
    torch.cuda.synchronize()
    lib.virnet_debug_timing_buffer(log.data_ptr())
    for _ in range(args.loaded):
        ops.conv_mfma(x, pw, **kw)
    ops.conv_mfma(x, pw, **kw)
    torch.cuda.synchronize()
    lib.virnet_debug_timing_buffer(None)
    t = log.cpu().numpy()[:256 * 64 * 4].reshape(256, 64, 4)
    nst = 3 * (c // 16)
    per_wg = (t[:, :, 0] != 0).sum(1)
    print(f"{int((per_wg > 0).sum())} workgroups, {per_wg[per_wg > 0].min()}-{per_wg.max()} items each ({args.shape} {args.mode})")
    rows = t[t[:, :, 0] != 0]
    kl, lastc, ep = rows[:, 2] - rows[:, 0], rows[:, 2] - rows[:, 1], rows[:, 3] - rows[:, 2]
    print(f"per item (median cycles): K loop {np.median(kl):.0f} ({np.median(kl) / nst:.0f} per stage; last chunk {np.median(lastc) / 3:.0f} per stage, "
          f"the others {np.median(kl - lastc) / max(1, nst - 3):.0f}), epilogue {np.median(ep):.0f}, item {np.median(rows[:, 3] - rows[:, 0]):.0f}")
    # gap between items of one workgroup (epilogue end -> next K loop start) and the first item's start offset
    gaps = []
    for b in range(256):
        k = per_wg[b]
        if k > 1:
            gaps += list(t[b, 1:k, 0] - t[b, :k - 1, 3])
    if gaps:
        print(f"between items: {np.median(gaps):.0f} cycles")
    # a persistent workgroup's own span (one CU, one counter): the launch lasts as long as the slowest one
    span = np.array([t[b, per_wg[b] - 1, 3] - t[b, 0, 0] for b in range(256) if per_wg[b] > 0], dtype=np.float64)
    xcd = np.array([b & 7 for b in range(256) if per_wg[b] > 0])
    print(f"workgroup span (first K-loop start -> last epilogue end): min {span.min():.0f} median {np.median(span):.0f} max {span.max():.0f} "
          f"(max / median {span.max() / np.median(span):.3f}); sum of its items / span {np.median([ (t[b, :per_wg[b], 3] - t[b, :per_wg[b], 0]).sum() for b in range(256) if per_wg[b] > 0] / span):.3f}")
    print("per XCD median span: " + " ".join(f"{np.median(span[xcd == x]):.0f}" for x in range(8)))
    first = np.array([t[b, 0, 3] - t[b, 0, 0] for b in range(256) if per_wg[b] > 0])
    print(f"first item of a workgroup {np.median(first):.0f}, later items {np.median((rows[:, 3] - rows[:, 0])):.0f}")


if __name__ == "__main__":
    main()
