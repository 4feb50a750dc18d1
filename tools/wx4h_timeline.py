#!/usr/bin/env python
"""Where a workgroup of conv_wx4h_kernel spends its time and how the two workgroups of a CU overlap (a -DVIRNET_F16_TIMING build:
tools/build_variant.sh timing -DVIRNET_F16_TIMING):
    VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_timing.so python tools/wx4h_timeline.py [--shape l0] [--mode pre]
Wave 0 stamps s_memtime at start / after the prologue / after the K loop / first exchange / slab 0 stored / exit and its HW_ID / XCC_ID."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VIRNET_CONV_FORM"] = "wx4"
os.environ.setdefault("VIRNET_WX4_ROWS", "8")
from virnet_amd import _native as nat, ops  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402
from bench_conv import SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="l0")
    ap.add_argument("--mode", default="pre")
    args = ap.parse_args()
    rows = int(os.environ["VIRNET_WX4_ROWS"])
    n, h, w, c = SHAPES[args.shape]
    lib = nat.load()
    lib.virnet_debug_timing_buffer.argtypes = [C.c_void_p]
    cp = ConvParam(c, c, 3).cuda()
    x = torch.rand(n, h, w, c, device="cuda") - 0.5
    res = torch.rand(n, h, w, c, device="cuda") - 0.5
    kw = {"res": dict(res=res, want_raw=True), "pre": dict(in_slope=0.2, want_raw=False, want_act=True)}[args.mode]
    pw = cp.packed()
    ntiles = n * ((h + rows - 1) // rows) * ((w + 31) // 32)
    ncb = (c + 95) // 96
    nwg = ntiles * ncb
    log = torch.zeros((nwg + 64) * 8 + (nwg + 64) * 8 * 32 + 4096, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.conv_mfma(x, pw, **kw)
    torch.cuda.synchronize()
    lib.virnet_debug_timing_buffer(log.data_ptr())
    ops.conv_mfma(x, pw, **kw)
    torch.cuda.synchronize()
    lib.virnet_debug_timing_buffer(None)
    t = log.cpu().numpy()
    st = t[:(nwg + 64) * 8].reshape(-1, 8)
    st = st[st[:, 0] != 0]
    pro, kl, ep = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
    nst = c // 16
    print(f"{len(st)} workgroups; per workgroup (median cycles of the 100 MHz... s_memtime clock): prologue {np.median(pro):.0f}, K loop {np.median(kl):.0f} "
          f"({np.median(kl) / (3 * nst):.0f} per stage), epilogue {np.median(ep):.0f}; tile {np.median(st[:, 3] - st[:, 0]):.0f}")
    print(f"span of the launch {st[:, 3].max() - st[:, 0].min()} ; sum of tiles / (CUs x 2 slots) = {np.sum(st[:, 3] - st[:, 0]) / 512:.0f}")
    if rows != 8:
        return
    hw, xcc = st[:, 4], st[:, 5] & 0xF
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    # per CU: fraction of time with 0 / 1 / 2 workgroups inside their K loop, and the phase offset of co-resident K loops
    tot = np.zeros(3)
    offs = []
    for cid in np.unique(cu):
        s = st[cu == cid]
        ev = sorted([(r[1], 1) for r in s] + [(r[2], -1) for r in s])
        t0, t1 = s[:, 0].min(), s[:, 3].max()
        cur, last = 0, t0
        for tt, d in ev:
            tot[min(cur, 2)] += tt - last
            last = tt
            cur += d
        tot[0] += t1 - last
        # offset of each K-loop start against the K loop that is running on the CU at that moment, in tile periods
        for r in s:
            others = s[(s[:, 1] < r[1]) & (s[:, 2] > r[1])]
            if len(others):
                offs.append((r[1] - others[0][1]) / float(others[0][2] - others[0][1]))
    tot /= tot.sum()
    print(f"CUs seen {len(np.unique(cu))}; share of CU time with 0 / 1 / 2 workgroups in their K loop: {tot[0]:.3f} / {tot[1]:.3f} / {tot[2]:.3f}")
    if offs:
        print("K-loop start offset against the co-resident K loop (fraction of its length), histogram 10 bins:",
              np.histogram(offs, bins=10, range=(0, 1))[0].tolist())


if __name__ == "__main__":
    main()
