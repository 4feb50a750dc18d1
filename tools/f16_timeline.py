#!/usr/bin/env python
"""Per-workgroup timeline of conv_f16_kernel from a -DVIRNET_F16_TIMING build (tools/build_variant.sh timing -DVIRNET_F16_TIMING):

    VIRNET_HIP_LIB=$PWD/virnet_amd/lib/libvirnet_hip_timing.so python tools/f16_timeline.py [--shape l0] [--mode pre]

Wave 0 of every workgroup logs s_memtime (shader cycles) at start, after the prologue, after the K loop and at exit, plus the CU it
ran on; the summary shows where a workgroup's life goes and how the two workgroups of one CU overlap."""
import argparse
import collections
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from virnet_amd import _native as nat, ops  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402
from bench_conv import SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="l0")
    ap.add_argument("--mode", default="pre")
    args = ap.parse_args()
    n, h, w, c = SHAPES[args.shape]
    lib = nat.load()
    lib.virnet_debug_timing_buffer.argtypes = [C.c_void_p]
    cp = ConvParam(c, c, 3).cuda()
    x = torch.rand(n, h, w, c, device="cuda") - 0.5
    res = torch.rand(n, h, w, c, device="cuda") - 0.5
    kw = {"res": dict(res=res, want_raw=True), "pre": dict(in_slope=0.2, want_raw=False, want_act=True)}[args.mode]
    pw = cp.packed()
    nwg = 8 * ((n * ((h + 7) // 8) * ((w + 31) // 32) + 7) // 8) * max(1, c // 96 if c % 96 == 0 else c // 64 if c % 64 == 0 else c // 32)
    log = torch.zeros(nwg * 2 + 1024, 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.conv_mfma(x, pw, **kw)
    torch.cuda.synchronize()
    lib.virnet_debug_timing_buffer(log.data_ptr())
    ops.conv_mfma(x, pw, **kw)
    torch.cuda.synchronize()
    lib.virnet_debug_timing_buffer(None)
    t = log.cpu().numpy()
    blk = np.nonzero(t[:, 0] != 0)[0]
    t = t[blk]
    hw, xcc = t[:, 4], t[:, 5] & 0xF
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    # s_memtime counters are per XCD: make every time relative to the earliest start on the same XCD
    t0 = np.zeros(len(t), dtype=np.int64)
    for xc in set(xcc.tolist()):
        t0[xcc == xc] = t[xcc == xc, 0].min()
    start, pro, kl, end = (t[:, i] - t0 for i in range(4))
    print(f"{len(t)} workgroups, kernel span {int(end.max())} cycles; distinct CUs {len(set(cu.tolist()))}")
    first = collections.defaultdict(list)
    for i in np.argsort(start):
        if len(first[int(cu[i])]) < 3:
            first[int(cu[i])].append(int(blk[i]))
    print("first three blocks on some CUs (cu id: blockIdx ...):", {hex(k): v for k, v in list(first.items())[:6]})
    second = sorted(v[1] for v in first.values() if len(v) > 1)
    print("blockIdx of the SECOND workgroup per CU: min %d max %d, below 256: %d of %d" % (second[0], second[-1], sum(b < 256 for b in second), len(second)))
    print("per workgroup (cycles, median / p10 / p90): prologue %d / %d / %d, K loop %d / %d / %d, epilogue %d / %d / %d, total %d" % (
        *np.percentile(pro - start, [50, 10, 90]), *np.percentile(kl - pro, [50, 10, 90]), *np.percentile(end - kl, [50, 10, 90]),
        np.median(end - start)))
    if t[:, 6].any():
        e6, e7 = t[:, 6] - t0, t[:, 7] - t0
        print("epilogue split (median cycles): K-loop end -> loads issued %d, -> first slab stored %d, -> exit %d" % (
            np.median(e6 - kl), np.median(e7 - kl), np.median(end - kl)))
    # per CU: fraction of the kernel span with 0 / 1 / 2 workgroups inside their K loop
    by = collections.defaultdict(list)
    for i in range(len(t)):
        by[int(cu[i])].append((int(pro[i]), int(kl[i]), int(start[i]), int(end[i])))
    span = int(end.max())
    occ = np.zeros(3)
    gaps = []
    for k, lst in by.items():
        ev = []
        for p, q, s_, e_ in lst:
            ev += [(p, 1), (q, -1)]
        ev.sort()
        cur, last = 0, 0
        for tt, d in ev:
            occ[min(cur, 2)] += tt - last
            cur += d
            last = tt
        occ[0] += span - last
        ends = sorted(e_ for *_, e_ in lst)
        starts = sorted(s_ for _, _, s_, _ in lst)
        # a slot's turnover: time from an exit to the next start on this CU
        for e_ in ends:
            nxt = [s_ for s_ in starts if s_ >= e_]
            if nxt:
                gaps.append(nxt[0] - e_)
    occ /= occ.sum()
    print("CU time with 0 / 1 / 2+ workgroups in their K loop: %.1f%% / %.1f%% / %.1f%%" % tuple(100 * occ))
    print("exit -> next start on the same CU (cycles): median %d, p90 %d" % (np.median(gaps), np.percentile(gaps, 90)))
    wgs = sorted(by.values(), key=len)
    print("workgroups per CU: min %d max %d" % (len(wgs[0]), len(wgs[-1])))
    one = sorted(by[int(cu[0])], key=lambda r: r[2])[:8]
    print("first CU timeline (start, K-loop begin, K-loop end, exit):")
    for p, q, s_, e_ in one:
        print("   %8d %8d %8d %8d" % (s_, p, q, e_))


if __name__ == "__main__":
    main()
