#!/usr/bin/env python
"""Edge kernels of the metric's step (entries, exits, transposed and stride-2 convs at 32 x 256^2) one by one, next to the stream rates
this box reaches on tensors of the same size (write-only fill, copy, read-only sum): python tools/probes/edge_time.py [--iters 30]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from virnet_amd import ops, _native as nat  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=30); ap.add_argument("--n", type=int, default=32)
ap.add_argument("--only", default="")
a = ap.parse_args()
N = a.n
torch.manual_seed(0)


def timeit(fn, iters=a.iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def row(name, fn, gb, note=""):
    if a.only and a.only not in name:
        return
    med, mn = timeit(fn)
    print(f"{name:44s} median {med:7.3f} ms  min {mn:7.3f} ms  {gb:6.3f} GB algorithmic = {gb / med:5.2f} TB/s  {note}", flush=True)


dev = "cuda"
big = torch.rand(N, 256, 256, 96, device=dev) - 0.5
big2 = torch.empty_like(big)
gb = big.numel() * 4 / 1e9
row("stream: fill (write-only) 96ch", lambda: big2.fill_(1.0), gb)
row("stream: copy 96ch (read + write)", lambda: big2.copy_(big), 2 * gb)
row("stream: sum (read-only) 96ch", lambda: big.sum(), gb)

x_in = torch.rand(N, 3, 256, 256, device=dev)
sig = torch.rand(N, 1, 256, 256, device=dev) + 0.01
head = ConvParam(4, 96, 3).cuda(); c1 = ConvParam(3, 64, 3).cuda()
with ops.forward_scope():
    row("entry head 4->96 (+sqrt map) -> raw", lambda: ops.conv_entry(x_in, head.packed(), 256, 256, map_=sig, map_sqrt=True), gb)
    row("entry SNet 3->64 -> lrelu", lambda: ops.conv_entry(x_in, c1.packed(), 256, 256, want_act=True, slope=0.25), gb * 64 / 96)
    tail = ConvParam(96, 3, 3).cuda(); last = ConvParam(64, 1, 3).cuda()
    s64 = torch.rand(N, 256, 256, 64, device=dev) - 0.5
    row("exit tail 96->3 + x_in", lambda: ops.conv_f16_nchw(big, tail.packed(), (256, 256), op=nat.NCHW_ADD, res=x_in), gb + 2 * N * 3 * 65536 * 4 / 1e9)
    row("exit SNet last 64->1 exp(clamp)", lambda: ops.conv_f16_nchw(s64, last.packed(), (256, 256), op=nat.NCHW_EXPCLAMP, clamp=(-23.0, 4.6)), gb * 64 / 96 + N * 65536 * 4 / 1e9)
    up1 = ConvParam(192, 96, 2, transposed=True, stride=2).cuda(); x1 = torch.rand(N, 128, 128, 192, device=dev) - 0.5
    row("convT 192->96 @128^2 + bridge", lambda: ops.conv_mfma(x1, up1.packed(), res=big, want_raw=True), x1.numel() * 4 / 1e9 + 2 * gb)
    up2 = ConvParam(288, 192, 2, transposed=True, stride=2).cuda(); x2 = torch.rand(N, 64, 64, 288, device=dev) - 0.5
    row("convT 288->192 @64^2 + bridge", lambda: ops.conv_mfma(x2, up2.packed(), res=x1, want_raw=True), x2.numel() * 4 / 1e9 + 2 * x1.numel() * 4 / 1e9)
    d1 = ConvParam(96, 192, 3, stride=2).cuda()
    row("s2 96->192 @256^2", lambda: ops.conv_mfma(big, d1.packed(), stride=2, want_raw=True), gb + x1.numel() * 4 / 1e9, f"{2 * N * 128 * 128 * 96 * 192 * 9 / 1e9:.1f} GFLOP")
    d2 = ConvParam(192, 288, 3, stride=2).cuda()
    row("s2 192->288 @128^2", lambda: ops.conv_mfma(x1, d2.packed(), stride=2, want_raw=True), (x1.numel() + x2.numel()) * 4 / 1e9, f"{2 * N * 64 * 64 * 192 * 288 * 9 / 1e9:.1f} GFLOP")
