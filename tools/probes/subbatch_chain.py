#!/usr/bin/env python
"""Level-0 res-block chain in sub-batches (round 5): the joule ledger says HBM + fabric traffic is ~21 % of a 96-channel launch's energy and
a launch whose tiles stay in the Infinity Cache runs 9 % faster -- so: the three res-blocks of level 0 (six 96->96 convs at 256^2) on 32
images, the batch cut into sub-batches of SB images that go through ALL six convs before the next sub-batch starts (intermediates are then
SB/32 x 805 MB and are re-used from cache; the allocator hands the same blocks to every sub-batch).

Round 6: ``--streams S`` -- the sub-batch chains go round-robin onto S HIP streams, so that the tail of one chain's launch (a sub-batch of
two images is ONE round of 256 workgroups; r05: 1 285 W, not capped -- the chip waits at every launch boundary) is filled by another chain's
workgroups.  Every stream's allocator pool re-uses its own blocks: working set = S x 3 tensors x SB x 25 MB.

    python tools/probes/subbatch_chain.py [--sb 32,16,8,4,2] [--streams 1,2,3] [--blocks 3] [--shape 32,256,256,96]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from virnet_amd import ops  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402
from bench import PowerSampler  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sb", default="32,16,8,4,2"); ap.add_argument("--blocks", type=int, default=3)
    ap.add_argument("--shape", default="32,256,256,96"); ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--streams", default="1")
    a = ap.parse_args()
    n, h, w, c = (int(v) for v in a.shape.split(","))
    torch.manual_seed(0)
    cps = [ConvParam(c, c, 3).cuda() for _ in range(2 * a.blocks)]
    pws = [cp.packed() for cp in cps]
    x = torch.rand(n, h, w, c, device="cuda") - 0.5

    def chain(xs):
        cur = xs
        for b in range(a.blocks):
            _, t = ops.conv_mfma(cur, pws[2 * b], in_slope=0.2, want_raw=False, want_act=True)
            cur, _ = ops.conv_mfma(t, pws[2 * b + 1], res=cur, want_raw=True)
        return cur

    ref = None
    for sb, ns in [(int(v), int(u)) for v in a.sb.split(",") for u in a.streams.split(",")]:
        if ns > 1 and sb >= n:
            continue
        streams = [torch.cuda.Stream() for _ in range(ns)] if ns > 1 else None

        def run():
            if streams is None:
                with ops.forward_scope():
                    return [chain(x[i:i + sb]) for i in range(0, n, sb)]
            outs = []
            cur = torch.cuda.current_stream()
            for s_ in streams:
                s_.wait_stream(cur)
            for k, i in enumerate(range(0, n, sb)):
                with torch.cuda.stream(streams[k % ns]), ops.forward_scope():
                    outs.append(chain(x[i:i + sb]))
            for s_ in streams:
                cur.wait_stream(s_)
            return outs
        for _ in range(3):
            out = run()
        torch.cuda.synchronize()
        with PowerSampler(0, period=0.01) as ps:
            t0 = time.perf_counter()
            for _ in range(a.iters):
                out = run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.iters * 1e3
        p = ps.summary(skip_s=0.1) or {}
        y = torch.cat(out)
        if ref is None:
            ref = y
        print(f"sb {sb:3d} streams {ns}: {dt:8.3f} ms per {n} images ({dt / (2 * a.blocks):.3f} ms per conv-equivalent)  {p.get('socket_w_mean')} W  sclk {p.get('sclk_mhz_mean')}  "
              f"max|diff| vs first {float((y - ref).abs().max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
