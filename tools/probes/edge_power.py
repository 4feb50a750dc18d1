"""Socket power and shader clock while ONE edge launch (entry / exit / fill) runs back to back for a few seconds: is the launch capped, and
which clock does the power management give a launch that waits on memory?  python tools/probes/edge_power.py [--seconds 3]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import PowerSampler  # noqa: E402
from virnet_amd import ops, _native as nat  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=3.0)
a = ap.parse_args()
dev = "cuda"
N = 32
torch.manual_seed(0)
x_in = torch.rand(N, 3, 256, 256, device=dev)
sig = torch.rand(N, 1, 256, 256, device=dev) + 0.01
big = torch.rand(N, 256, 256, 96, device=dev) - 0.5
big2 = torch.empty_like(big)
head = ConvParam(4, 96, 3).cuda(); tail = ConvParam(96, 3, 3).cuda()
with ops.forward_scope():
    cases = [("fill 805 MB", lambda: big2.fill_(1.0), 0.805), ("copy 805 MB", lambda: big2.copy_(big), 1.611),
             ("entry head 4->96", lambda: ops.conv_entry(x_in, head.packed(), 256, 256, map_=sig, map_sqrt=True), 0.805),
             ("exit tail 96->3 + x_in", lambda: ops.conv_f16_nchw(big, tail.packed(), (256, 256), op=nat.NCHW_ADD, res=x_in), 0.856)]
    with PowerSampler(0, period=0.01) as ps:
        time.sleep(0.8)
    idle = ps.summary(skip_s=0.2)
    print(f"idle {idle['socket_w_mean']:.0f} W")
    for name, fn, gb in cases:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n = 0
        with PowerSampler(0, period=0.01) as ps:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < a.seconds:
                for _ in range(50):
                    fn()
                torch.cuda.synchronize()
                n += 50
            dt = time.perf_counter() - t0
        s = ps.summary(skip_s=0.3)
        ms = dt / n * 1e3
        print(f"{name:26s} {ms:7.3f} ms per launch  {gb / ms:5.2f} TB/s  socket {s['socket_w_mean']:7.1f} W  sclk {s['sclk_mhz_mean']:7.1f} MHz (min {s['sclk_mhz_min']:.0f})  J per launch {s['socket_w_mean'] * ms / 1e3:.3f}", flush=True)
