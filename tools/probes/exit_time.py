"""Exit convs at the metric shape, taps-as-rows kernel (conv_exit) against conv_f16's planar form: python tools/probes/exit_time.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from virnet_amd import ops, _native as nat  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402
os.environ.setdefault("VIRNET_CONV_FORM", "wx4")
for (name, n, h, w, c, cout, op) in [("tail 96->3 +x_in", 32, 256, 256, 96, 3, nat.NCHW_ADD), ("SNet last 64->1 exp(clamp)", 32, 256, 256, 64, 1, nat.NCHW_EXPCLAMP),
                                      ("tail 128^2 x64", 64, 128, 128, 96, 3, nat.NCHW_ADD), ("tail one 256^2", 1, 256, 256, 96, 3, nat.NCHW_ADD)]:
    cp = ConvParam(c, cout, 3).cuda()
    x = torch.rand(n, h, w, c, device="cuda") - 0.5
    res = torch.rand(n, cout, h, w, device="cuda")
    pw = cp.packed()
    for form in ("rows", "f16"):
        os.environ["VIRNET_EXIT_FORM"] = form
        kw = dict(op=op, res=res if op == nat.NCHW_ADD else None, clamp=(-20.0, 4.0))
        for _ in range(3):
            ops.conv_f16_nchw(x, pw, (h, w), **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(20):
            e0.record(); ops.conv_f16_nchw(x, pw, (h, w), **kw); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        gb = (n * h * w * c * 4 + n * h * w * cout * 4 * (2 if op == nat.NCHW_ADD else 1)) / 1e9
        print(f"{name:28s} form {form:4s}: median {ts[10]:.3f} ms  min {ts[0]:.3f} ms   {gb / ts[10]:.2f} TB/s algorithmic", flush=True)
