"""Transposed 2x2/s2 conv + bridge at the metric shapes, slab-group forms A/B: python tools/probes/convt_time.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from virnet_amd import ops  # noqa: E402
from virnet_amd.networks.params import ConvParam  # noqa: E402
os.environ.setdefault("VIRNET_CONV_FORM", "wx4")
for (n, h, w, cin, cout) in [(32, 128, 128, 192, 96), (32, 64, 64, 288, 192), (16, 128, 128, 160, 96), (16, 64, 64, 224, 160)]:
    cp = ConvParam(cin, cout, 2, transposed=True, stride=2).cuda()
    x = torch.rand(n, h, w, cin, device="cuda") - 0.5
    br = torch.rand(n, 2 * h, 2 * w, cout, device="cuda") - 0.5
    pw = cp.packed()
    ref = None
    for form in ("ks2", "ks3", "2"):
        os.environ.pop("VIRNET_CONVT_SLABS", None); os.environ.pop("VIRNET_CONVT_KS", None)
        if form == "ks3":
            os.environ["VIRNET_CONVT_KS"] = "3"
        elif form != "ks2":
            os.environ["VIRNET_CONVT_SLABS"] = form
        for _ in range(3):
            y, _ = ops.conv_mfma(x, pw, res=br, want_raw=True)
        if ref is None:
            ref = y.clone()
        assert torch.equal(y, ref), form
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(20):
            e0.record(); ops.conv_mfma(x, pw, res=br, want_raw=True); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        gb = (n * h * w * cin * 4 + 2 * n * 4 * h * w * cout * 4) / 1e9
        print(f"convT {cin}->{cout} @{h}x{w} x{n}  form {form}: median {ts[10]:.3f} ms  min {ts[0]:.3f}   {gb / ts[10]:.2f} TB/s", flush=True)
