"""Cost of the T emission per launch: the res-block convs of configs[4]'s levels, plain vs emitting (VIRNET_HIP_LIB selects a probe build)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from virnet_amd import ops
from virnet_amd.networks.params import ConvParam
torch.manual_seed(0)
for form in os.environ.get("FORMS", "wx4,bf16").split(","):
    os.environ["VIRNET_CONV_FORM"] = form
    for (n, h, w, c) in [(32, 128, 128, 96), (32, 64, 64, 192), (32, 32, 32, 288)]:
        cp = ConvParam(c, c, 3).cuda()
        x = torch.rand(n, h, w, c, device="cuda") - 0.5
        res = torch.rand(n, h, w, c, device="cuda") - 0.5
        for mode, kw in (("pre", dict(in_slope=0.2, want_raw=False, want_act=True)), ("res", dict(res=res, want_raw=True)),
                         ("mask+res", dict(res=res, mask=x, want_raw=True))):
            out = {}
            for emit in (None, dict(act=None, colsum=None), dict(act=0.2, colsum=c)):
                ts = []
                for it in range(25):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    r = ops.conv_mfma(x, cp.packed(), **kw) if emit is None else ops.conv_mfma(x, cp.packed(), emit=emit, **kw)
                    e1.record(); e1.synchronize()
                    if emit is not None and r[2] is not None:
                        ops.t_release(r[2])
                    if it >= 5:
                        ts.append(e0.elapsed_time(e1))
                ts.sort()
                out["plain" if emit is None else ("T" if emit["colsum"] is None else "T+act+sums")] = ts[len(ts) // 2]
            print(f"{form:5s} {n}x{h}x{w}x{c} {mode:9s} " + "  ".join(f"{k} {v * 1e3:7.1f} us" for k, v in out.items()), flush=True)
