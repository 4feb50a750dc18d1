import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_ops_gpu import make_conv, nhwc, rnd
from virnet_amd import ops
cin, cout, n, h, w = [int(v) for v in sys.argv[1:6]]
cp = make_conv(cin, cout, seed=202).cuda()
x = nhwc(rnd(n, cin, h, w, seed=302))
outs = {}
for form in ("1", "2", "direct"):
    os.environ["VIRNET_CONV_FORM"] = "direct" if form == "direct" else "f16x3"
    os.environ["VIRNET_F16_MREP"] = form if form != "direct" else "1"
    outs[form] = ops.conv_mfma(x, cp.packed(), want_raw=True)[0].cpu()
for form in ("1", "2"):
    e = (outs[form] - outs["direct"]).abs()
    bad = (e > 1e-4).nonzero()
    print("form", form, "max err", float(e.max()), "bad elements", len(bad))
    if len(bad):
        print(" first bad (n,y,x,c):", bad[:5].tolist(), " last:", bad[-3:].tolist())
        ys = sorted(set(bad[:, 1].tolist())); xs = sorted(set(bad[:, 2].tolist())); cs = sorted(set(bad[:, 3].tolist()))
        print(" rows", ys[:12], "cols", xs[:12], "chans", cs[:8], "...", len(cs))
