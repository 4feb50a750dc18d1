import os, sys, torch
sys.path.insert(0, "/root/repo")
from virnet_amd import ops
from virnet_amd.networks.params import ConvParam
torch.manual_seed(0)
c = 160
for (n, h, w) in [(1, 18, 40), (1, 8, 32), (2, 16, 64)]:
    cp = ConvParam(c, c, 3).cuda()
    x = (torch.rand(n, h, w, c, device="cuda") - 0.5)
    os.environ["VIRNET_WX4_MIN_WGS"] = "1"; os.environ["VIRNET_WX4_ROWS"] = "8"
    os.environ["VIRNET_WX4_WIDE"] = "0"
    ref, _ = ops.conv_mfma(x, cp.packed(), want_raw=True)
    os.environ["VIRNET_WX4_WIDE"] = "1"
    out, _ = ops.conv_mfma(x, cp.packed(), want_raw=True)
    d = (out - ref).abs()
    print((n, h, w), "max", float(d.max()))
    print("  per slab:", [round(float(d[..., s * 32:(s + 1) * 32].max()), 4) for s in range(5)])
    print("  per row :", [round(float(d[:, r].max()), 4) for r in range(h)])
    print("  per col8:", [round(float(d[:, :, q * 8:(q + 1) * 8].max()), 4) for q in range((w + 7) // 8)])
