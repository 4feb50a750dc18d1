import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from virnet_amd import ops
from virnet_amd.networks.params import ConvParam
N = 16
x = torch.rand(N, 3, 64, 64, device="cuda"); vec = torch.rand(N, 5, device="cuda")
head = ConvParam(8, 96, 3).cuda()
with ops.forward_scope():
    fn = lambda: ops.conv_entry(x, head.packed(), 256, 256, sf=4, vec=vec)
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(30):
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); print(f"entry 8->96 (SISR head, 16 x 256^2): median {ts[15]*1e3:.1f} us, min {ts[0]*1e3:.1f} us")
