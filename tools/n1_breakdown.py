import sys, os, torch
sys.path.insert(0, os.getcwd())
from bench import SYN_CFG
from virnet_amd import ops
from virnet_amd.networks import VIRAttResUNet
from virnet_amd.utils.synth import synth_images, synth_state_dict
net = VIRAttResUNet(im_chn=3, sigma_chn=1, **SYN_CFG)
net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}))
net = net.cuda().eval()
for shape in [(1,3,481,321),(1,3,128,128)]:
    x = synth_images(*shape).cuda()
    with torch.no_grad():
        for _ in range(3): net(x)
        t = ops.LaunchTimer(); ops.set_launch_timer(t)
        for _ in range(5): net(x)
        ops.set_launch_timer(None)
    torch.cuda.synchronize()
    per = {}
    for var, flops, e0, e1 in t.records:
        d = per.setdefault((var, round(flops / 1e9, 2)), [0, 0.0])
        d[0] += 1; d[1] += e0.elapsed_time(e1)
    for (var, gf), (cnt, ms) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("      ", var, "%.2f GF/launch" % gf, "x%d" % (cnt // 5), "%.1f us/launch" % (ms / cnt * 1e3), "%.1f TF" % (gf / (ms / cnt)))
    s = t.summary()
    tot = sum(v["ms"] for v in s.values())/5
    print(shape, "conv time per forward %.3f ms" % tot)
    for k, v in sorted(s.items(), key=lambda kv: -kv[1]["ms"]):
        print("   ", k, "launches/fwd", v["launches"]//5, "ms/fwd %.3f" % (v["ms"]/5), "TF %.1f" % (v["flops"]/v["ms"]/1e9))
