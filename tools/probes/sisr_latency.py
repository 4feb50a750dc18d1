import sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import build_net
dev = torch.device("cuda", 0)
net, sd = build_net(dev, "sisr")
net.load_state_dict(sd, strict=True); net = net.to(dev).eval()
from virnet_amd.utils.synth import synth_images
from virnet_amd import ops
for n in (1, 16):
    x = synth_images(n, 3, 64, 64).to(dev)
    with torch.no_grad():
        for _ in range(5): net(x, 4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): net(x, 4)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        timer = ops.LaunchTimer() if hasattr(ops, "LaunchTimer") else None
    print(f"sisr x4 eager ({n}, 3, 64, 64) -> 256x256: {dt*1e3:7.3f} ms / forward")
