import cProfile, pstats, sys, torch
sys.path.insert(0, "/root/repo")
from bench import build_net
from virnet_amd.utils.synth import synth_images
dev = torch.device("cuda", 0)
net, sd = build_net(dev, "denoise"); net.load_state_dict(sd, strict=True); net = net.to(dev).eval()
x = synth_images(1, 3, 256, 256).to(dev)
with torch.no_grad():
    for _ in range(5): net(x)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): net(x)
    torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
