"""cProfile of the eager single-image SISR x4 forward's HOST side (the forward is host-bound: ~50 launches of ~24 us of GPU work each):
python tools/probes/host_profile_sisr.py"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_net
from virnet_amd.utils.synth import synth_images
dev = torch.device("cuda", 0)
net, sd = build_net(dev, "sisr"); net.load_state_dict(sd, strict=True); net = net.to(dev).eval()
x = synth_images(1, 3, 64, 64).to(dev)
with torch.no_grad():
    for _ in range(5): net(x, 4)
    torch.cuda.synchronize()
    os.environ["VIRNET_RANGE_GUARD"] = "0"
    t0 = time.perf_counter()
    for _ in range(200): net(x, 4)
    t1 = time.perf_counter()                       # host time to ENQUEUE 200 forwards (no sync inside: guard off)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"guard off: host enqueue {1e3 * (t1 - t0) / 200:.3f} ms per forward, with the final sync {1e3 * (t2 - t0) / 200:.3f} ms")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): net(x, 4)
    torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
