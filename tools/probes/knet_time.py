"""KNet alone (head + body + tail + pooling), persistent body against the per-layer path: python tools/probes/knet_time.py [n ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_net  # noqa: E402
from virnet_amd import engine  # noqa: E402
from virnet_amd.utils.synth import synth_images  # noqa: E402
dev = torch.device("cuda", 0)
net, sd = build_net(dev, "sisr"); net.load_state_dict(sd, strict=True); net = net.to(dev).eval()
for n in [int(a) for a in sys.argv[1:]] or [1, 16]:
    x = synth_images(n, 3, 64, 64).to(dev)
    for p in ("1", "0"):
        os.environ["VIRNET_KNET_PERSISTENT"] = p
        with torch.no_grad():
            for _ in range(5):
                engine.knet_forward(net.KNet, x)
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                engine.knet_forward(net.KNet, x)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                engine.knet_forward(net.KNet, x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                g.replay()
            e0.record()
            for _ in range(20):
                g.replay()
            e1.record(); e1.synchronize()
            tg = e0.elapsed_time(e1) / 20
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50):
                engine.knet_forward(net.KNet, x)
            torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 50 * 1e3
        print(f"KNet n={n} persistent={p}: graph replay {tg * 1e3:7.1f} us (device time), eager {te * 1e3:7.1f} us", flush=True)
