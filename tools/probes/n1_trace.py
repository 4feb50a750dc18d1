"""Per-launch timeline of ONE single-image forward (256^2): run under rocprofv3 --kernel-trace; prints nothing itself."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import SYN_CFG
from virnet_amd.networks import VIRAttResUNet
from virnet_amd.utils.synth import synth_images, synth_state_dict
net = VIRAttResUNet(im_chn=3, sigma_chn=1, **SYN_CFG)
net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}))
net = net.cuda().eval()
size = os.environ.get("N1_SIZE", "256")          # "256" or "481x321"
hh, ww = (int(v) for v in size.split("x")) if "x" in size else (int(size), int(size))
x = synth_images(1, 3, hh, ww).cuda()
with torch.no_grad():
    for _ in range(20):
        net(x)
    torch.cuda.synchronize()
