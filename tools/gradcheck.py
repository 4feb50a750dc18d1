import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import cpu_ref
from virnet_amd.networks import VIRAttResUNet
from virnet_amd.utils.synth import synth_images, synth_state_dict
from test_backward_gpu import _elbo
from test_ops_gpu import rnd
cfg = dict(im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input")
net = VIRAttResUNet(**cfg)
sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=5)
net.load_state_dict(sd); net = net.cuda().train()
n, c, h, w = 2, 3, 32, 32
gt = synth_images(n, c, h, w, seed=1)
sig_gt = (rnd(n, 1, h, w, seed=2, lo=0.02, hi=0.3) ** 2).contiguous()
noisy = gt + rnd(n, c, h, w, seed=3, lo=-0.3, hi=0.3)
mu, sigma = net(noisy.cuda()); _elbo(mu, sigma, noisy.cuda(), gt.cuda(), sig_gt.cuda(), eps2=1e-2).backward()
kw = {k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn")}
def oracle(dtype):
    ref = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items()}
    mu_r, s_r = cpu_ref.virnet_denoise(ref, noisy.to(dtype), **kw)
    _elbo(mu_r, s_r, noisy.to(dtype), gt.to(dtype), sig_gt.to(dtype), eps2=1e-2).backward()
    return ref
r32, r64 = oracle(torch.float32), oracle(torch.float64)
worst = []
for name, p in net.named_parameters():
    g64 = r64[name].grad; sc = float(g64.abs().max())
    e_hip = float((p.grad.cpu().double() - g64).abs().max()) / sc
    e_cpu = float((r32[name].grad.double() - g64).abs().max()) / sc
    worst.append((e_hip, e_cpu, name))
worst.sort(reverse=True)
for e in worst[:6]: print("hip %.2e  cpu32 %.2e  %s" % e)
print("max hip %.2e, max cpu32 %.2e" % (max(w[0] for w in worst), max(w[1] for w in worst)))
