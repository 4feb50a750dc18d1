import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import cpu_ref
from virnet_amd import train, ops
from virnet_amd.networks import VIRAttResUNet
from virnet_amd.utils.synth import synth_images, synth_state_dict
from test_backward_gpu import _elbo
from test_ops_gpu import rnd, nchw
cfg = dict(im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input")
net = VIRAttResUNet(**cfg)
sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=5)
net.load_state_dict(sd); net = net.cuda().train()
n, c, h, w = 2, 3, 32, 32
gt = synth_images(n, c, h, w, seed=1)
sig_gt = (rnd(n, 1, h, w, seed=2, lo=0.02, hi=0.3) ** 2).contiguous()
noisy = gt + rnd(n, c, h, w, seed=3, lo=-0.3, hi=0.3)
# oracle with recording
rec = {}
orig = cpu_ref.att_res_block
def patched(sdx, prefix, x, extra):
    f1 = cpu_ref._conv(sdx, prefix + "conv1", F.leaky_relu(x, 0.2)); f1.retain_grad(); x.retain_grad()
    f2 = cpu_ref._conv(sdx, prefix + "conv2", F.leaky_relu(f1, 0.2))
    out = x + f2; out.retain_grad()
    rec[prefix] = (x, f1, out)
    return out
cpu_ref.att_res_block = patched
ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
kw = {k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn")}
mu_r, s_r = cpu_ref.virnet_denoise(ref, noisy, **kw)
_elbo(mu_r, s_r, noisy, gt, sig_gt, eps2=1e-2).backward()
# ours: run forward_train + backward manually capturing d tensors
with torch.no_grad():
    mu, sigma, tape = train.denoise_forward_train(net, noisy.cuda())
mu_l = mu.clone().requires_grad_(True); s_l = sigma.clone().requires_grad_(True)
_elbo(mu_l, s_l, noisy.cuda(), gt.cuda(), sig_gt.cuda(), eps2=1e-2).backward()
print("dmu err", float((mu_l.grad.cpu() - mu_r.grad if mu_r.grad is not None else torch.zeros(1)).abs().max()) if False else "", "mu err %.2e" % float((mu.cpu()-mu_r).abs().max()))
# replicate backward loop with capture
names = {}
for name, m in net.named_modules(): names[m] = name
g16 = ops.pack_input(mu_l.grad.contiguous(), 32, 32, zero_pad=True)
dx, _ = ops.conv_mfma(g16, net.RNet.tail.packed_dgrad(), want_raw=True)
nb = tape.misc["nbridges"]; dbridge = [None]*nb
for kind, mod, x_in, aux in reversed(tape.misc["order"]):
    if kind == "block":
        pre = names[mod] + "."
        xr, f1r, outr = rec[pre]
        e_dout = float((nchw(dx) - outr.grad).abs().max()) / float(outr.grad.abs().max())
        f1a = aux
        sign_mismatch = int(((nchw(f1a) > 0) != (f1r.detach() > 0)).sum())
        d_f1, _ = ops.conv_mfma(dx, mod.conv2.packed_dgrad(), mask=f1a, mask_slope=0.2, want_raw=True)
        e_df1 = float((nchw(d_f1) - f1r.grad).abs().max()) / float(f1r.grad.abs().max())
        xsign = int(((nchw(x_in) > 0) != (xr.detach() > 0)).sum())
        dx, _ = ops.conv_mfma(d_f1, mod.conv1.packed_dgrad(), mask=x_in, mask_slope=0.2, res=dx, want_raw=True)
        e_dx = float((nchw(dx) - xr.grad).abs().max()) / float(xr.grad.abs().max())
        print(f"{pre:34s} dout {e_dout:.1e}  f1 sign flips {sign_mismatch:3d}  d_f1 {e_df1:.1e}  x sign flips {xsign:3d}  dx {e_dx:.1e}")
    elif kind == "up":
        dbridge[aux] = dx
        s2d = ops.space_to_depth2(dx)
        dx, _ = ops.conv_mfma(s2d, mod.packed_dgrad(), want_raw=True)
    else:
        nb -= 1
        dx, _ = ops.conv_mfma(ops.zero_stuff2(dx), mod.packed_dgrad(), res=dbridge[nb], want_raw=True)
