"""GPU parity of the training-step kernels against torch autograd on the CPU oracle's ops.

Weight/bias gradients contract over every pixel of the batch (fp32 atomics, arbitrary order): they are held to 2e-5 of the
gradient's own scale; input gradients are ordinary convs (<= 2e-5 on O(1) data)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_ref
from virnet_amd import ops
from virnet_amd.networks.params import ConvParam
from test_ops_gpu import make_conv, maxerr, nchw, nhwc, rnd

pytestmark = pytest.mark.gpu
TOL = 2e-5


def relerr(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def autograd_conv(x, w, b, dy, *, stride=1, in_slope=None, mul=None, add=None):
    x = x.clone().requires_grad_(True); w = w.clone().requires_grad_(True); b = b.clone().requires_grad_(True)
    a = x
    if mul is not None:
        a = a * mul.view(*mul.shape, 1, 1) + add.view(*add.shape, 1, 1)
    if in_slope is not None:
        a = F.leaky_relu(a, in_slope)
    y = F.conv2d(a, w, b, stride=stride, padding=w.shape[-1] // 2)
    y.backward(dy)
    return x.grad, w.grad, b.grad


@pytest.mark.parametrize("cin,cout,h,w,n,act,sft", [(96, 96, 17, 33, 2, True, False), (64, 64, 12, 40, 2, False, False),
                                                    (32, 96, 9, 20, 3, True, True), (192, 192, 8, 32, 1, True, False),
                                                    (96, 160, 10, 35, 1, False, False)])
def test_wgrad_bias_3x3(cin, cout, h, w, n, act, sft):
    cp = make_conv(cin, cout)
    x, dy = rnd(n, cin, h, w, seed=60), rnd(n, cout, h, w, seed=61)
    mul, add = (rnd(n, cin, seed=62, lo=0.2, hi=1.0), rnd(n, cin, seed=63)) if sft else (None, None)
    _, dw_ref, db_ref = autograd_conv(x, cp.weight.detach(), cp.bias.detach(), dy, in_slope=0.2 if act else None, mul=mul, add=add)
    dw = ops.conv_wgrad(nhwc(x), nhwc(dy), tuple(cp.weight.shape), in_slope=0.2 if act else None,
                        in_mul=None if mul is None else mul.cuda(), in_add=None if add is None else add.cuda())
    db = ops.colsum(nhwc(dy))
    assert relerr(dw.cpu(), dw_ref) <= TOL and relerr(db.cpu(), db_ref) <= TOL


def test_wgrad_thin_layers():
    """tail 96->3 (gradient stored as a 16-channel record) and head 4->96 (input stored as a 16-channel record)."""
    x, dy3 = rnd(2, 96, 13, 37, seed=64), rnd(2, 3, 13, 37, seed=65)
    w = rnd(3, 96, 3, 3, seed=66) * 0.1
    _, dw_ref, db_ref = autograd_conv(x, w, torch.zeros(3), dy3)
    dy16 = torch.zeros(2, 16, 13, 37); dy16[:, :3] = dy3
    dw = ops.conv_wgrad(nhwc(x), nhwc(dy16), (3, 96, 3, 3))
    assert relerr(dw.cpu(), dw_ref) <= TOL and relerr(ops.colsum(nhwc(dy16), 3).cpu(), db_ref) <= TOL
    x4, dy = rnd(2, 4, 13, 37, seed=67), rnd(2, 96, 13, 37, seed=68)
    w2 = rnd(96, 4, 3, 3, seed=69) * 0.2
    _, dw2_ref, _ = autograd_conv(x4, w2, torch.zeros(96), dy)
    x16 = torch.zeros(2, 16, 13, 37); x16[:, :4] = x4
    dw2 = ops.conv_wgrad(nhwc(x16), nhwc(dy), (96, 4, 3, 3))
    assert relerr(dw2.cpu(), dw2_ref) <= TOL


def test_dgrad_3x3_with_mask_and_residual():
    """dx = dout + dgrad(d_f1) * lrelu'(x): the backward of conv(lrelu(x)) inside a residual block (AttResUNet.py:55,59)."""
    cp = make_conv(96, 96)
    x, dy, dout = rnd(2, 96, 11, 34, seed=70), rnd(2, 96, 11, 34, seed=71), rnd(2, 96, 11, 34, seed=72)
    dx_ref, _, _ = autograd_conv(x, cp.weight.detach(), cp.bias.detach(), dy, in_slope=0.2)
    cp.cuda()
    pw = ops.pack_weight(cp.weight, None, dgrad=True)
    dx, _ = ops.conv_mfma(nhwc(dy), pw, mask=nhwc(x), mask_slope=0.2, res=nhwc(dout), want_raw=True)
    assert maxerr(nchw(dx), dx_ref + dout) <= TOL
    # Cin != Cout and no mask
    cp2 = make_conv(64, 160)
    x2, dy2 = rnd(1, 64, 9, 33, seed=73), rnd(1, 160, 9, 33, seed=74)
    dx2_ref, _, _ = autograd_conv(x2, cp2.weight.detach(), cp2.bias.detach(), dy2)
    cp2.cuda()
    dx2, _ = ops.conv_mfma(nhwc(dy2), ops.pack_weight(cp2.weight, None, dgrad=True), want_raw=True)
    assert tuple(dx2.shape) == (1, 9, 33, 64) and maxerr(nchw(dx2), dx2_ref) <= TOL


@pytest.mark.parametrize("cin,cout,h,w,n", [(96, 192, 16, 40, 2), (192, 288, 8, 64, 1), (96, 160, 10, 6, 1)])
def test_stride2_conv_backward(cin, cout, h, w, n):
    """DownBlock.downsampler (AttResUNet.py:67): dgrad = 3x3 conv of the zero-stuffed gradient, wgrad with stride 2."""
    cp = make_conv(cin, cout, stride=2)
    x, dy = rnd(n, cin, h, w, seed=75), rnd(n, cout, h // 2, w // 2, seed=76)
    dx_ref, dw_ref, db_ref = autograd_conv(x, cp.weight.detach(), cp.bias.detach(), dy, stride=2)
    cp.cuda()
    z = ops.zero_stuff2(nhwc(dy))
    dx, _ = ops.conv_mfma(z, ops.pack_weight(cp.weight, None, dgrad=True), want_raw=True)
    dw = ops.conv_wgrad(nhwc(x), nhwc(dy), tuple(cp.weight.shape), stride=2)
    assert maxerr(nchw(dx), dx_ref) <= TOL and relerr(dw.cpu(), dw_ref) <= TOL
    assert relerr(ops.colsum(nhwc(dy)).cpu(), db_ref) <= TOL


@pytest.mark.parametrize("cin,cout,h,w,n", [(192, 96, 9, 20, 2), (288, 192, 8, 8, 1), (160, 96, 6, 7, 2)])
def test_transposed_conv_backward(cin, cout, h, w, n):
    """UpBlock.upsampler (AttResUNet.py:80): dgrad / wgrad as pointwise GEMMs over the space-to-depth gradient."""
    cp = make_conv(cin, cout, ks=2, stride=2, transposed=True)
    x = rnd(n, cin, h, w, seed=77).requires_grad_(True)
    wt = cp.weight.detach().clone().requires_grad_(True)
    dy = rnd(n, cout, 2 * h, 2 * w, seed=78)
    F.conv_transpose2d(x, wt, None, stride=2).backward(dy)
    cp.cuda()
    s2d = ops.space_to_depth2(nhwc(dy))
    assert tuple(s2d.shape) == (n, h, w, 4 * cout)
    dx, _ = ops.conv_mfma(s2d, ops.pack_weight(cp.weight, None, transposed=True, dgrad=True), want_raw=True)
    dw = ops.conv_wgrad(nhwc(x.detach()), s2d, tuple(cp.weight.shape), transposed=True)
    assert maxerr(nchw(dx), x.grad) <= TOL and relerr(dw.cpu(), wt.grad) <= TOL


def test_pack_input_backward_is_the_adjoint():
    """d sigma from the gradient of the padded 16-channel records: reflect-pad adjoint + d sqrt (util_net.py:20-25, VIRNet.py:44)."""
    sig = rnd(2, 1, 37, 45, seed=79, lo=0.1, hi=2.0).requires_grad_(True)
    g = rnd(2, 40, 48, 16, seed=80)
    cpu_ref.pad_to_multiple(sig.sqrt(), 4).backward(g[..., 3].unsqueeze(1))
    got = ops.pack_input_backward(g.cuda(), 3, (37, 45), map_=sig.detach().cuda(), map_sqrt=True)
    assert maxerr(got.cpu(), sig.grad) <= 1e-5
    acc = torch.ones(2, 1, 37, 45, device="cuda")
    ops.pack_input_backward(g.cuda(), 3, (37, 45), map_=sig.detach().cuda(), map_sqrt=True, into=acc)
    assert maxerr(acc.cpu(), sig.grad + 1) <= 1e-5


def _elbo(mu, sigma, im_noisy, im_gt, sigma_gt, eps2=1e-6, var_window=7):
    """The training loss exactly as train_denoising_syn.py:157,172,177 assembles it (virnet_amd/loss.py is pinned to the
    reference's loss/ELBO_simple.py by tests/test_loss.py)."""
    from virnet_amd.loss import elbo_denoising_simple
    alpha0 = torch.tensor([0.5 * var_window ** 2], dtype=mu.dtype, device=mu.device)
    return elbo_denoising_simple(mu, sigma, im_noisy, im_gt, eps2, alpha0, alpha0 * sigma_gt)[0]


@pytest.mark.parametrize("cfg,shape", [
    (dict(im_chn=3, sigma_chn=1, n_feat=[64, 96], dep_S=4, n_resblocks=2, noise_cond=True, extra_mode="Input"), (2, 3, 24, 40)),
    (dict(im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input"), (2, 3, 32, 32)),
    (dict(im_chn=1, sigma_chn=1, n_feat=[64, 128], dep_S=3, n_resblocks=1, noise_cond=False, extra_mode="Null"), (2, 1, 18, 22)),
    (dict(im_chn=3, sigma_chn=3, n_feat=[64, 96], dep_S=3, n_resblocks=1, noise_cond=True, extra_mode="Input"), (1, 3, 21, 19)),
    # BASELINE configs[4]'s patch size (train_denoising_syn.py, 128x128) at a batch autograd through the oracle finishes in seconds
    (dict(im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input"), (2, 3, 128, 128)),
])
def test_training_step_gradients_match_autograd_oracle(cfg, shape):
    """One ELBO step (train_denoising_syn.py:176-179): every parameter gradient of the HIP backward against torch autograd
    through the CPU oracle on identical weights and data (odd sizes exercise the reflect-pad adjoint)."""
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_images, synth_state_dict
    net = VIRAttResUNet(**cfg)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=5)
    net.load_state_dict(sd)
    net = net.cuda().train()
    n, c, h, w = shape
    gt = synth_images(n, c, h, w, seed=1)
    sig_gt = (rnd(n, 1, h, w, seed=2, lo=0.02, hi=0.3) ** 2).expand(n, cfg["sigma_chn"], h, w).contiguous()
    noisy = gt + rnd(n, c, h, w, seed=3, lo=-0.3, hi=0.3)
    eps2 = 1e-2      # (the reference's 1e-6 scales the loss by 1e6: same gradients up to that factor, harder to read)
    mu, sigma = net(noisy.cuda())
    assert mu.requires_grad and sigma.requires_grad
    loss = _elbo(mu, sigma, noisy.cuda(), gt.cuda(), sig_gt.cuda(), eps2=eps2)
    loss.backward()
    # oracle: autograd through cpu_ref on leaf copies of the same weights
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kw = {k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn")}
    mu_r, sigma_r = cpu_ref.virnet_denoise(ref, noisy, **kw)
    loss_r = _elbo(mu_r, sigma_r, noisy, gt, sig_gt, eps2=eps2)
    loss_r.backward()
    assert abs(float(loss) - float(loss_r)) <= 1e-4 * abs(float(loss_r))
    # LeakyReLU is not smooth: a pre-activation within fp32 re-association noise of 0 can land on the other side of the kink than in
    # the oracle and changes that ONE element's derivative from 1 to 0.2 (seen: 1 flip in 36 864 elements of one layer of the full
    # config -> 5.7e-3 on the MAX error of that layer's gradient).  A flip is a sparse perturbation: it moves the maximum, not the
    # bulk -- so against the UNMASKED oracle every gradient is held to fp32 noise in its median element and to 2e-2 in its worst.
    for name, p in net.named_parameters():
        g, gr = p.grad.cpu(), ref[name].grad
        assert g.shape == gr.shape, name
        scale = max(float(gr.abs().max()), 1e-12)
        assert float((g - gr).abs().median()) / scale <= 5e-5, name
        assert float((g - gr).abs().max()) / scale <= 2e-2, name
    # The sharp check: differentiate the SAME piecewise-linear function on both sides.  The oracle is run again with every LeakyReLU
    # taking its branch decisions (forward and derivative) from the HIP forward's own pre-activation signs, read off the tensors the
    # HIP backward uses as masks; then no element may differ by more than 1e-4 of its tensor's scale -- no allowance for flips.
    masks = _hip_lrelu_masks(net, noisy.cuda(), h, w)
    ref2 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    with _masked_lrelu(masks):
        mu_m, sigma_m = cpu_ref.virnet_denoise(ref2, noisy, **kw)
    assert not masks, "the oracle and the HIP tape disagree on the number of LeakyReLU sites"
    assert float((mu_m - mu_r).abs().max()) <= 1e-4          # (decisions differ only where the pre-activation is ~0)
    _elbo(mu_m, sigma_m, noisy, gt, sig_gt, eps2=eps2).backward()
    worst = 0.0
    for name, p in net.named_parameters():
        g, gr = p.grad.cpu(), ref2[name].grad
        scale = max(float(gr.abs().max()), 1e-12)
        err = float((g - gr).abs().max()) / scale
        worst = max(worst, err)
        assert err <= 1e-4, (name, err, scale)
    print(f"worst gradient element against the sign-matched oracle: {worst:.2e} of its tensor's scale")


def _hip_lrelu_masks(net, x, h, w):
    """Branch decisions (pre-activation > 0) of every LeakyReLU of the denoise-syn forward, in the order oracle/cpu_ref.py evaluates them,
    taken from the tensors the HIP backward reads as masks (virnet_amd/train.py's tape: NHWC, RNet at the padded size)."""
    from virnet_amd import train
    with torch.no_grad():
        _, _, tape = train.denoise_forward_train(net, x)
    out = [nchw(a) > 0 for a in tape.snet["acts"]]                      # DnCNN.py:23,27 (post-activation: the sign survives)
    for kind, _mod, x_in, aux in tape.misc["order"]:
        if kind == "block":                                             # AttResUNet.py:55 on the block input, :58 on conv1's output
            out += [nchw(x_in) > 0, nchw(aux[0]) > 0]                    # (aux = (f1a, emitted operand images))
    return out


class _MaskedLReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, slope):
        ctx.save_for_backward(mask)
        ctx.slope = slope
        return torch.where(mask, x, x * slope)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return torch.where(mask, g, g * ctx.slope), None, None


class _masked_lrelu:
    """Context manager: oracle.cpu_ref's F.leaky_relu consumes `masks` in call order (shapes are checked)."""
    def __init__(self, masks):
        self.masks = masks

    def __enter__(self):
        self.real = cpu_ref.F.leaky_relu
        masks = self.masks

        def lrelu(x, slope=0.01, *a, **k):
            m = masks.pop(0)
            assert m.shape == x.shape, (tuple(m.shape), tuple(x.shape))
            return _MaskedLReLU.apply(x, m, slope)

        class _F:
            def __getattr__(self_, name):
                return lrelu if name == "leaky_relu" else getattr(torch.nn.functional, name)
        self.saved = cpu_ref.F
        cpu_ref.F = _F()
        return self

    def __exit__(self, *exc):
        cpu_ref.F = self.saved
        return False


@pytest.mark.parametrize("factor", [1.0, 1e6, 1e-6])
def test_backward_does_not_depend_on_the_loss_scale(factor):
    """A mean-reduced MSE hands the backward ~6e-8 per entry of d mu, a sum-reduced loss or a GradScaler 1e4 and more; the split-fp16
    GEMMs of the backward are exact only between 6e-5 and 65504.  The fused backward rescales the incoming gradient by a power of two
    (virnet_amd/train.py), so the parameter gradients of `factor * loss` must be factor times those of the oracle -- to fp32 noise,
    for tiny and for huge factors alike (without the rescaling the 1e-6 case flushes to zero and the 1e6 case overflows)."""
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_images, synth_state_dict
    cfg = dict(im_chn=3, sigma_chn=1, n_feat=[96, 192], dep_S=4, n_resblocks=1, noise_cond=True, extra_mode="Input")
    net = VIRAttResUNet(**cfg)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=6)
    net.load_state_dict(sd)
    net = net.cuda().train()
    n, c, h, w = 2, 3, 40, 48
    gt = synth_images(n, c, h, w, seed=1)
    noisy = gt + rnd(n, c, h, w, seed=3, lo=-0.3, hi=0.3)
    mu, sigma = net(noisy.cuda())
    loss = (F.mse_loss(mu, gt.cuda()) + 0.1 * sigma.mean()) * factor            # mean reductions: d mu ~ 1e-7 * factor
    loss.backward()
    masks = _hip_lrelu_masks(net, noisy.cuda(), h, w)
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kw = {k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn")}
    with _masked_lrelu(masks):
        mu_r, sigma_r = cpu_ref.virnet_denoise(ref, noisy, **kw)
    (F.mse_loss(mu_r, gt) + 0.1 * sigma_r.mean()).backward()
    for name, p in net.named_parameters():
        g, gr = p.grad.cpu() / factor, ref[name].grad
        assert bool(torch.isfinite(g).all()), name
        scale = max(float(gr.abs().max()), 1e-30)
        assert float((g - gr).abs().max()) / scale <= 1e-4, (name, factor, float((g - gr).abs().max()) / scale, scale)


def test_train_step_config4_shape_is_mean_of_per_image_steps(monkeypatch):
    """BASELINE configs[4]'s shape, [32,3,128,128] through forward + ELBO + backward (train_denoising_syn.py:171-184), by a
    size-independent property: the loss is a mean over the batch and no op couples samples, so every parameter
    gradient of the batch step equals the mean of the 32 single-image steps -- up to the fp32 atomics' summation order in the
    weight-gradient kernel (and bit for bit in the forward, with the kernel form pinned: the default rule of ops.wx4_shape_ok looks at
    the launch size and would run the single-image steps on the direct kernel)."""
    monkeypatch.setenv("VIRNET_WX4_MIN_WGS", "0")
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_images, synth_state_dict
    cfg = dict(im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input")
    net = VIRAttResUNet(**cfg)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=5))
    net = net.cuda().train()
    n = 32
    gt = synth_images(n, 3, 128, 128, seed=1).cuda()
    sig_gt = (0.02 + 0.25 * synth_images(n, 1, 128, 128, seed=2).cuda()) ** 2
    noisy = gt + (synth_images(n, 3, 128, 128, seed=3).cuda() - 0.5) * 0.4
    params = dict(net.named_parameters())

    def step(sl):
        for p in params.values():
            p.grad = None
        mu, sigma = net(noisy[sl].contiguous())
        loss = _elbo(mu, sigma, noisy[sl], gt[sl], sig_gt[sl], eps2=1e-2)
        loss.backward()
        return float(loss), mu.detach(), {k: p.grad.double().clone() for k, p in params.items()}

    loss_b, mu_b, g_b = step(slice(0, n))
    acc = {k: torch.zeros_like(v) for k, v in g_b.items()}
    loss_sum = 0.0
    for i in range(n):
        li, mi, gi = step(slice(i, i + 1))
        assert torch.equal(mi[0], mu_b[i])                         # batch independence of the forward, bit for bit
        loss_sum += li
        for k in acc:
            acc[k] += gi[k]
    assert abs(loss_b - loss_sum / n) <= 1e-5 * abs(loss_b)
    worst = 0.0
    for k in acc:
        ref = acc[k] / n
        scale = max(float(ref.abs().max()), 1e-12)
        err = float((g_b[k] - ref).abs().max()) / scale
        worst = max(worst, err)
        assert err <= 1e-4, (k, err, scale)
    assert all(torch.isfinite(v).all() for v in g_b.values()) and worst > 0.0


def test_optimizer_step_and_repack():
    """Adam step on the HIP gradients, then the next forward must see the updated weights (packed copies follow ._version)."""
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_images, synth_state_dict
    net = VIRAttResUNet(3, sigma_chn=1, n_feat=[64, 96], dep_S=3, n_resblocks=1).cuda()
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=6))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x = synth_images(2, 3, 16, 32).cuda()
    gt = synth_images(2, 3, 16, 32, seed=9).cuda()
    losses = []
    for _ in range(4):
        opt.zero_grad()
        mu, sigma = net(x)
        loss = ((mu - gt) ** 2).mean() + 0.01 * (sigma.log() ** 2).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for n_, p in net.named_parameters() if "rnet" in n_.lower()], 1e3)
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0]


def test_training_step_bf16_operand_variant(monkeypatch):
    """BASELINE configs[4] as written (bf16): forward + ELBO + backward with VIRNET_CONV_FORM=bf16 -- the C->C 3x3 convs of the forward and
    of the input-gradient chain take bf16-rounded operands (one product per MAC, fp32 accumulation, fp32 master weights); weight
    gradients, entries / exits, strided and transposed convs stay fp32-class.  Tolerance: bf16 rounds each operand to 2^-9 relative; through
    ~40 layers of this network on the synthetic weights that is ~1e-2 on mu (BASELINE.md measured 7.5e-3 for an all-bf16 forward), so
    loss and gradients are held to 5 % of their scale against the fp32 oracle -- and the step must still descend."""
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_images, synth_state_dict
    monkeypatch.setenv("VIRNET_CONV_FORM", "bf16")
    cfg = dict(im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input")
    net = VIRAttResUNet(**cfg)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=5)
    net.load_state_dict(sd)
    net = net.cuda().train()
    n, c, h, w = 2, 3, 32, 32
    gt = synth_images(n, c, h, w, seed=1)
    sig_gt = (rnd(n, 1, h, w, seed=2, lo=0.02, hi=0.3) ** 2).contiguous()
    noisy = gt + rnd(n, c, h, w, seed=3, lo=-0.3, hi=0.3)
    mu, sigma = net(noisy.cuda())
    loss = _elbo(mu, sigma, noisy.cuda(), gt.cuda(), sig_gt.cuda(), eps2=1e-2)
    loss.backward()
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kw = {k: v for k, v in cfg.items() if k not in ("im_chn", "sigma_chn")}
    mu_r, sigma_r = cpu_ref.virnet_denoise(ref, noisy, **kw)
    loss_r = _elbo(mu_r, sigma_r, noisy, gt, sig_gt, eps2=1e-2)
    loss_r.backward()
    assert 1e-5 < float((mu.detach().cpu() - mu_r.detach()).abs().max()) < 0.1            # really reduced precision, and bounded
    assert abs(float(loss.detach()) - float(loss_r.detach())) <= 5e-2 * abs(float(loss_r.detach()))
    worst = 0.0
    for name, p in net.named_parameters():
        g, gr = p.grad.cpu(), ref[name].grad
        scale = max(float(gr.abs().max()), 1e-12)
        worst = max(worst, float((g - gr).abs().median()) / scale)
        assert float((g - gr).abs().median()) / scale <= 5e-2, name
    assert worst > 1e-5


def test_train_step_config4_as_written_bf16_full_size(monkeypatch):
    """BASELINE configs[4] AS WRITTEN: [32,3,128,128], bf16 operands (VIRNET_CONV_FORM=bf16: the large-grid instantiations of the bf16
    forward / input-gradient kernels and conv_wgrad_f16_kernel<..., BF=1> at their real sizes).  Size-independent properties: the forward
    is batch independent bit for bit; every parameter gradient of the batch step equals the mean of the single-image steps (checked on
    a subset of the images against the matching sub-batch: the bf16 rounding is per operand, so it commutes with batching); everything
    is finite; the fp32-class step and the bf16 step agree to the bf16 operand error (5 % of each gradient's scale in the median
    element); and three Adam steps on the batch descend."""
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_images, synth_state_dict
    cfg = dict(im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=3, noise_cond=True, extra_mode="Input")
    net = VIRAttResUNet(**cfg)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=5)
    net.load_state_dict(sd)
    net = net.cuda().train()
    n = 32
    gt = synth_images(n, 3, 128, 128, seed=1).cuda()
    sig_gt = (0.02 + 0.25 * synth_images(n, 1, 128, 128, seed=2).cuda()) ** 2
    noisy = gt + (synth_images(n, 3, 128, 128, seed=3).cuda() - 0.5) * 0.4
    params = dict(net.named_parameters())

    def step(sl):
        for p in params.values():
            p.grad = None
        mu, sigma = net(noisy[sl].contiguous())
        loss = _elbo(mu, sigma, noisy[sl], gt[sl], sig_gt[sl], eps2=1e-2)
        loss.backward()
        return float(loss), mu.detach(), {k: p.grad.double().clone() for k, p in params.items()}

    monkeypatch.setenv("VIRNET_CONV_FORM", "wx4")
    _, mu_f32, g_f32 = step(slice(0, n))
    monkeypatch.setenv("VIRNET_CONV_FORM", "bf16")
    loss_b, mu_b, g_b = step(slice(0, n))
    assert all(bool(torch.isfinite(v).all()) for v in g_b.values()) and bool(torch.isfinite(mu_b).all())
    assert 1e-5 < float((mu_b - mu_f32).abs().max()) < 0.1                   # really the reduced-precision kernels, and bounded
    for k in g_b:
        scale = max(float(g_f32[k].abs().max()), 1e-12)
        assert float((g_b[k] - g_f32[k]).abs().median()) / scale <= 5e-2, k
    # batch-mean property on the first four images (the sub-batch step vs the mean of its single-image steps), forward bit for bit
    _, mu4, g4 = step(slice(0, 4))
    assert torch.equal(mu4, mu_b[:4])
    acc = {k: torch.zeros_like(v) for k, v in g4.items()}
    for i in range(4):
        _, mi, gi = step(slice(i, i + 1))
        assert torch.equal(mi[0], mu_b[i])
        for k in acc:
            acc[k] += gi[k]
    for k in acc:
        ref = acc[k] / 4
        scale = max(float(ref.abs().max()), 1e-12)
        assert float((g4[k] - ref).abs().max()) / scale <= 1e-4, (k, float((g4[k] - ref).abs().max()) / scale)
    # the step descends (train_denoising_syn.py:176-184: clip + Adam)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        mu, sigma = net(noisy)
        loss = _elbo(mu, sigma, noisy, gt, sig_gt, eps2=1e-2)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for n_, p in net.named_parameters() if "rnet" in n_.lower()], 1e3)
        torch.nn.utils.clip_grad_norm_([p for n_, p in net.named_parameters() if "snet" in n_.lower()], 1e2)
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0], losses


@pytest.mark.parametrize("form", ["wx4", "bf16"])
def test_training_step_with_and_without_T_emission_agree(form, monkeypatch):
    """The convolutions of the step emit the weight-gradient GEMM's operand images from their epilogues (csrc TE = 1 kernels); VIRNET_T_EMIT=0
    re-lays every operand with virnet_chsplit instead.  The emitted image is bit for bit the re-laid one, so every WEIGHT gradient must be
    bitwise identical between the two runs (ragged 52 x 76 images: tiles that overhang the image on both axes, Winograd form on level 0);
    bias gradients are sums in a different order (per-workgroup partial sums): fp32 noise."""
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_images, synth_state_dict
    monkeypatch.setenv("VIRNET_CONV_FORM", form)
    cfg = dict(im_chn=3, sigma_chn=1, n_feat=[96, 192, 288], dep_S=5, n_resblocks=2, noise_cond=True, extra_mode="Input")
    net = VIRAttResUNet(**cfg)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=5))
    net = net.cuda().train()
    n, h, w = 12, 52, 76
    gt = synth_images(n, 3, h, w, seed=1).cuda()
    sig_gt = (0.02 + 0.25 * synth_images(n, 1, h, w, seed=2).cuda()) ** 2
    noisy = gt + (synth_images(n, 3, h, w, seed=3).cuda() - 0.5) * 0.4

    def step():
        for p in net.parameters():
            p.grad = None
        mu, sigma = net(noisy)
        _elbo(mu, sigma, noisy, gt, sig_gt, eps2=1e-2).backward()
        return mu.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}

    mu_a, g_a = step()
    monkeypatch.setenv("VIRNET_T_EMIT", "0")
    mu_b, g_b = step()
    # (the forward may pick another tile form for the emitting launches: same arithmetic class, not the same bits -- compare gradients
    #  only where the forms coincide, i.e. under the pinned form below; here: fp32 noise)
    assert float((mu_a - mu_b).abs().max()) <= 2e-5 * max(1.0, float(mu_b.abs().max()))
    for k in g_a:
        scale = max(float(g_b[k].abs().max()), 1e-12)
        assert float((g_a[k] - g_b[k]).abs().max()) <= (2e-2 if form == "bf16" else 1e-4) * scale, k
    # pinned kernel form and tile height: emission changes no weight-gradient bit
    monkeypatch.setenv("VIRNET_DETERMINISTIC", "1")
    _, g_d0 = step()                                     # VIRNET_T_EMIT=0
    monkeypatch.delenv("VIRNET_T_EMIT")
    _, g_d1 = step()
    thin = ("SNet.conv1.weight", "SNet.conv_last.weight", "RNet.head.weight", "RNet.tail.weight")
    for k in g_d0:
        if form == "bf16" and k in thin:
            # (bf16 form: a few-channel layer takes the emitted bf16 image of its big operand; without emission it re-lays it in split fp16)
            assert float((g_d0[k] - g_d1[k]).abs().max()) <= 2e-2 * max(float(g_d0[k].abs().max()), 1e-12), k
        elif k.endswith(".weight"):
            assert torch.equal(g_d0[k], g_d1[k]), k
        else:
            assert float((g_d0[k] - g_d1[k]).abs().max()) <= 2e-5 * max(float(g_d0[k].abs().max()), 1e-12), k
