"""DDP training step on the HIP path: two ranks, torch DistributedDataParallel around the drop-in module, gradients
all-reduced by torch.distributed.  The GPU box has ONE device, so both ranks share cuda:0 and the backend is gloo (RCCL needs
one GPU per rank); the driver's multi-GPU runs use backend "nccl" = RCCL with the same code (virnet_amd/dist.py::init)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      VIRNET_DIST_BACKEND="gloo")
    from virnet_amd import dist as vdist
    from virnet_amd.loss import elbo_denoising_simple
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_images, synth_state_dict
    vdist.init()
    dev = torch.device("cuda", 0)
    torch.manual_seed(100 + rank)
    net = VIRAttResUNet(3, sigma_chn=1, n_feat=[64, 96], dep_S=3, n_resblocks=1).to(dev)
    if rank == 0:
        net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=4))
    ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0])       # broadcasts rank 0's parameters (train_denoising_syn.py:71)
    opt = torch.optim.Adam(ddp.parameters(), lr=1e-3)
    alpha0 = torch.tensor([24.5], device=dev)
    a, b = vdist.shard_range(4, world, rank)                                  # DistributedSampler's role: disjoint shards
    x_all, gt_all = synth_images(4, 3, 16, 32).to(dev), synth_images(4, 3, 16, 32, seed=9).to(dev)
    x, gt = x_all[a:b].contiguous(), gt_all[a:b].contiguous()
    sig_gt = torch.full((b - a, 1, 16, 32), 0.01, device=dev)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        mu, sigma = ddp(x)
        loss = elbo_denoising_simple(mu, sigma, x, gt, 1e-2, alpha0, alpha0 * sig_gt)[0]
        loss.backward()                                                        # DDP all-reduces the HIP-computed gradients
        opt.step()
        losses.append(float(loss.detach()))
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gsum = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).double().sum()
    ret[rank] = dict(psum=float(flat.double().sum()), gsum=float(gsum), losses=losses)
    dist.destroy_process_group()


def test_ddp_training_two_ranks():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0["psum"] == r1["psum"] and r0["gsum"] == r1["gsum"]      # same averaged gradients -> parameters stay identical
    assert r0["losses"] != r1["losses"]                                # different shards
    assert r0["losses"][-1] < r0["losses"][0] and r1["losses"][-1] < r1["losses"][0]
