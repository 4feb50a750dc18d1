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


def _worker(rank, world, port, ret, own_reducer=False):
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
    if own_reducer:
        ddp = vdist.DistributedTrainer(net, bucket_bytes=1 << 20)               # bucketed all-reduce overlapped with the backward kernels
    else:
        ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0])   # broadcasts rank 0's parameters (train_denoising_syn.py:71)
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
    ret[rank] = dict(psum=float(flat.double().sum()), gsum=float(gsum), losses=losses,
                     buckets=ddp.reducer.bucket_sizes if own_reducer else None)
    dist.destroy_process_group()


def test_ddp_training_two_ranks():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0["psum"] == r1["psum"] and r0["gsum"] == r1["gsum"]      # same averaged gradients -> parameters stay identical
    assert r0["losses"] != r1["losses"]                                # different shards
    assert r0["losses"][-1] < r0["losses"][0] and r1["losses"][-1] < r1["losses"][0]


def test_distributed_trainer_matches_ddp():
    """virnet_amd.dist.DistributedTrainer (own bucketed reducer fed from inside the backward) against torch's DDP on the same run."""
    out = []
    for own in (False, True):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ret = mp.Manager().dict()
        mp.spawn(_worker, args=(2, port, ret, own), nprocs=2, join=True)
        assert ret[0]["psum"] == ret[1]["psum"] and ret[0]["gsum"] == ret[1]["gsum"]
        out.append(dict(ret[0]))
    ddp, own = out
    assert len(own["buckets"]) >= 2                                    # more than one collective per step
    assert own["losses"] == pytest.approx(ddp["losses"], rel=1e-5)
    assert own["psum"] == pytest.approx(ddp["psum"], rel=1e-6) and own["gsum"] == pytest.approx(ddp["gsum"], rel=1e-4, abs=1e-6)
