"""World-size-2 gloo tests of the N>1 path's host logic (sharding, the single weight broadcast, result gather, MAX timing)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from virnet_amd import dist as vdist
from virnet_amd.networks import VIRAttResUNet


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 64, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [vdist.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        vdist.shard_range(8, 2, 2)


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    vdist.init(backend="gloo")
    torch.manual_seed(100 + rank)   # different random init per rank
    net = VIRAttResUNet(3, sigma_chn=1, n_feat=[64, 128], dep_S=3, n_resblocks=1)
    before = net.RNet.tail.weight.detach().clone()
    v0 = net.RNet.tail.weight._version
    nbytes = vdist.broadcast_parameters(net, src=0)
    flat = torch.cat([p.detach().reshape(-1) for p in net.state_dict().values()])
    a, b = vdist.shard_range(5, world, rank)
    local = torch.arange(a, b, dtype=torch.float32).view(-1, 1) * 10
    full = vdist.gather_shards(local, 5)
    tmax = vdist.max_over_ranks(1.0 + rank)
    ret[rank] = dict(nbytes=nbytes, sum=float(flat.double().sum()), changed=not torch.equal(before, net.RNet.tail.weight),
                     bumped=net.RNet.tail.weight._version > v0, full=full.view(-1).tolist(), tmax=tmax,
                     nparam=sum(p.numel() for p in net.parameters()))
    dist.destroy_process_group()


def test_weight_broadcast_and_gather_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0["nbytes"] == r1["nbytes"] == 4 * r0["nparam"]          # ONE flat collective covering every parameter
    assert r0["sum"] == r1["sum"]                                     # identical weights after the broadcast
    assert not r0["changed"] and r1["changed"] and r1["bumped"]       # rank 1 received rank 0's values, in place
    assert r0["full"] == r1["full"] == [0.0, 10.0, 20.0, 30.0, 40.0]
    assert r0["tmax"] == r1["tmax"] == 2.0
