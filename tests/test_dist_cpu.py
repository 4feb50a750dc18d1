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


def _shard_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    vdist.init(backend="gloo")
    a, b = vdist.shard_range(256, world, rank)
    owned = torch.zeros(256, dtype=torch.int64)
    owned[a:b] = 1
    dist.all_reduce(owned)                                 # how many ranks own each image
    local = torch.arange(a, b, dtype=torch.float32).view(-1, 1)
    full = vdist.gather_shards(local, 256)
    ret[rank] = dict(span=(a, b), owners_min=int(owned.min()), owners_max=int(owned.max()), full_ok=bool(torch.equal(full.view(-1), torch.arange(256.0))),
                     tmax=vdist.max_over_ranks(float(rank)))
    dist.destroy_process_group()


def test_eight_ranks_shard_256_images_exactly():
    """BASELINE configs[2]: 256 images over 8 ranks (one per GPU) -- every image owned by exactly one rank, 32 per rank, the gathered
    result in image order, max-over-ranks timing; eight gloo processes on the CPU."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(8, port, ret), nprocs=8, join=True)
    assert sorted(ret[r]["span"] for r in range(8)) == [(32 * r, 32 * r + 32) for r in range(8)]
    for r in range(8):
        assert ret[r]["owners_min"] == ret[r]["owners_max"] == 1 and ret[r]["full_ok"] and ret[r]["tmax"] == 7.0


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


def _reducer_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    vdist.init(backend="gloo")
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((300,), (40, 10), (7,), (1000,), (3, 3))]
    frozen = torch.nn.Parameter(torch.zeros(5), requires_grad=False)
    red = vdist.GradientReducer(params + [frozen], bucket_bytes=2000)          # 500 floats per bucket
    steps = []
    for step in range(3):
        red.start()
        order = [4, 2, 0, 3] if step < 2 else [0, 4, 2, 3]                       # parameter 1 never produces a gradient; order may change
        for i in order:
            red.push({params[i]: torch.full(params[i].shape, float((rank + 1) * (i + 1) + step))})
        out = red.finish()
        steps.append({i: float(out[params[i]].reshape(-1)[0]) for i in range(5)})
        assert all(out[p].shape == p.shape for p in params) and frozen not in out
        assert all(torch.all(out[params[i]] == out[params[i]].reshape(-1)[0]) for i in range(5))
    ret[rank] = dict(steps=steps, buckets=red.bucket_sizes)
    dist.destroy_process_group()


def test_gradient_reducer_buckets_average_world2():
    """Bucketed asynchronous averaging: mean over ranks, zeros for parameters without a gradient, layout fixed by the
    first step's production order, later steps reduce bucket by bucket as they fill."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(_reducer_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] == ret[1]
    assert ret[0]["buckets"] == [316, 1000, 400]          # (9 + 7 + 300) | 1000 | the parameter never produced (400), last
    for step, got in enumerate(ret[0]["steps"]):
        for i in range(5):
            want = 0.0 if i == 1 else (1 * (i + 1) + step + 2 * (i + 1) + step) / 2.0
            assert got[i] == want, (step, i, got[i], want)


def test_gradient_reducer_single_process_is_identity():
    p = [torch.nn.Parameter(torch.zeros(4, 4)), torch.nn.Parameter(torch.zeros(3))]
    red = vdist.GradientReducer(p)
    for _ in range(2):
        red.start()
        red.push({p[1]: torch.arange(3.0)})
        red.push({p[0]: torch.ones(4, 4)})
        out = red.finish()
        assert torch.equal(out[p[0]], torch.ones(4, 4)) and torch.equal(out[p[1]], torch.arange(3.0))


def test_reducer_gradients_do_not_alias_buckets():
    """finish() must not hand out views of the flat buckets -- autograd adopts what it gets as p.grad, and the next
    step's push() would overwrite it before `p.grad += new` runs (2x gradients with zero_grad(set_to_none=False) / accumulation)."""
    import torch
    from virnet_amd.dist import GradientReducer
    ps = [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(3, 2))]
    red = GradientReducer(ps, bucket_bytes=16)
    for step in range(3):
        red.start()
        red.push({ps[0]: torch.full((5,), 1.0 + step), ps[1]: torch.full((3, 2), 10.0 + step)})
        out = red.finish()
        for p in ps:                                       # what AccumulateGrad does: adopt on the first step, accumulate afterwards
            if p.grad is None:
                p.grad = out[p]
            else:
                p.grad += out[p]
        bucket_ptrs = {b.data_ptr() for b in red._buckets}
        assert all(out[p].data_ptr() not in bucket_ptrs for p in ps)
    assert torch.equal(ps[0].grad, torch.full((5,), 1.0 + 2.0 + 3.0))          # accumulation over three steps, not 2x the last
    assert torch.equal(ps[1].grad, torch.full((3, 2), 10.0 + 11.0 + 12.0))


def test_rank_topology_refuses_shared_devices_when_enough_are_visible():
    """bench.py's multi-GPU guard: ranks piled on one device are an error when every rank sees enough devices, the
    declared 1-GPU rehearsal (fewer devices than ranks) passes, and the block reports what the driver needs to see."""
    from virnet_amd.dist import rank_topology
    ok = rank_topology([(f"GPU-{i}", 8) for i in range(8)], 8)
    assert ok["ranks_seen"] == 8 and ok["ranks"] == 8 and ok["devices_visible_per_rank"] == [8] * 8
    rehearsal = rank_topology([("GPU-0", 1)] * 8, 8)                   # one visible device per rank: rehearsal on a 1-GPU box
    assert rehearsal["ranks_seen"] == 1
    with pytest.raises(RuntimeError, match="refusing to report"):
        rank_topology([("GPU-0", 8), ("GPU-0", 8)] + [(f"GPU-{i}", 8) for i in range(2, 8)], 8)      # two ranks on GPU-0, 8 visible
    with pytest.raises(RuntimeError, match="distinct devices"):
        rank_topology([("GPU-3", 4)] * 4, 4)
    with pytest.raises(ValueError):
        rank_topology([("GPU-0", 1)], 2)
