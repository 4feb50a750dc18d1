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
        ddp = vdist.DistributedTrainer(net, bucket_bytes=1 << 18)               # bucketed all-reduce overlapped with the backward kernels
    else:
        ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0])   # broadcasts rank 0's parameters (train_denoising_syn.py:71)
    opt = torch.optim.Adam(ddp.parameters(), lr=1e-3)
    alpha0 = torch.tensor([24.5], device=dev)
    a, b = vdist.shard_range(4, world, rank)                                  # DistributedSampler's role: disjoint shards
    x_all, gt_all = synth_images(4, 3, 16, 32).to(dev), synth_images(4, 3, 16, 32, seed=9).to(dev)
    x, gt = x_all[a:b].contiguous(), gt_all[a:b].contiguous()
    sig_gt = torch.full((b - a, 1, 16, 32), 0.01, device=dev)
    losses = []
    # instrumentation: position of every all-reduce start relative to the weight-gradient launches of the same backward
    from virnet_amd import ops as vops
    events, real_ar, real_wgrad = [], dist.all_reduce, vops.conv_wgrad

    def ar(*a, **k):
        events.append("ar")
        return real_ar(*a, **k)

    def wg(*a, **k):
        events.append("wgrad")
        return real_wgrad(*a, **k)
    early = []
    for _ in range(3):
        opt.zero_grad()
        mu, sigma = ddp(x)
        loss = elbo_denoising_simple(mu, sigma, x, gt, 1e-2, alpha0, alpha0 * sig_gt)[0]
        events.clear()
        dist.all_reduce, vops.conv_wgrad = ar, wg
        try:
            loss.backward()                                                    # DDP all-reduces the HIP-computed gradients
        finally:
            dist.all_reduce, vops.conv_wgrad = real_ar, real_wgrad
        last_wgrad = max(i for i, e in enumerate(events) if e == "wgrad")
        early.append(sum(1 for e in events[:last_wgrad] if e == "ar"))
        opt.step()
        losses.append(float(loss.detach()))
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gsum = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).double().sum()
    ret[rank] = dict(psum=float(flat.double().sum()), gsum=float(gsum), losses=losses,
                     buckets=ddp.reducer.bucket_sizes if own_reducer else None, early_allreduces=early)
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
    # overlap: in EVERY step (the first included) at least two buckets are already being reduced while weight-gradient kernels are
    # still being launched; torch's DDP on the one-Function module reduces only after the whole backward
    assert all(e >= 2 for e in own["early_allreduces"]), own["early_allreduces"]
    assert all(e == 0 for e in ddp["early_allreduces"]), ddp["early_allreduces"]
    # (two runs of the same step differ by the summation order of the weight-gradient kernel's fp32 atomics; Adam's normalisation
    # amplifies that on near-zero gradients, so three steps later the losses agree to ~1e-5, not bit for bit)
    assert own["losses"] == pytest.approx(ddp["losses"], rel=1e-4), (own["losses"], ddp["losses"])
    assert own["psum"] == pytest.approx(ddp["psum"], rel=1e-5), (own["psum"], ddp["psum"])
    assert own["gsum"] == pytest.approx(ddp["gsum"], rel=1e-3, abs=1e-5), (own["gsum"], ddp["gsum"])


def test_bench_self_spawns_two_ranks():
    """`python bench.py --gpus 2` with NO torchrun environment must launch its own ranks (the driver's 8-GPU command may arrive
    either way) and rank 0 must print ONE JSON line for the whole job.  One GPU here, so both ranks share cuda:0 over gloo; on the
    8-GPU node the same code path runs backend "nccl" = RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["VIRNET_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 64 and d["config"]["images_per_gpu"] == 32
    assert "42157328 B" in d["config"]["parallelism"]              # the one flat weight broadcast: 10 539 332 fp32 parameters
    assert d["value"] > 0 and d["value"] == pytest.approx(64 * 2 / (d["ms_per_step"] * 2e-3), rel=1e-3)
    assert 0.0 < d["roofline"]["frac"] <= 1.0 and 0.0 < d["roofline"]["frac_algorithmic"] <= d["roofline"]["frac"]
    mg = d["multi_gpu"]                                               # per-rank diagnostics of an N-rank line
    assert mg["backend"] == "gloo" and len(mg["per_rank"]) == 2 and sorted(r["rank"] for r in mg["per_rank"]) == [0, 1]
    assert mg["ranks_seen"] == 1                                      # both ranks share this box's one GPU (8 distinct UUIDs on the 8-GPU node)
    assert all(r["images_per_s"] > 0 for r in mg["per_rank"]) and mg["broadcast_bytes"] == 42157328 and mg["broadcast_ms"] > 0


def test_bench_train_two_ranks_averages_gradients():
    """`bench.py --task train --gpus 2`: the training step of the 2-rank job is data-parallel (DistributedTrainer: gradients all-reduced
    inside the backward), and says so in its JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["VIRNET_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--task", "train", "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "4", "--size", "64", "--no-cpu-baseline", "--no-roofline"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "gradient all-reduce per step" in d["config"]["parallelism"]
    assert d["config"]["global_batch"] == 8 and d["value"] > 0


def _worker_scales(rank, world, port, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      VIRNET_DIST_BACKEND="gloo")
    from virnet_amd import dist as vdist
    from virnet_amd.networks import VIRAttResUNet
    from virnet_amd.utils.synth import synth_images, synth_state_dict
    vdist.init()
    dev = torch.device("cuda", 0)
    cfg = dict(sigma_chn=1, n_feat=[64, 96], dep_S=3, n_resblocks=1)
    net = VIRAttResUNet(3, **cfg).to(dev)
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=4))
    plain = VIRAttResUNet(3, **cfg).to(dev)
    plain.load_state_dict(net.state_dict())
    ddp = vdist.DistributedTrainer(net, bucket_bytes=1 << 18)
    factors = (1.0, 3.0e-6)                                  # the ranks' incoming gradients sit 18 binades apart
    xs = [synth_images(2, 3, 16, 32, seed=20 + r).to(dev) for r in range(world)]
    gts = [synth_images(2, 3, 16, 32, seed=30 + r).to(dev) for r in range(world)]

    def loss_of(m, r):
        mu, sigma = m(xs[r])
        return (((mu - gts[r]) ** 2).mean() + 0.1 * sigma.mean()) * factors[r]
    loss_of(ddp, rank).backward()
    got = {k: p.grad.double().clone() for k, p in net.named_parameters()}
    want = None
    for r in range(world):                                   # what the average must be: every rank's term from a plain single-rank backward
        for p in plain.parameters():
            p.grad = None
        loss_of(plain, r).backward()
        term = {k: p.grad.double() / world for k, p in plain.named_parameters()}
        want = term if want is None else {k: want[k] + term[k] for k in want}
    worst = 0.0
    for k in want:
        worst = max(worst, float((got[k] - want[k]).abs().max()) / max(float(want[k].abs().max()), 1e-30))
    ret[rank] = dict(worst=worst, gsum=float(torch.cat([g.reshape(-1) for g in got.values()]).sum()))
    dist.destroy_process_group()


def test_distributed_trainer_ranks_with_different_loss_scales():
    """ADVICE r03 (train.py:243): the backward's power-of-two rescaling must use ONE factor on every rank, or the all-reduced sum mixes
    differently scaled terms.  Two ranks whose losses differ by 3e5: every rank ends with mean_k(f_k g_k), identical on both."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker_scales, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0]["gsum"] == ret[1]["gsum"]
    assert ret[0]["worst"] <= 1e-4 and ret[1]["worst"] <= 1e-4, (ret[0]["worst"], ret[1]["worst"])


def test_bench_eight_rank_rehearsal_is_not_host_bound():
    """Pre-flight of the driver's 8-GPU run on the one GPU of this box (gloo, 4 images per rank): `bench.py --gpus 8` self-spawns 8
    ranks, every rank reports, the one flat weight broadcast is 42 157 328 bytes, and with 8 Python processes sharing the host no rank
    needs more host time to ENQUEUE a step than a fraction of what the metric's 32-image step takes on the device (~22 ms): the ranks'
    CPU thread pools are capped (bench.py), the hot path enqueues from one thread."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["VIRNET_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--batch", "4", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["global_batch"] == 32 and d["config"]["images_per_gpu"] == 4
    mg = d["multi_gpu"]
    assert mg["backend"] == "gloo" and sorted(r["rank"] for r in mg["per_rank"]) == list(range(8))
    assert mg["ranks_seen"] == 1 and mg["broadcast_bytes"] == 42157328
    assert d["value"] == pytest.approx(32 * 4 / (d["ms_per_step"] * 4e-3), rel=1e-3)
    for r in mg["per_rank"]:
        assert r["images_per_s"] > 0 and 1 <= r["cpu_threads"] <= 8, r
        # enqueueing one step on an empty queue, 8 processes at it together: << the 22 ms the 32-image step keeps a GPU busy
        # (enqueue_ms_per_step is NOT that: with the 8 ranks time-sharing this box's one GPU the launch queue fills and the call blocks)
        assert r["host_ms_one_step_empty_queue"] <= 8.0, r


def test_bench_dry_run_topology_two_ranks_on_one_gpu():
    """`bench.py --gpus 2 --dry-run-topology`: no timed steps; every rank's device is reported, the 42-MB weight broadcast
    is timed on its own, and the 1-GPU rehearsal (fewer devices than ranks) is let through -- virnet_amd.dist.rank_topology refuses the
    same picture when enough devices are visible (tests/test_dist_cpu.py)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["VIRNET_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-topology"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])["dry_run_topology"]
    assert d["ranks"] == 2 and d["ranks_seen"] == 1 and len(d["uuids"]) == 2 and d["backend"] == "gloo"
    assert d["broadcast_bytes"] == 42157328 and len(d["broadcast_ms"]) == 3 and min(d["broadcast_ms"]) > 0
