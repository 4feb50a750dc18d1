"""Multi-GPU plumbing: one process per GPU, images sharded, weights broadcast once; for training, bucketed gradient averaging.

The path shards by independent images (no batch statistics, no cross-sample op -- SURVEY.md 8e), so the only collective
is ONE flat fp32 weight broadcast at start-up (42 MB for the denoise-syn net).  On ROCm ``backend="nccl"`` is RCCL; on the
fully connected xGMI mesh rank 0 feeds its 7 peers over 7 distinct links.  There is no per-image communication.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults otherwise)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("VIRNET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Images [start, stop) of rank ``rank``: contiguous, sizes differ by at most one, every image owned exactly once."""
    if total < 0 or world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad shard request total={total} world={world} rank={rank}")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


@torch.no_grad()
def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> int:
    """Replace every parameter/buffer by rank ``src``'s values with ONE collective over a flat fp32 buffer.

    Returns the number of bytes broadcast.  A no-op (0) when torch.distributed is not initialised."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    tensors = list(module.state_dict().values())
    if not tensors:
        return 0
    flat = torch.cat([t.detach().reshape(-1).float() for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))   # in-place: bumps ._version, so packed weights are rebuilt
        off += n
    return flat.numel() * 4


def rank_topology(seen, world: int) -> dict:
    """``seen`` = one ``(device uuid, devices visible to that rank)`` pair per rank.  Returns the topology block of the benchmark line and
    raises when two ranks share a device ALTHOUGH every rank sees at least ``world`` devices -- a launcher that hides devices per rank or a
    wrong LOCAL_RANK must not produce a "scaling" number (fewer devices than ranks is the declared rehearsal mode of a 1-GPU box).
    Reference launch pattern: train_denoising_syn.py:280-297 (one process per visible GPU)."""
    uuids, counts = [u for u, _ in seen], [int(c) for _, c in seen]
    if len(seen) != world:
        raise ValueError(f"rank_topology: {len(seen)} reports for {world} ranks")
    topo = {"ranks": world, "ranks_seen": len(set(uuids)), "devices_visible_per_rank": counts, "uuids": uuids}
    if topo["ranks_seen"] < world and min(counts) >= world:
        raise RuntimeError(f"{world} ranks landed on {topo['ranks_seen']} distinct devices although every rank sees >= {world} devices ({uuids}): "
                           "refusing to report a multi-GPU number (check LOCAL_RANK / HIP_VISIBLE_DEVICES)")
    return topo


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    """MAX all-reduce of a python float (the benchmark's wall time)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_shards(local: torch.Tensor, total: int) -> Optional[torch.Tensor]:
    """Optional result collection: all ranks receive the full [total, ...] batch (excluded from images/s)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [shard_range(total, world, r) for r in range(world)]
    mx = max(b - a for a, b in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: b - a] for o, (a, b) in zip(outs, sizes)], 0)


class GradientReducer:
    """Bucketed, asynchronous gradient averaging for the training step (the job DistributedDataParallel does at
    ``train_denoising_syn.py:71``, shaped for this path).

    The whole network is ONE autograd Function here, so torch's per-parameter hooks would all fire at the end of the backward and
    nothing would overlap.  Instead the backward hands every layer's gradients to ``push`` as soon as its wgrad kernels are
    queued; they are scaled by 1/world into a flat fp32 bucket and a bucket is all-reduced (``async_op=True``: RCCL runs it on its
    own stream behind the work already queued) while the remaining layers' dgrad/wgrad kernels execute.  ``finish`` waits and returns
    COPIES of the bucket slices as the gradients (a p.grad aliasing a bucket would be overwritten by the next step's push).  Buckets are laid out in the order the first backward produced the gradients.
    On the xGMI mesh the 42 MB of denoise-syn gradients are 6 buckets of <= 8 MB; RCCL chooses the algorithm per message.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 8 << 20, group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.bucket_bytes = int(bucket_bytes)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._order: List[torch.nn.Parameter] = []          # production order seen during the first step
        self._slots: Optional[Dict[int, Tuple[int, int, int]]] = None   # id(param) -> (bucket, offset, numel)
        self._buckets: List[torch.Tensor] = []
        self._remaining: List[int] = []
        self._members: List[int] = []
        self._works: list = []
        self._observing = True                              # first step: record the production order, re-lay the buckets after it

    # -- bucket layout ------------------------------------------------------------------------------------------------
    def _build(self, device: torch.device, order_hint: Optional[List[torch.nn.Parameter]] = None) -> None:
        base = self._order if order_hint is None else order_hint
        known = {id(p) for p in base}
        order = list(base) + [p for p in self.params if id(p) not in known]       # never-produced parameters go last
        self._slots, sizes = {}, []
        cur, used = 0, 0
        for p in order:
            n = p.numel()
            if used and (used + n) * 4 > self.bucket_bytes:
                sizes.append(used)
                cur, used = cur + 1, 0
            self._slots[id(p)] = (cur, used, n)
            used += n
        sizes.append(used)
        self._buckets = [torch.zeros(n, dtype=torch.float32, device=device) for n in sizes]
        self._members = [0] * len(sizes)
        for b, _, _ in self._slots.values():
            self._members[b] += 1

    @property
    def bucket_sizes(self) -> List[int]:
        return [int(b.numel()) for b in self._buckets]

    # -- per step -----------------------------------------------------------------------------------------------------
    def start(self) -> None:
        self._works = []
        if self._slots is None:
            # first step: the production order is not known yet -- lay the buckets out in REVERSE registration order (the guess
            # torch's DDP makes too: the backward visits the modules roughly back to front), so step 1 already overlaps; the
            # observed order replaces it after the step
            self._build(self.params[0].device, order_hint=list(reversed(self.params)))
        self._remaining = list(self._members)
        for b in self._buckets:
            b.zero_()

    def push(self, grads: Dict[torch.nn.Parameter, torch.Tensor]) -> None:
        """Scale the gradients into their bucket slots ON THE CALLER'S CURRENT STREAM (the weight-gradient side stream during the
        backward) and start the all-reduce of every bucket that became complete; the collective is ordered after that stream."""
        scale = 1.0 / self.world
        for p, g in grads.items():
            if g is None or not p.requires_grad:
                continue
            if self._observing:
                self._order.append(p)
            b, off, n = self._slots[id(p)]
            self._buckets[b][off:off + n].copy_(g.reshape(-1)).mul_(scale)
            self._remaining[b] -= 1
            if self._remaining[b] == 0 and self.world > 1:
                self._works.append(dist.all_reduce(self._buckets[b], group=self.group, async_op=True))

    def finish(self) -> Dict[torch.nn.Parameter, torch.Tensor]:
        if self.world > 1:
            for b, left in enumerate(self._remaining):    # buckets holding parameters nobody produced this step
                if left > 0:
                    self._works.append(dist.all_reduce(self._buckets[b], group=self.group, async_op=True))
                    self._remaining[b] = 0
        for w in self._works:
            w.wait()
        self._works = []
        # Copies, not bucket views: autograd's AccumulateGrad adopts what it is handed as p.grad, and a p.grad aliasing a bucket
        # is rewritten by the NEXT step's push() before `p.grad += new` runs (zero_grad(set_to_none=False) and gradient
        # accumulation would then see 2x the new gradient).  42 MB of device copies per step are ~20 us on this GPU.
        out = {}
        for p in self.params:
            b, off, n = self._slots[id(p)]
            out[p] = self._buckets[b][off:off + n].clone().view_as(p)
        if self._observing:                               # from step 2 on: buckets in the order the backward really produced them
            self._observing = False
            self._build(self._buckets[0].device)
        return out


class DistributedTrainer(torch.nn.Module):
    """``DistributedDataParallel``-shaped wrapper for the drop-in modules: broadcasts rank 0's parameters once, then averages the
    gradients of every backward through a :class:`GradientReducer` that overlaps the collectives with the backward kernels.

        net = DistributedTrainer(VIRAttResUNet(...).cuda(rank))      # train_denoising_syn.py:71
        mu, sigma = net(x); loss.backward(); optimizer.step()
    """

    def __init__(self, module: torch.nn.Module, bucket_bytes: int = 8 << 20, group=None):
        super().__init__()
        self.module = module
        broadcast_parameters(module, src=0, group=group)
        self.reducer = GradientReducer(module.parameters(), bucket_bytes=bucket_bytes, group=group)
        module._grad_reducer = self.reducer

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
