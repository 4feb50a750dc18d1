"""Multi-GPU inference plumbing: one process per GPU, images sharded, weights broadcast once.

The path shards by independent images (no batch statistics, no cross-sample op -- SURVEY.md 8e), so the only collective
is ONE flat fp32 weight broadcast at start-up (42 MB for the denoise-syn net).  On ROCm ``backend="nccl"`` is RCCL; on the
fully connected xGMI mesh rank 0 feeds its 7 peers over 7 distinct links.  There is no per-image communication.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

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
