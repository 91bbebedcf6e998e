"""Batch sharding over GPUs: one process per GPU, trajectories are independent units, so the
only collective is the gather of per-trajectory results and the max-over-ranks of the timing
(SURVEY.md §8e).  Backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` units for `rank`; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_world() -> tuple[int, int, int]:
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def init(backend: str | None = None) -> tuple[int, int, int]:
    rank, local_rank, world = env_world()
    # MPCG_DIST_FORCE=1: create the process group even at world size 1, so that the RCCL collectives of this module run
    # (a communicator of one) on a single-GPU box instead of being dead code there
    force = os.environ.get("MPCG_DIST_FORCE") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:      # MPCG_DIST_BACKEND=gloo: dry-run the N>1 flow where only one GPU exists
            backend = os.environ.get("MPCG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def barrier():
    if dist.is_initialized():
        dist.barrier()


def backend_name() -> str:
    if not dist.is_initialized():
        return "none (single process)"
    b = dist.get_backend()
    return "nccl (RCCL)" if b == "nccl" else b


def _coll_device(device):
    """Collectives run on the GPU with nccl (RCCL), on the host with gloo."""
    return device if dist.is_initialized() and dist.get_backend() == "nccl" else "cpu"


def max_over_ranks(value: float, device="cpu") -> float:
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def all_gather_floats(value: float, device="cpu") -> list[float]:
    """One float of every rank, in rank order (per-rank kernel times of the bench line)."""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def gather_results(iters: torch.Tensor, exits: torch.Tensor, total: int | None) -> tuple[torch.Tensor, torch.Tensor]:
    """All-gather the per-trajectory (iters, max_iter_exit) of every rank's shard into global
    order.  total = N: shards follow shard_range(N, rank, world) and may be ragged, so each rank pads to the
    largest shard; total = None: every rank holds the same number of trajectories (weak scaling)."""
    if not dist.is_initialized():
        return iters.clone(), exits.clone()
    world = dist.get_world_size()
    if total is None:
        total = iters.numel() * world
    cap = max(shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world))
    cdev = _coll_device(iters.device)
    buf = torch.zeros(cap, 2, dtype=torch.int32, device=cdev)
    buf[: iters.numel(), 0] = iters.to(torch.int32)
    buf[: exits.numel(), 1] = exits.to(torch.int32)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    it, ex = [], []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        it.append(out[r][: hi - lo, 0])
        ex.append(out[r][: hi - lo, 1])
    return torch.cat(it).to(iters.device), torch.cat(ex).to(torch.uint8).to(iters.device)
