"""Data-parallel ray sharding.  One process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).  The reference has no multi-GPU path; the semantics follow the
only data-parallel precedent in the repo (RegNeRF/train.py:246-274): every rank renders its contiguous
slice of the step's ray batch, gradients are summed with ONE all-reduce of the flat fp32 gradient buffer
(4.77 MB at D=8/W=256 coarse+fine), scaled by 1/world, THEN clipped and applied by the replicated optimizer.

Masked losses are means over data-dependent sets, so the two set sizes are all-reduced first
(`global_mask_counts`) and each rank normalises by the GLOBAL counts; with that, an N-rank step equals the
1-rank step on the concatenated batch up to summation order."""
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the process group if
    WORLD_SIZE>1.  MASTER_ADDR should be 127.0.0.1 on a single node."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def shard_bounds(n: int, r: Optional[int] = None, w: Optional[int] = None) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of an n-ray batch owned by rank r of w (remainder to the first ranks)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    base, rem = divmod(n, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def shard_batch(*tensors, dim: int = 0):
    """Slice every tensor along `dim` to this rank's rays (all ranks hold the identical global batch, built
    from the identically seeded ray bank)."""
    out = []
    for t in tensors:
        if t is None:
            out.append(None)
            continue
        lo, hi = shard_bounds(t.shape[dim])
        out.append(t.narrow(dim, lo, hi - lo))
    return out[0] if len(out) == 1 else tuple(out)


def allreduce_mean_(flat_grad: torch.Tensor) -> torch.Tensor:
    """Sum the flat gradient over ranks, scale by 1/world.  One collective per step.  Use when every rank's
    loss is already a GLOBAL mean contribution scaled by world (plain per-rank means of equal shards)."""
    if world() > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        flat_grad.mul_(1.0 / world())
    return flat_grad


def allreduce_sum_(flat_grad: torch.Tensor) -> torch.Tensor:
    """Sum only: for losses whose per-ray weights were already normalised by global counts."""
    if world() > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


def global_mask_counts(mask: torch.Tensor) -> torch.Tensor:
    """(n_masked, n_unmasked) over ALL ranks as a 2-float tensor (8-byte all-reduce), for hardmask_losses."""
    m = mask.reshape(-1)
    c = torch.stack([(m == 1).sum(), (m == 0).sum()]).to(torch.float32)
    if world() > 1:
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return c


def allreduce_scalar_sum(x: torch.Tensor) -> torch.Tensor:
    if world() > 1:
        x = x.clone()
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
    return x


def barrier():
    if world() > 1:
        dist.barrier()
