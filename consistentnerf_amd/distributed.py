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
    # CNERF_FORCE_DIST=1 builds the group even for a single rank, so that the RCCL init / all-reduce / barrier calls can be
    # exercised on a 1-GPU box (collectives over one rank are identities)
    if (world > 1 or os.environ.get("CNERF_FORCE_DIST") == "1") and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
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
    if dist.is_initialized():
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        if world() > 1:
            flat_grad.mul_(1.0 / world())
    return flat_grad


def allreduce_sum_(flat_grad: torch.Tensor) -> torch.Tensor:
    """Sum only: for losses whose per-ray weights were already normalised by global counts."""
    if dist.is_initialized():
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


class GradReducer:
    """The step's ONE gradient exchange, issued in contiguous slices of FusedAdam's flat fp32 gradient as they become
    final, so that a slice travels over xGMI while the rest of the backward still computes.

    Each network's parameters are one contiguous slice of the flat buffer.  `_MlpFn.backward` (run_nerf.py) reports a
    network when the LAST of its pending backward nodes has accumulated into the buffer (a network evaluated twice in a
    step — one net for both levels R:402, the second render of ss_consistency — is reported once, after both); the slice's
    all-reduce is issued right there with async_op=True: RCCL's stream waits for the wgrad reduction that produced the
    slice (an event on the launch stream), the launch stream carries on with the other network's backward.
    `finish()` — call it between loss.backward() and optimizer.step() — issues whatever was not reported (networks
    whose backward did not run still contribute zeros on this rank), makes the launch stream wait for the collectives
    and applies the 1/world scale (`mean=True`: per-rank losses are means over equal shards; `mean=False`: per-ray weights
    were already normalised by global counts, `global_mask_counts`).  With the coarse and fine backward merged into one
    dgrad and one wgrad grid (the default when both networks are FusedAdam-owned, run_nerf._MlpFn), both slices become
    final together and go out as two back-to-back messages (2 x 2.38 MB at C2); with CNERF_MERGE_BWD=0 the fine slice's
    exchange overlaps the coarse network's backward.  Without an initialised process group everything is a no-op.

    Timing (bench.py): `last_exposed_ms()` = HIP-event time on the launch stream from the moment the last slice was
    issued (all gradient compute queued) to the moment the launch stream may proceed — the part of the exchange the step
    actually waits for."""

    def __init__(self, optimizer, modules, mean: bool = True, timing: bool = False):
        self.opt, self.mean, self.timing = optimizer, mean, timing
        self.modules = [m for m in modules if m is not None]
        self.slices = {}
        for m in self.modules:
            self.slices[id(m)] = optimizer.slice_of(list(m.parameters()))
            m._cnerf_reducer = self
            m._cnerf_pending = 0
        self._works, self._done, self._ev = [], set(), None
        self.bytes_per_step = 4 * sum(hi - lo for lo, hi in self.slices.values())
        self.exposed = []

    def _issue(self, m):
        lo, hi = self.slices[id(m)]
        self._done.add(id(m))
        if not dist.is_initialized():
            return
        if self.timing and len(self._done) == len(self.modules):
            self._ev = torch.cuda.Event(enable_timing=True)
            self._ev.record(torch.cuda.current_stream())
        self._works.append(dist.all_reduce(self.opt.flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def network_ready(self, m):
        """called by _MlpFn.backward when network `m` has no backward node pending in this step"""
        if id(m) in self.slices and id(m) not in self._done:
            self._issue(m)

    def finish(self):
        for m in self.modules:
            if id(m) not in self._done:
                self._issue(m)
            m._cnerf_pending = 0
        for w in self._works:
            w.wait()                 # the launch stream waits for RCCL's stream; the host does not block
        if self.timing and self._ev is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream())
            self.exposed.append((self._ev, e1))
        if self.mean and world() > 1:
            self.opt.flat_grad.mul_(1.0 / world())
        self._works, self._done, self._ev = [], set(), None

    def exposed_ms(self):
        return [a.elapsed_time(b) for a, b in self.exposed]


def global_mask_counts(mask: torch.Tensor) -> torch.Tensor:
    """(n_masked, n_unmasked) over ALL ranks as a 2-float tensor (8-byte all-reduce), for hardmask_losses."""
    m = mask.reshape(-1)
    c = torch.stack([(m == 1).sum(), (m == 0).sum()]).to(torch.float32)
    if dist.is_initialized():
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return c


def allreduce_scalar_sum(x: torch.Tensor) -> torch.Tensor:
    if world() > 1:
        x = x.clone()
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
    return x


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


# ---- inference: a frame's rows sharded over ranks (precedent RegNeRF/internal/models.py:311-322) ----------
def row_block(H: int, r: Optional[int] = None, w: Optional[int] = None):
    """Rank r renders image rows [lo, hi); every rank renders `rows` = ceil(H/w) rows so that the gather is
    rectangular: the `rows - (hi - lo)` extra ones repeat the block's last row (edge padding) and are dropped when
    the frame is reassembled.  Returns (lo, hi, row_index[rows])."""
    r = rank() if r is None else r
    w = world() if w is None else w
    lo, hi = shard_bounds(H, r, w)
    rows = -(-H // w)
    idx = torch.arange(lo, lo + rows).clamp_(max=max(hi - 1, lo)).clamp_(max=H - 1)
    return lo, hi, idx


def gather_rows(block: torch.Tensor, H: int) -> torch.Tensor:
    """All-gather of the per-rank row blocks [rows, W, ...] -> the frame [H, W, ...] on every rank (padding rows
    dropped).  The only collective of the render path, one per output per frame."""
    w = world()
    if w == 1:
        return block[:H]
    parts = [torch.empty_like(block) for _ in range(w)]
    dist.all_gather(parts, block.contiguous())
    keep = []
    for r, p in enumerate(parts):
        lo, hi = shard_bounds(H, r, w)
        keep.append(p[:hi - lo])
    return torch.cat(keep, 0)


def render_image_sharded(H, W, K, chunk, c2w, render_kwargs, render_fn=None, get_rays_fn=None):
    """One frame of `render_path` (R:156) with its rows split over the ranks: each rank renders its row block through
    the unchanged `render(rays=...)` (so viewdirs / NDC are derived exactly as for the full frame), then the blocks
    are all-gathered.  Returns (rgb [H,W,3], disp [H,W]) on every rank — rays are independent, so the frame equals the
    single-GPU one bit for bit."""
    if render_fn is None or get_rays_fn is None:
        from . import run_nerf, run_nerf_helpers
        render_fn = render_fn or run_nerf.render
        get_rays_fn = get_rays_fn or run_nerf_helpers.get_rays
    lo, hi, idx = row_block(H)
    rays_o, rays_d = get_rays_fn(H, W, K, c2w)
    idx = idx.to(rays_o.device)
    rays = torch.stack([rays_o[idx], rays_d[idx]], 0)            # [2, rows, W, 3]
    out = render_fn(H, W, K, chunk=chunk, rays=rays, **render_kwargs)
    rgb, disp = out[0], out[1]
    return gather_rows(rgb, H), gather_rows(disp, H)


def render_path_sharded(render_poses, hwf, K, chunk, render_kwargs, render_factor=0, render_fn=None, get_rays_fn=None):
    """`render_path` (R:140-178) over all ranks -> (rgbs [N,H,W,3], disps [N,H,W]) numpy on every rank."""
    import numpy as np
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    rgbs, disps = [], []
    for c2w in render_poses:
        with torch.no_grad():
            rgb, disp = render_image_sharded(H, W, K, chunk, c2w[:3, :4], render_kwargs, render_fn, get_rays_fn)
        rgbs.append(rgb.cpu().numpy())
        disps.append(disp.cpu().numpy())
    return np.stack(rgbs, 0), np.stack(disps, 0)
