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

    def __init__(self, optimizer, modules, mean: bool = True, timing: bool = False, fold_scale: bool = False):
        """fold_scale=True: finish() leaves the SUM in the flat gradient and the caller hands `grad_scale` (1/world when
        `mean`) to FusedAdam.step(grad_scale=...), where the multiply is free inside the Adam kernel (before its clip — the
        order of RegNeRF/train.py:246-274); the default applies it here with one more pass over the buffer, which is what a
        caller clipping the .grad views with torch's own clip_grad_value_ between finish() and step() needs."""
        self.opt, self.mean, self.timing, self.fold_scale = optimizer, mean, timing, fold_scale
        self.hold = False           # True: network_ready() is ignored and finish() issues every slice (graph.GraphedStep, split)
        self.modules = [m for m in modules if m is not None]
        self.slices = {}
        for m in self.modules:
            self.slices[id(m)] = optimizer.slice_of(list(m.parameters()))
            m._cnerf_reducer = self
            m._cnerf_pending = 0
        self._works, self._done, self._ev = [], set(), None
        self.bytes_per_step = 4 * sum(hi - lo for lo, hi in self.slices.values())
        self.exposed = []
        self.messages = self.steps = 0      # all-reduce calls issued / finish() calls: messages per step = their ratio

    def _issue(self, *ms):
        """all-reduce the slices of the networks `ms`; adjacent slices leave as ONE message (xGMI is point-to-point: fewer,
        larger messages — and every message costs two cross-stream hand-overs on the launch stream)"""
        spans = sorted(self.slices[id(m)] for m in ms)
        merged = [list(spans[0])]
        for lo, hi in spans[1:]:
            if lo == merged[-1][1]:
                merged[-1][1] = hi
            else:
                merged.append([lo, hi])
        for m in ms:
            self._done.add(id(m))
        if not dist.is_initialized():
            return
        if self.timing and len(self._done) == len(self.modules):
            self._ev = torch.cuda.Event(enable_timing=True)
            self._ev.record(torch.cuda.current_stream())
        for lo, hi in merged:
            self._works.append(dist.all_reduce(self.opt.flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            self.messages += 1

    @property
    def grad_scale(self) -> float:
        """what FusedAdam.step(grad_scale=) has to apply after finish() (1.0 unless fold_scale and mean and world > 1)"""
        return 1.0 / world() if (self.fold_scale and self.mean and world() > 1) else 1.0

    def network_ready(self, m):
        """called by _MlpFn.backward when network `m` has no backward node pending in this step"""
        if not self.hold and id(m) in self.slices and id(m) not in self._done:
            self._issue(m)

    def networks_ready(self, ms):
        """several networks became final together (the merged coarse+fine backward): one message for adjacent slices"""
        ms = [m for m in ms if id(m) in self.slices and id(m) not in self._done]
        if ms and not self.hold:
            self._issue(*ms)

    def reset(self):
        """forget the bookkeeping of a backward pass whose exchange is not going to be finished (a recorded, not executed one)"""
        for m in self.modules:
            m._cnerf_pending = 0
        self._works, self._done, self._ev = [], set(), None

    def finish(self):
        rest = [m for m in self.modules if id(m) not in self._done]
        if rest:
            if hasattr(self.opt, "materialize_grad"):
                self.opt.materialize_grad()  # (a network without a backward in this step contributes zeros, not a dropped buffer)
            self._issue(*rest)
        for m in self.modules:
            m._cnerf_pending = 0
        self.steps += 1
        for w in self._works:
            w.wait()                 # the launch stream waits for RCCL's stream; the host does not block
        if self.timing and self._ev is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream())
            self.exposed.append((self._ev, e1))
        if self.mean and world() > 1 and not self.fold_scale:
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
    """RegNeRF-style block of a frame's rows for rank r (RegNeRF/internal/models.py:311-322 pads the RAYS): rows [lo, hi) plus
    `ceil(H/w) - (hi - lo)` repeats of the block's last row (edge padding), as an index vector.  render_path_sharded pads the
    rendered OUTPUT block instead (same frame, nothing rendered twice); this form is what the tests' hand-made splits through
    render(rays=...) use.  Returns (lo, hi, row_index[rows])."""
    r = rank() if r is None else r
    w = world() if w is None else w
    lo, hi = shard_bounds(H, r, w)
    rows = -(-H // w)
    idx = torch.arange(lo, lo + rows).clamp_(max=max(hi - 1, lo)).clamp_(max=H - 1)
    return lo, hi, idx


def _pad_rows(t: torch.Tensor, n: int) -> torch.Tensor:
    """[m, ...] -> [n, ...] by repeating the last row (edge padding of the OUTPUT block: same frame as padding the rays the
    way RegNeRF/internal/models.py:311-322 does, without rendering the padding)."""
    if t.shape[0] == n:
        return t
    if t.shape[0] == 0:       # a rank without rows (H < world): its block is all padding
        return t.new_zeros((n,) + tuple(t.shape[1:]))
    return torch.cat([t, t[-1:].expand(n - t.shape[0], *t.shape[1:])], 0)


def gather_blocks_to_root(block: torch.Tensor, out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """ONE gather of the equal-sized per-rank blocks to rank 0 -> [world, *block.shape] there (written into `out` when given),
    None elsewhere.  RCCL gathers device buffers directly; gloo (the one-GPU tests) is staged through the host."""
    w = world()
    if w == 1:
        if out is None:
            return block[None]
        out[0].copy_(block)
        return out
    block = block.contiguous()
    if dist.get_backend() == "gloo" and block.is_cuda:
        hb = block.cpu()
        parts = [torch.empty_like(hb) for _ in range(w)] if rank() == 0 else None
        dist.gather(hb, parts, dst=0)
        if rank() != 0:
            return None
        full = torch.stack(parts, 0).to(block.device)
        if out is None:
            return full
        out.copy_(full)
        return out
    if rank() == 0:
        if out is None:
            out = torch.empty((w,) + tuple(block.shape), device=block.device, dtype=block.dtype)
        dist.gather(block, [out[r] for r in range(w)], dst=0)
        return out
    dist.gather(block, None, dst=0)
    return None


def _render_block(H, W, K, chunk, c2w, lo, hi, render_kwargs, render_fn, get_rays_fn, want_acc):
    """This rank's rows [lo, hi) of one frame -> packed [(hi-lo)*W, 4|5] = rgb | disp (| acc).  The stock query renders them
    through the in-kernel camera path (`run_nerf.render_pixels`: rays generated inside the kernels from `cam.first = lo*W`, no
    [H*W, 11] ray tensor on any rank); anything else (custom network_query_fn, per-ray near/far, DEBUG) takes the ray tensor of
    the block through the unchanged render(rays=...)."""
    from . import run_nerf
    if hi <= lo:
        p = next(render_kwargs["network_fn"].parameters()) if "network_fn" in render_kwargs else torch.zeros(())
        return p.new_zeros((0, 5 if want_acc else 4))
    if render_fn is None and run_nerf.camera_path_ok(c2w, render_kwargs):
        o = run_nerf.render_pixels(H, W, K, chunk, c2w, lo * W, (hi - lo) * W, **render_kwargs)
        cols = [o["rgb_map"], o["disp_map"][:, None]] + ([o["acc_map"][:, None]] if want_acc else [])
        return torch.cat(cols, 1)
    if render_fn is None or get_rays_fn is None:
        from . import run_nerf_helpers
        render_fn = render_fn or run_nerf.render
        get_rays_fn = get_rays_fn or run_nerf_helpers.get_rays
    rays_o, rays_d = get_rays_fn(H, W, K, c2w)
    rays = torch.stack([rays_o[lo:hi], rays_d[lo:hi]], 0)            # [2, rows, W, 3]
    out = render_fn(H, W, K, chunk=chunk, rays=rays, **render_kwargs)
    cols = [out[0].reshape(-1, 3), out[1].reshape(-1, 1)] + ([out[2].reshape(-1, 1)] if want_acc else [])
    return torch.cat(cols, 1)


def render_image_sharded(H, W, K, chunk, c2w, render_kwargs, render_fn=None, get_rays_fn=None, want_acc=False, out=None):
    """One frame of `render_path` (R:156) with its rows split over the ranks: every rank renders ITS rows only, the packed
    blocks (rgb | disp [| acc], the short ones edge-padded to ceil(H/world) rows so the gather is rectangular) go to rank 0
    in ONE gather.  Returns the frame [H, W, 4|5] on rank 0 (a view of `out` [world, rows*W, 4|5] when given), None on the
    other ranks — render_path only ever needs the frame on one host (R:157-159).  Rays are independent, so the frame equals
    the single-GPU one bit for bit."""
    lo, hi = shard_bounds(H)
    rows = -(-H // world())
    blk = _render_block(H, W, K, chunk, c2w, lo, hi, render_kwargs, render_fn, get_rays_fn, want_acc)
    full = gather_blocks_to_root(_pad_rows(blk, rows * W), out)
    if full is None:
        return None
    if world() == 1:
        return full[0].view(H, W, -1)
    keep = [full[r, :(shard_bounds(H, r)[1] - shard_bounds(H, r)[0]) * W] for r in range(world())]
    return torch.cat(keep, 0).view(H, W, -1)


def render_path_sharded(render_poses, hwf, K, chunk, render_kwargs, render_factor=0, render_fn=None, get_rays_fn=None,
                        want_acc=False, frame_times=None):
    """`render_path` (R:140-178; V:252-294 with want_acc) over all ranks -> (rgbs [N,H,W,3], disps [N,H,W][, accs]) numpy on
    rank 0, None elsewhere.  Frame i's device-to-host copy runs on its own stream into one of two pinned buffers while frame
    i+1 renders; `frame_times` (a list) receives rank 0's host time per frame."""
    import time
    import numpy as np
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    nch = 5 if want_acc else 4
    root = rank() == 0
    cuda = torch.cuda.is_available()
    frames, pending = [], []
    copy_stream = torch.cuda.Stream() if (cuda and root) else None
    host = [None, None]

    def drain(k):
        ev, buf = pending.pop(k)
        ev.synchronize()
        frames.append(np.array(buf.numpy(), copy=True))

    for i, c2w in enumerate(render_poses):
        t0 = time.perf_counter()
        with torch.no_grad():
            frame = render_image_sharded(H, W, K, chunk, c2w[:3, :4], render_kwargs, render_fn, get_rays_fn, want_acc)
        if root:
            if copy_stream is None or not frame.is_cuda:
                frames.append(frame.cpu().numpy())
            else:
                if len(pending) == 2:       # both pinned buffers busy: the older copy finished a frame ago
                    drain(0)
                j = i & 1
                if host[j] is None:
                    host[j] = torch.empty((H, W, nch), dtype=torch.float32).pin_memory()
                copy_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(copy_stream):
                    host[j].copy_(frame, non_blocking=True)
                    frame.record_stream(copy_stream)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                pending.append((ev, host[j]))
        if frame_times is not None:
            frame_times.append(time.perf_counter() - t0)
    while pending:
        drain(0)
    if not root:
        return None
    full = np.stack(frames, 0)
    out = (np.ascontiguousarray(full[..., :3]), np.ascontiguousarray(full[..., 3]))
    return out + ((np.ascontiguousarray(full[..., 4]),) if want_acc else ())
