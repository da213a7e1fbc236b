"""Optimiser tail of the training step as ONE kernel over flat buffers:
Adam(betas=(0.9,0.999), eps=1e-8) as created at R:210, optional clip_grad_value_ (V:1983) folded in,
the caller keeps driving `param_groups[i]['lr']` per step exactly like R:784-788.

Exposes the subset of torch.optim.Adam the reference driver touches: zero_grad(), step(), param_groups,
state_dict(), load_state_dict() (R:768,780,787,802) — the state dict uses torch.optim.Adam's layout so
checkpoints interchange.  Parameters are re-pointed into one flat fp32 buffer (same names / shapes /
values), gradients into a second one: that flat gradient is also the single RCCL all-reduce message of the
data-parallel path (distributed.py)."""
from typing import Iterable, List

import torch

from . import ops


# What a FusedAdam-owned parameter's view of the flat gradient holds: zeros nothing has written since / a gradient / a gradient
# that zero_grad() dropped (stale values the next writer must overwrite, or zero first).
GRAD_ZERO, GRAD_LIVE, GRAD_DROPPED = 0, 1, 2
GRAD_DETACHED = 3      # the caller set p.grad = None and autograd attached a tensor of its own: step() copies it into the view


def _materialize_on_tensor_route(p):
    """Hook of a FusedAdam-owned parameter: autograd computed a per-tensor gradient for it (AccumulateGrad is about to ADD it to
    the .grad view) — a view whose contents zero_grad() dropped has to be zero first."""
    def hook(g):
        if g is None:        # (the direct route hands autograd no gradient; the engine still runs the hook of the leaf)
            return None
        if p.grad is None or p.grad.data_ptr() != p._cnerf_view_ptr:
            # the caller detached the view (model.zero_grad(), p.grad = None): AccumulateGrad SETS a tensor of its own, and ADDS
            # to it on every further backward before step() (gradient accumulation) — the state stays DETACHED until
            # materialize_grad() / zero_grad() bring the view back (it used to flip to LIVE on the second backward,
            # after which the direct route overwrote the first gradient and Adam stepped on a stale flat buffer)
            p._cnerf_grad_state = GRAD_DETACHED
            return None
        if p._cnerf_grad_state == GRAD_DROPPED:
            with torch.no_grad():
                p.grad.zero_()
        p._cnerf_grad_state = GRAD_LIVE
        return None
    return hook


class FusedAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr=5e-4, betas=(0.9, 0.999), eps=1e-8,
                 clip_value: float = 0.0):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("FusedAdam got an empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise ops.CnerfError("FusedAdam needs GPU parameters (no CPU path)")
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps, clip_value=clip_value)]
        self._step = 0
        self._hyp_ring = self._hyp_ev = self._hyp_dev = None
        sizes = [p.numel() for p in self.params]
        self._offsets = [0]
        for n in sizes:
            self._offsets.append(self._offsets[-1] + n)
        total = self._offsets[-1]
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self.flat_param = torch.empty(total, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(total, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o, n in zip(self.params, self._offsets, sizes):
                self.flat_param[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.flat_param[o:o + n].view(p.shape)
                p.grad = self.flat_grad[o:o + n].view(p.shape)
                p._cnerf_direct_grad = True     # _MlpFn.backward accumulates into this view directly
                p._cnerf_view_ptr = p.grad.data_ptr()   # (the direct route is taken only while .grad IS this view)
                p._cnerf_grad_state = GRAD_ZERO
                if p.requires_grad:      # (a frozen parameter gets no gradient from either route)
                    p.register_hook(_materialize_on_tensor_route(p))

    def slice_of(self, params):
        """[lo, hi) of the flat buffers covered by `params` (e.g. one network's parameters); they must be contiguous in it."""
        idx = sorted(self._index[id(p)] for p in params)
        if idx != list(range(idx[0], idx[-1] + 1)):
            raise ValueError("parameters are not contiguous in the flat buffer")
        return self._offsets[idx[0]], self._offsets[idx[-1] + 1]

    # -- torch.optim surface -------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True):
        """set_to_none=True (torch.optim's own default): the gradient is DROPPED, not cleared — no fill launch.  The per-parameter
        .grad views stay attached (the direct-accumulate route needs them) but the contents of those a backward wrote are
        undefined until the next backward, whose wgrad reduction then overwrites the flat buffer instead of adding to it
        (run_nerf._MlpFn.backward); whatever else needs defined values first — the tensor route (a hook on every parameter),
        step() without a backward, the GradReducer's all-reduce of a network that had none — goes through materialize_grad().
        Views nothing has written since they were last zero (parameters no loss reaches) stay zero at no cost.
        set_to_none=False: zero now.  Code that reads `flat_grad` itself before step() (gradient-norm logging, a hand-written
        all-reduce) must call materialize_grad() first: until then the views of parameters no backward reached hold stale values."""
        for p, o in zip(self.params, self._offsets):
            if p.grad is None or p.grad.data_ptr() != p._cnerf_view_ptr:
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
                # (its gradient lived in autograd's own tensor, or nowhere: whatever the view holds is stale)
                if p._cnerf_grad_state in (GRAD_DETACHED, GRAD_LIVE):
                    p._cnerf_grad_state = GRAD_DROPPED
        if set_to_none:
            for p in self.params:
                if p._cnerf_grad_state == GRAD_LIVE:
                    p._cnerf_grad_state = GRAD_DROPPED
        else:
            self.flat_grad.zero_()
            for p in self.params:
                p._cnerf_grad_state = GRAD_ZERO

    def materialize_grad(self):
        """Gradients dropped by zero_grad() and not overwritten by a backward since become zeros (one fill per contiguous run)."""
        runs = []
        for p, o in zip(self.params, self._offsets):
            # keyed on WHERE .grad points, not on the state flag: a gradient that lives in a tensor of autograd's own (the caller
            # set p.grad = None before the backward) is copied into the view whatever the flag says
            if p.grad is None:
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
                if p._cnerf_grad_state in (GRAD_DETACHED, GRAD_LIVE):   # detached and no backward since: nothing to step on
                    p._cnerf_grad_state = GRAD_DROPPED
            elif p.grad.data_ptr() != p._cnerf_view_ptr:
                view = self.flat_grad[o:o + p.numel()].view(p.shape)
                with torch.no_grad():
                    view.copy_(p.grad)
                p.grad = view
                p._cnerf_grad_state = GRAD_LIVE
            elif p._cnerf_grad_state == GRAD_DETACHED:
                p._cnerf_grad_state = GRAD_LIVE
            if p._cnerf_grad_state == GRAD_DROPPED:
                p._cnerf_grad_state = GRAD_ZERO
                if runs and runs[-1][1] == o:
                    runs[-1][1] = o + p.numel()
                else:
                    runs.append([o, o + p.numel()])
        for lo, hi in runs:
            self.flat_grad[lo:hi].zero_()

    def step(self, grad_scale: float = 1.0):
        """One Adam step over the flat buffers.  `grad_scale` multiplies the gradient inside the kernel BEFORE the clip (the
        1/world of a summed data-parallel gradient rides here for free: distributed.GradReducer(fold_scale=True))."""
        g = self.param_groups[0]
        self.materialize_grad()
        if self._hyp_ring is not None:
            # graph-capturable form (graph.GraphedStep): the scalars of the step live in device memory.  Eagerly, advance()
            # uploads them (stream-ordered, from a ring of pinned buffers) right before the kernel; under capture only the
            # kernel is recorded and GraphedStep calls advance() before every replay.
            if not torch.cuda.is_current_stream_capturing():
                self.advance(grad_scale)
            ops.adam_step_dev(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self._hyp_dev)
        else:
            self._step += 1
            ops.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self._step, g['lr'],
                          g['betas'][0], g['betas'][1], g['eps'], g.get('clip_value', 0.0), grad_scale)
        self.bump_epoch()

    def bump_epoch(self):
        """The kernel wrote the weights behind autograd's back: invalidate the packed-weight caches keyed on the parameters
        (run_nerf._packed_gen).  GraphedStep calls this after every replay (the recorded Python of step() ran only once)."""
        for p in self.params:
            p._cnerf_epoch = getattr(p, "_cnerf_epoch", 0) + 1

    RING = 4    # pinned scalar blocks in flight: the host may run this many steps ahead of the GPU before advance() waits

    def make_capturable(self):
        """Switch to the device-scalar Adam kernel so that step() can sit inside a captured hipGraph."""
        if self._hyp_ring is None:
            self._hyp_ring = [torch.zeros(8, dtype=torch.float32).pin_memory() for _ in range(self.RING)]
            self._hyp_ev = [None] * self.RING
            self._hyp_dev = torch.zeros(8, device=self.flat_param.device, dtype=torch.float32)
        return self

    def advance(self, grad_scale: float = 1.0):
        """Host half of a capturable step: count it, write its scalars (incl. the current param_groups lr) to the next pinned
        block of the ring and enqueue the 32-byte upload on the CURRENT stream, i.e. ahead of the step's kernels (eager) or of
        the graph replay that follows (GraphedStep).  A block is only rewritten once the upload that last read it has
        completed (event), so a host running several steps ahead of the GPU can never hand step N the scalars of step N+k."""
        g = self.param_groups[0]
        self._step += 1
        i = self._step % self.RING
        if self._hyp_ev[i] is not None:
            self._hyp_ev[i].synchronize()
        ops.adam_hyper(self._hyp_ring[i], self._step, g['lr'], g['betas'][0], g['betas'][1], g['eps'], g.get('clip_value', 0.0),
                       grad_scale)
        self._hyp_dev.copy_(self._hyp_ring[i], non_blocking=True)
        ev = self._hyp_ev[i] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._hyp_ev[i] = ev

    def state_dict(self):
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self._offsets)):
            n = p.numel()
            state[i] = {'step': torch.tensor(float(self._step)),
                        'exp_avg': self.exp_avg[o:o + n].view(p.shape).clone(),
                        'exp_avg_sq': self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        g = self.param_groups[0]
        group = {k: v for k, v in g.items() if k != 'params'}
        group.update(params=list(range(len(self.params))), amsgrad=False, weight_decay=0, maximize=False)
        return {'state': state if self._step > 0 else {}, 'param_groups': [group]}

    def load_state_dict(self, sd):
        g = sd['param_groups'][0]
        self.param_groups[0]['lr'] = g['lr']
        self.param_groups[0]['betas'] = tuple(g.get('betas', (0.9, 0.999)))
        self.param_groups[0]['eps'] = g.get('eps', 1e-8)
        st = sd.get('state', {})
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self._offsets)):
                s = st.get(i, st.get(str(i)))
                if s is None:
                    continue
                n = p.numel()
                self.exp_avg[o:o + n].copy_(s['exp_avg'].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(s['exp_avg_sq'].reshape(-1))
                self._step = int(float(s['step']))
