"""Drop-in for the render / model-construction surface of the reference's run_nerf.py (R):

  batchify R:27 | run_network R:37 | batchify_rays R:55 | render R:70 | render_path R:140 |
  create_nerf R:181 | raw2outputs R:265 | render_rays R:311

Same signatures, kwargs dictionaries and return structures; the bodies launch the gfx950 kernels of
libcnerf_hip.so (include/cnerf.h).  Autograd is wired with two torch.autograd.Function nodes (fused
encoding+MLP, compositing); sampling carries no gradient, exactly like the reference (R:397).

`run_nerf_view.py` of this package layers the ConsistentNeRF additions (depth outputs, warp, hard masks,
masked losses) on top of this module.
"""
import os
import time
import weakref

import numpy as np
import torch

from . import ops
from .optim import FusedAdam
from .run_nerf_helpers import (NeRF, get_embedder, get_rays, img2mse, mse2psnr, ndc_coefficients,  # noqa: F401
                               get_rays_np, ndc_rays, pytest_uniform, sample_pdf, sample_u, to8b)

DEBUG = False


# ----------------------------------------------------------------------------- autograd nodes
class RayPoints:
    """Lazy stand-in for the [N_rays, N_samples, 3] point tensor of R:384: the fused kernel evaluates
    o + d*z itself, so the 9.4 MB/step point cloud is never written.  `.materialize()` gives the tensor."""

    def __init__(self, rays, z_vals, live=None, skip=0):
        self.rays, self.z_vals = rays, z_vals
        self.live = live          # device int32 [1]: only the first live[0] rays are real (a batch padded to a fixed capacity)
        self.skip = skip          # (with live) the first `skip` rays get zero seeds: the backward of this query may leave them out
        self.shape = torch.Size([z_vals.shape[0], z_vals.shape[1], 3])
        self.device = z_vals.device

    def materialize(self):
        return self.rays[:, None, 0:3] + self.rays[:, None, 3:6] * self.z_vals[..., None]


def _pack_state(model):
    """(kernel tensors, their identity / version key, the cached (key, buffer, generation, [buffer 0, buffer 1]) or None)"""
    ts = model.kernel_tensors()
    key = tuple((t.data_ptr(), t._version, getattr(t, "_cnerf_epoch", 0)) for t in ts)
    return ts, key, model.__dict__.get("_cnerf_packed")


def _packed_gen(model):
    """Kernel-layout weights of `model`, re-packed only when a parameter changed (optimizer step / load).  Two buffers
    alternate, so the copy a forward pass used stays intact for its backward without a per-step clone: it is overwritten
    by the SECOND re-pack after it — two parameter updates between a forward and its backward, where the reference
    itself fails (autograd's version check on the modified weights).  Returns (buffer, generation)."""
    ts, key, cache = _pack_state(model)
    if cache is None or cache[0] != key:
        gen = 0 if cache is None else cache[2] + 1
        bufs = [None, None] if cache is None else cache[3]
        with torch.no_grad():
            bufs[gen & 1] = ops.pack_weights(model.spec(), ts, bufs[gen & 1])
        cache = (key, bufs[gen & 1], gen, bufs)
        model.__dict__["_cnerf_packed"] = cache
    return cache[1], cache[2]


def _packed(model):
    return _packed_gen(model)[0]


def _prepack_pair(model_a, model_b):
    """render_rays is about to query both networks: when BOTH kernel-layout copies are stale (every training step: the optimizer
    just wrote both) re-pack them with one launch instead of one each.  Same cache protocol as _packed_gen."""
    if model_a is None or model_b is None or model_a is model_b:
        return
    st = []
    for m in (model_a, model_b):
        if not hasattr(m, "kernel_tensors"):
            return
        ts, key, cache = _pack_state(m)
        if cache is not None and cache[0] == key:
            return                               # (at most one is stale: its own query re-packs it)
        gen = 0 if cache is None else cache[2] + 1
        bufs = [None, None] if cache is None else cache[3]
        st.append((m, ts, key, gen, bufs))
    if st[0][1][0].device != st[1][1][0].device or not st[0][1][0].is_cuda:
        return
    (ma, tsa, ka, ga, ba), (mb, tsb, kb, gb, bb) = st
    with torch.no_grad():
        ba[ga & 1], bb[gb & 1] = ops.pack_weights_pair(ma.spec(), tsa, ba[ga & 1], mb.spec(), tsb, bb[gb & 1])
    ma.__dict__["_cnerf_packed"] = (ka, ba[ga & 1], ga, ba)
    mb.__dict__["_cnerf_packed"] = (kb, bb[gb & 1], gb, bb)


def _packed_bf_gen(model, planes):
    """bf16-plane panels of `model` (opt-in reduced-precision forward / the bf16x3 training kernels), re-packed when a parameter
    changed.  Same two-buffer / generation protocol as _packed_gen: the copy a training forward used survives ONE optimizer step
    before its backward (forward A, step, forward B, backward A is legal with the fp32 panels, so it is here); a second re-pack
    overwrites it and the backward refuses (_packed_bf_still_valid).  Returns (buffer, generation)."""
    ts = model.kernel_tensors()
    key = (planes,) + tuple((t.data_ptr(), t._version, getattr(t, "_cnerf_epoch", 0)) for t in ts)
    cache = model.__dict__.get("_cnerf_packed_bf")
    if cache is None or cache[0] != key:
        same_planes = cache is not None and cache[0][0] == planes
        gen = cache[2] + 1 if cache is not None else 0          # (monotonic also across a change of plane count)
        bufs = cache[3] if same_planes else [None, None]
        with torch.no_grad():
            bufs[gen & 1] = ops.pack_weights_bf(model.spec(), ts, planes, bufs[gen & 1])
        cache = (key, bufs[gen & 1], gen, bufs)
        model.__dict__["_cnerf_packed_bf"] = cache
    return cache[1], cache[2]


def _packed_bf(model, planes):
    return _packed_bf_gen(model, planes)[0]


def _packed_bf_still_valid(model, gen, buf):
    cache = model.__dict__.get("_cnerf_packed_bf")
    return cache is not None and cache[2] - gen <= 1 and any(b is buf for b in cache[3])


# Training arithmetic: "fp32" (default: exact fp32 MFMA, the arithmetic every parity statement and the headline bench line are
# made in) or "bf16x3" (opt-in: bf16 matrix cores, three planes per operand, fp32 accumulation — reported as a SECOND bench line
# only).  Per model (`NeRF.training_precision`), or for the whole process with CNERF_TRAIN_PRECISION (what the GPU suite uses to
# run every test in that arithmetic as well).
DEFAULT_TRAINING_PRECISION = os.environ.get("CNERF_TRAIN_PRECISION", "fp32")


def training_precision(model):
    """An explicit `model.training_precision` is obeyed (an architecture the bf16x3 kernels are not compiled for then fails
    loudly in the launch); the process-wide default only applies to architectures they cover (view directions, W = 128 | 256,
    the 10 / 4-frequency encodings) and leaves every other network on the exact-fp32 path."""
    prec = getattr(model, "training_precision", None)
    if prec is None:
        prec = DEFAULT_TRAINING_PRECISION
        if prec == "bf16x3":
            sp = model.spec()
            if not (sp.use_viewdirs and sp.W in (128, 256) and sp.multires == 10 and sp.multires_views == 4):
                prec = "fp32"
    if prec not in ("fp32", "bf16x3"):
        raise ValueError("training_precision must be 'fp32' or 'bf16x3'")
    return prec


def _packed_still_valid(model, gen):
    cache = model.__dict__.get("_cnerf_packed")
    return cache is not None and cache[2] - gen <= 1


def _probe_engine_query():
    """The direct-accumulate route and the merged coarse+fine backward ask the autograd engine, from inside a backward node,
    whether another node is going to run in the same pass: `torch._C._will_engine_execute_node` — a PRIVATE symbol.  This
    probe runs once at import, on the CPU: the symbol must exist, answer True for a leaf's AccumulateGrad node under
    .backward(), and refuse (RuntimeError) or answer False under torch.autograd.grad().  Anything else -> (None, reason) and
    the module selects the plain route explicitly: per-tensor gradients handed back to autograd, one backward per level."""
    fn = getattr(torch._C, "_will_engine_execute_node", None)
    if fn is None:
        return None, "torch._C._will_engine_execute_node does not exist in this torch"
    seen = []

    class _Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.w = w
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            with torch.enable_grad():
                acc = ctx.w.view_as(ctx.w).grad_fn.next_functions[0][0]
            try:
                seen.append(bool(fn(acc)))
            except RuntimeError:
                seen.append("refused")
            return g, None

    try:
        w = torch.ones(2, requires_grad=True)
        _Probe.apply(w * 2.0, w).sum().backward()
        torch.autograd.grad(_Probe.apply(w * 2.0, w).sum(), [w])
    except Exception as e:   # noqa: BLE001 — any failure of the probe selects the plain route
        return None, f"probing torch._C._will_engine_execute_node raised {type(e).__name__}: {e}"
    if len(seen) != 2 or seen[0] is not True or seen[1] not in (False, "refused"):
        return None, f"torch._C._will_engine_execute_node answered {seen} (expected [True, False | refused])"
    return fn, None


_ENGINE_QUERY, _ENGINE_QUERY_WHY = _probe_engine_query()
if _ENGINE_QUERY is None:
    import warnings
    warnings.warn("consistentnerf_amd: " + _ENGINE_QUERY_WHY + " — training falls back to the plain autograd route "
                  "(per-tensor gradients returned to autograd, coarse and fine backward launched separately); results are "
                  "identical, the step is ~2 % slower", RuntimeWarning)


def _engine_accumulates(p):
    """True when the running backward pass will ACCUMULATE into p.grad (loss.backward(), also with inputs=[...]); False
    under torch.autograd.grad(), where the engine captures the gradient of a leaf instead (it refuses the query for a
    leaf's AccumulateGrad node in that mode — that refusal is the signal), and False when the engine query is unusable in
    this torch (_probe_engine_query)."""
    if _ENGINE_QUERY is None:
        return False
    with torch.enable_grad():
        acc = p.view_as(p).grad_fn.next_functions[0][0]
    try:
        return bool(_ENGINE_QUERY(acc))
    except RuntimeError:
        return False


# The coarse and the fine network's backward passes share nothing once the forward is done (z is detached at R:397): when
# both are FusedAdam-owned they run as ONE dgrad grid + ONE wgrad grid (cnerf_mlp_bwd_pair) instead of two of each — the
# 8-round coarse launches otherwise pay their own ramp and tail.  CNERF_MERGE_BWD=0 keeps them separate.
MERGE_BWD = os.environ.get("CNERF_MERGE_BWD", "1") != "0"


class _LevelPair:
    """Links the coarse and the fine _MlpFn node of one render_rays call.  The fine node runs first in the backward pass
    (it was created last); if the engine is going to run the coarse node too, it parks its inputs here and the coarse node
    launches both."""
    __slots__ = ("coarse", "fine", "parked")

    def __init__(self, coarse_node, fine_node):
        self.coarse, self.fine, self.parked = weakref.ref(coarse_node), weakref.ref(fine_node), None


def _link_levels(raw_coarse, raw_fine):
    nc, nf = raw_coarse.grad_fn, raw_fine.grad_fn
    if not MERGE_BWD or _ENGINE_QUERY is None or nc is None or nf is None:
        return
    if getattr(nc, "stash", None) is None or getattr(nf, "stash", None) is None or nc.model is nf.model:
        return
    nc.pair = nf.pair = _LevelPair(nc, nf)


def _direct_ok(needs, params):
    # .grad must BE FusedAdam's view of the flat gradient: after `p.grad = None` autograd attaches a tensor of its own, and a
    # direct write into that one would bypass (and, on the overwrite route, destroy) what the optimiser steps on
    return all(needs) and all(getattr(p, "_cnerf_direct_grad", False) and p.grad is not None
                              and p.grad.data_ptr() == p._cnerf_view_ptr and p.grad.is_contiguous() for p in params)


def _take_dropped(*param_lists):
    """The direct route is about to write these parameters' views of FusedAdam's flat gradient.  True when none of them holds a
    live gradient (all dropped by zero_grad(), or still zero): the wgrad reduction then OVERWRITES (accumulate=0) and the fill
    launch of an eager zero_grad never happens.  In a mixed state the dropped views are zeroed here and the reduction adds."""
    from .optim import GRAD_DETACHED, GRAD_DROPPED, GRAD_LIVE
    ps = [p for params in param_lists for p in params]
    fresh = all(p._cnerf_grad_state not in (GRAD_LIVE, GRAD_DETACHED) for p in ps)
    for p in ps:
        if not fresh and p._cnerf_grad_state == GRAD_DROPPED:
            p.grad.zero_()
        p._cnerf_grad_state = GRAD_LIVE
    return fresh


def _report_ready(model, direct):
    """One backward node of `model` has accumulated into the flat gradient: tell the GradReducer when it was the last."""
    if hasattr(model, "_cnerf_pending"):
        model._cnerf_pending -= 1
        if direct and model._cnerf_pending <= 0:
            model._cnerf_reducer.network_ready(model)


def _report_ready_pair(model_a, model_b):
    """The merged backward finished BOTH networks at once: report them together, so that a reducer that owns both sends their
    (adjacent) slices of the flat gradient as ONE message instead of two back-to-back ones."""
    ready = []
    for m in (model_a, model_b):
        if hasattr(m, "_cnerf_pending"):
            m._cnerf_pending -= 1
            if m._cnerf_pending <= 0:
                ready.append(m)
    if len(ready) == 2 and ready[0]._cnerf_reducer is ready[1]._cnerf_reducer:
        ready[0]._cnerf_reducer.networks_ready(ready)
    else:
        for m in ready:
            m._cnerf_reducer.network_ready(m)


class _MlpFn(torch.autograd.Function):
    """Fused gamma(x), gamma(d) + MLP (replaces R:37-52 + H:44-45 + H:107-130 and their autograd)."""

    @staticmethod
    def forward(ctx, model, B, S, pts, rays, z, dirs, emb, live, skip, *params):
        spec = model.spec()
        packed, gen = _packed_gen(model)
        if any(ctx.needs_input_grad[3:8]):
            # fail loudly rather than return silently-missing gradients: the reference never differentiates through the
            # sample positions (z is detached at R:397, rays come from the data), and the dgrad kernel stops at layer 0
            raise ops.CnerfError("gradients w.r.t. sample positions / rays / view directions / pre-embedded inputs are "
                                 "not implemented (only w.r.t. the network parameters)")
        train = any(ctx.needs_input_grad[10:])
        if emb is not None:      # NeRF.forward(x) on pre-embedded inputs
            raw, stash = ops.mlp_forward_embedded(spec, packed, emb, want_stash=train)
        elif train and training_precision(model) == "bf16x3":
            # OPT-IN second training arithmetic (never the default): the forward GEMMs on the bf16 matrix cores at three planes
            # per operand; same stash, so the backward below is unchanged.  `packed` (fp32 panels) still feeds the dgrad.
            ctx.packed_bf, ctx.packed_bf_gen = _packed_bf_gen(model, 3)
            raw, stash = ops.mlp_forward_bf_train(spec, ctx.packed_bf, B, S, pts=pts, rays=rays, z=z, dirs=dirs)
        else:
            # `live` (device int32 [1]; RayPoints.live): the batch is padded to its capacity B and only the first live[0] rays are
            # real — the training kernels stop there (run_nerf_view.ss_step_loss: the one-render in-loop consistency step).  Every
            # other route simply processes the padding rays too (they are valid rays whose loss weight is 0: same result, more work).
            use_live = live if (train and pts is None and dirs is None and S % 32 == 0) else None
            raw, stash = ops.mlp_forward(spec, packed, B, S, pts=pts, rays=rays, z=z, dirs=dirs, want_stash=train, live=use_live)
            ctx.live = use_live
            # `skip` (with live): this level's first `skip` rays will get zero seeds — the merged backward leaves them out
            ctx.skip = int(skip) if (use_live is not None and skip) else 0
        if train:
            ctx.spec, ctx.B, ctx.S, ctx.stash, ctx.packed, ctx.packed_gen = spec, B, S, stash, packed, gen
            ctx.params, ctx.model = params, model
            if hasattr(model, "_cnerf_pending"):     # distributed.GradReducer counts this network's backward nodes
                model._cnerf_pending += 1
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        params = ctx.params
        if not _packed_still_valid(ctx.model, ctx.packed_gen):
            raise ops.CnerfError("the network's weights were updated twice between this forward pass and its backward "
                                 "(the kernel-layout copy it used has been re-packed)")
        if getattr(ctx, "packed_bf", None) is not None and not _packed_bf_still_valid(ctx.model, ctx.packed_bf_gen, ctx.packed_bf):
            raise ops.CnerfError("the network's weights were updated twice between this bf16x3 forward pass and its backward "
                                 "(the bf16-plane copy it used has been re-packed)")
        # Parameters owned by FusedAdam carry their .grad as a view into the flat gradient buffer: under loss.backward()
        # the wgrad reduction accumulates straight into it (what AccumulateGrad would do with ~50 add/copy launches per
        # step) and autograd gets no per-tensor gradients back.  Anything else — plain nn.Parameters, and
        # torch.autograd.grad() on FusedAdam-owned ones, where the engine captures gradients instead of accumulating them
        # — takes the tensor route and leaves the flat buffer untouched.
        direct = _direct_ok(ctx.needs_input_grad[10:], params) and _engine_accumulates(params[0])
        pair = getattr(ctx, "pair", None)
        nret = (None,) * (10 + len(params))
        live = getattr(ctx, "live", None)
        if direct and pair is not None and pair.fine() is ctx and pair.parked is None:
            c = pair.coarse()
            # park only when the coarse node is certain to run in this very pass, on the direct route as well
            if (c is not None and c.stash is not None and _direct_ok(c.needs_input_grad[10:], c.params)
                    and _ENGINE_QUERY(c)):
                pair.parked = (ctx.spec, ctx.packed, g_raw.contiguous(), ctx.B, ctx.S, ctx.stash, [p.grad for p in params],
                               ctx.model, getattr(ctx, "packed_bf", None), params, live, getattr(ctx, "skip", 0))
                ctx.stash = ctx.packed = ctx.params = ctx.model = None
                return nret
        parked = None
        if pair is not None and pair.coarse() is ctx and pair.parked is not None:
            parked, pair.parked = pair.parked, None
        if parked is not None and direct and parked[10] is live and (live is None or parked[3] == ctx.B):
            fs, fp, fg, fB, fS, fst, fgr, fmodel, fbf, fparams, _flive, fskip = parked
            ops.mlp_backward_pair(fs, fp, fg, fB, fS, fst, fgr, ctx.spec, ctx.packed, g_raw.contiguous(), ctx.B, ctx.S,
                                  ctx.stash, [p.grad for p in params], accumulate=not _take_dropped(fparams, params), packed_bf0=fbf,
                                  packed_bf1=getattr(ctx, "packed_bf", None), live=live, first0=fskip if live is not None else 0,
                                  first1=getattr(ctx, "skip", 0) if live is not None else 0)
            _report_ready_pair(fmodel, ctx.model)
            ctx.stash = ctx.packed = ctx.params = ctx.model = None
            return nret
        if parked is not None:        # (cannot happen — the fine node checked this node's route — but never drop a gradient)
            fs, fp, fg, fB, fS, fst, fgr, fmodel, fbf, fparams, flive, _fskip = parked
            ops.mlp_backward(fs, fp, fg, fB, fS, fst, grads=fgr, accumulate=not _take_dropped(fparams), packed_bf=fbf, live=flive)
            _report_ready(fmodel, True)
        out = [p.grad for p in params] if direct else None
        grads = ops.mlp_backward(ctx.spec, ctx.packed, g_raw.contiguous(), ctx.B, ctx.S, ctx.stash, grads=out,
                                 accumulate=direct and not _take_dropped(params), packed_bf=getattr(ctx, "packed_bf", None),
                                 live=live)
        _report_ready(ctx.model, direct)
        ctx.stash = ctx.packed = ctx.params = ctx.model = None
        return nret if direct else (None,) * 10 + tuple(grads)


class _CompositeFn(torch.autograd.Function):
    """raw2outputs (R:265-308) and its backward."""

    @staticmethod
    def forward(ctx, raw, z_vals, rays, noise, white_bkgd):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:   # (z is detached in the reference, R:397; rays are data)
            raise ops.CnerfError("raw2outputs: gradients w.r.t. z_vals / rays are not implemented (only w.r.t. raw)")
        rgb, disp, acc, weights, depth = ops.composite_forward(raw, z_vals, rays, noise, white_bkgd)
        ctx.save_for_backward(raw, z_vals, rays)
        ctx.noise, ctx.white = noise, white_bkgd
        ctx.mark_non_differentiable(weights)
        ctx.set_materialize_grads(False)
        return rgb, disp, acc, weights, depth

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_weights, g_depth):
        raw, z_vals, rays = ctx.saved_tensors
        d_raw = ops.composite_backward(raw, z_vals, rays, ctx.noise, ctx.white, g_rgb, g_disp, g_acc, g_depth)
        return d_raw, None, None, None, None


class _RenderLossFn(torch.autograd.Function):
    """raw2outputs of the LAST level (R:265-308) with the photometric loss of the training loop folded in (R:769-775):
    loss = img2mse(rgb_map, target) [+ img2mse(rgb0, target)] as ONE autograd node from (raw, raw_coarse) to the scalar.  Forward:
    the last level's compositing kernel also reduces the squared error and adds the coarse level's term (computed by ITS compositing
    launch, render_rays); backward: the two compositing-backward kernels form their seed (2 / n) (rgb - target) * g in registers.
    Gone from the step: two loss launches, their `+`, the two `d_x * g` of their backward.  The maps come back detached (values for
    logging, R:776): a caller that differentiates anything else of a map uses render() + img2mse()."""

    @staticmethod
    def forward(ctx, raw, raw_c, z, z_c, rays, noise, noise_c, white, target, rgb_c, loss_c):
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3] or ctx.needs_input_grad[4] or ctx.needs_input_grad[8]:
            raise ops.CnerfError("render_loss: gradients w.r.t. z_vals / rays / target are not implemented (only w.r.t. raw)")
        rgb, disp, acc, weights, depth, loss = ops.composite_forward_mse(raw, z, rays, noise, white, target, loss_add=loss_c)
        ctx.save_for_backward(raw, raw_c, z, z_c, rays, target, rgb, rgb_c)
        ctx.noise, ctx.noise_c, ctx.white = noise, noise_c, white
        ctx.mark_non_differentiable(rgb, disp, acc, weights, depth)
        ctx.set_materialize_grads(False)
        return loss.view(()), rgb, disp, acc, weights, depth

    @staticmethod
    def backward(ctx, g_loss, *_maps):
        raw, raw_c, z, z_c, rays, target, rgb, rgb_c = ctx.saved_tensors
        d_raw = d_raw_c = None
        if g_loss is not None:
            if ctx.needs_input_grad[0]:
                d_raw = ops.composite_backward_mse(raw, z, rays, ctx.noise, ctx.white, rgb, target, g_loss)
            if raw_c is not None and ctx.needs_input_grad[1]:
                d_raw_c = ops.composite_backward_mse(raw_c, z_c, rays, ctx.noise_c, ctx.white, rgb_c, target, g_loss)
        return (d_raw, d_raw_c) + (None,) * 9


class _RenderClossFn(torch.autograd.Function):
    """raw2outputs of the LAST level with ConsistentNeRF's whole step loss folded in (V:1645-1865; ops.ClossSpec): ONE autograd node
    from (raw, raw_coarse) to the scalar.  Forward: the last level's compositing launch leaves its masked-loss partial sums (the
    coarse level's launch left its own, render_rays), cnerf_closs_finish sums both, evaluates the patch term of both levels and
    assembles the loss; backward: the two compositing-backward launches form their rgb / depth / patch seeds in registers.  Gone from
    the step: 2 masked-loss + 2 patch-term launches and the ~25 ATen kernels between them (adds, muls, index copies, fills)."""

    @staticmethod
    def forward(ctx, raw, raw_c, z, z_c, rays, noise, noise_c, white, L, rgb_c, depth_c, ws_c):
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            raise ops.CnerfError("render_loss: gradients w.r.t. z_vals / rays are not implemented (only w.r.t. raw)")
        rgb, disp, acc, weights, depth, ws = ops.composite_forward_closs(raw, z, rays, noise, white, L)
        want = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        terms, stats, patch_d = ops.closs_finish(L, z.shape[0], ws, ws_c, depth, depth_c, want_grad=want)
        ctx.save_for_backward(raw, raw_c, z, z_c, rays, rgb, rgb_c, depth, depth_c, stats, patch_d)
        ctx.noise, ctx.noise_c, ctx.white, ctx.L = noise, noise_c, white, L
        ctx.mark_non_differentiable(terms, rgb, disp, acc, weights, depth)
        ctx.set_materialize_grads(False)
        return terms[0], terms, rgb, disp, acc, weights, depth

    @staticmethod
    def backward(ctx, g_loss, *_rest):
        raw, raw_c, z, z_c, rays, rgb, rgb_c, depth, depth_c, stats, patch_d = ctx.saved_tensors
        d_raw = d_raw_c = None
        if g_loss is not None:
            L = ctx.L
            w = 8 if L.seg_row else 4       # (two segments: per level [2][4] seed weights, cnerf_closs_finish_ss2)
            if ctx.needs_input_grad[0]:
                d_raw = ops.composite_backward_closs(raw, z, rays, ctx.noise, ctx.white, L, rgb, depth, stats[0:w], g_loss,
                                                     None if patch_d is None else patch_d[0])
            if raw_c is not None and ctx.needs_input_grad[1]:
                d_raw_c = ops.composite_backward_closs(raw_c, z_c, rays, ctx.noise_c, ctx.white, L, rgb_c, depth_c, stats[w:2 * w],
                                                       g_loss, None if patch_d is None else patch_d[1])
        return (d_raw, d_raw_c) + (None,) * 10


_ONES = {}


def backward(loss):
    """`loss.backward()` without the fill launch of its implicit ones_like seed: the seed is a cached device constant."""
    key = (str(loss.device), loss.dtype)
    one = _ONES.get(key)
    if one is None:
        one = torch.ones((), device=loss.device, dtype=loss.dtype)
        if not torch.cuda.is_current_stream_capturing():     # (a tensor allocated inside a recording belongs to the graph's pool)
            _ONES[key] = one
    torch.autograd.backward(loss, grad_tensors=one)


# ----------------------------------------------------------------------------- reference surface
def batchify(fn, chunk):
    """R:27-34.  Kept for API parity; the fused kernel tiles internally so chunking is a no-op."""
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """R:37-52.  inputs: [N_rays, N_samples, 3] tensor (or the lazy RayPoints render_rays passes);
    viewdirs: [N_rays, 3] or None.  embed_fn / embeddirs_fn only have to agree with the widths `fn` was
    built with (the encodings are generated inside the kernel); netchunk is advisory (results are
    chunk-invariant, R:79-80)."""
    if not isinstance(fn, NeRF):
        raise TypeError("run_network needs a consistentnerf_amd NeRF module")
    spec = fn.spec()
    if getattr(embed_fn, "out_dim", 3) != fn.input_ch:
        raise ValueError("embed_fn width does not match the network input_ch")
    if spec.use_viewdirs:
        if viewdirs is None:
            raise ValueError("network was built with use_viewdirs=True but viewdirs is None")
        if embeddirs_fn is not None and getattr(embeddirs_fn, "out_dim", 3) != fn.input_ch_views:
            raise ValueError("embeddirs_fn width does not match the network input_ch_views")
    B, S = inputs.shape[0], inputs.shape[1]
    dirs = viewdirs if (viewdirs is not None and spec.use_viewdirs) else None
    if dirs is not None and isinstance(inputs, RayPoints) and _is_viewdir_columns(dirs, inputs.rays):
        dirs = None      # the kernel reads the view directions from the last three columns of the ray rows (R:350): no copy
    elif dirs is not None:
        dirs = dirs.contiguous()
    params = fn.kernel_tensors()
    prec = getattr(fn, "inference_precision", "fp32")
    if prec != "fp32" and not (torch.is_grad_enabled() and any(p.requires_grad for p in params)):
        # OPT-IN reduced-precision inference (NeRF.inference_precision = "bf16" | "bf16x2" | "bf16x3"): bf16 matrix cores,
        # fp32 accumulation; never taken when a gradient could be asked for
        if prec not in ops.PRECISION_PLANES:
            raise ValueError(f"inference_precision must be fp32 or one of {sorted(ops.PRECISION_PLANES)}")
        planes = ops.PRECISION_PLANES[prec]
        pk = _packed_bf(fn, planes)
        if isinstance(inputs, RayPoints):
            return ops.mlp_forward_bf(spec, pk, planes, B, S, rays=inputs.rays, z=inputs.z_vals, dirs=dirs)
        return ops.mlp_forward_bf(spec, pk, planes, B, S, pts=inputs.reshape(-1, 3).contiguous(), dirs=dirs)
    if not torch.is_grad_enabled():
        # under torch.no_grad() a Function still sees needs_input_grad = True for the parameters: detach them, or the
        # inference pass would run the training kernel and write a 10 KB-per-point stash nobody reads
        params = [p.detach() for p in params]
    if isinstance(inputs, RayPoints):
        return _MlpFn.apply(fn, B, S, None, inputs.rays, inputs.z_vals, dirs, None, getattr(inputs, "live", None),
                            getattr(inputs, "skip", 0), *params)
    pts = inputs.reshape(-1, 3).contiguous()
    return _MlpFn.apply(fn, B, S, pts, None, None, dirs, None, None, 0, *params)


def _is_viewdir_columns(viewdirs, rays):
    """True when `viewdirs` IS rays[:, -3:] (same storage): what render_rays passes (R:350)."""
    return (rays.dim() == 2 and rays.shape[1] >= 11 and viewdirs.dim() == 2 and viewdirs.shape == (rays.shape[0], 3)
            and viewdirs.dtype == rays.dtype and rays.is_contiguous() and viewdirs.stride() == (rays.shape[1], 1)
            and viewdirs.data_ptr() == rays.data_ptr() + 4 * (rays.shape[1] - 3))


def _jitter_and_u(rows, Nc, Nf, dev, global_rows):
    """The stratified-jitter stream t_rand [rows, Nc] (R:376) and the resampling stream u [rows, Nf] (H:227) of one render_rays
    call from ONE generator call: a flat draw cut into two contiguous blocks (one launch instead of two; with `global_rows` the
    draw is for the whole batch and this call's rows are sliced, _rows_of_global)."""
    off, total = (0, rows) if global_rows is None else (int(global_rows[0]), int(global_rows[1]))
    r = torch.rand(total * (Nc + Nf), device=dev)
    t_rand = r[:total * Nc].view(total, Nc)[off:off + rows]
    u = r[total * Nc:].view(total, Nf)[off:off + rows] if Nf > 0 else None
    return t_rand, u


def _in_kernel_rng():
    """The jitter / resampling streams are generated inside coarse_z_k / resample_k (ops.rng_draw) — unless switched off, or a
    hipGraph is being recorded by something other than graph.GraphedStep (torch.rand is the graph-safe generator then)."""
    return ops.IN_KERNEL_RNG and (ops.RngCapture.active is not None or not torch.cuda.is_current_stream_capturing())


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False):
    """R:265-308 -> (rgb_map, disp_map, acc_map, weights, depth_map)."""
    B = z_vals.shape[0]
    rays = torch.cat([torch.zeros_like(rays_d), rays_d], -1).contiguous()
    noise = _density_noise(raw.shape[:2], raw_noise_std, pytest, raw.device)
    return _CompositeFn.apply(raw.contiguous(), z_vals.contiguous(), rays, noise, bool(white_bkgd))


def _rows_of_global(draw, rows, cols, global_rows):
    """`draw(n, cols)` for this call's `rows` rays — or, when the rays are rows [offset, offset + rows) of a GLOBAL batch of
    `total` rays sharded over ranks (`global_rows = (offset, total)`), the matching rows of the draw for the whole batch: every
    rank consumes the identical generator state for the identical global stream, so an N-rank step sees exactly the random
    numbers the 1-rank step on the whole batch sees ("t_rand / u generated from the global batch and sliced")."""
    if global_rows is None:
        return draw(rows, cols)
    off, total = global_rows
    return draw(int(total), cols)[int(off):int(off) + rows].contiguous()


def _pytest_rows(rows, cols, device, global_rows):
    """The reference's deterministic stream (np.random.seed(0); np.random.rand) for this call's rows — of the WHOLE batch when the
    rays are a shard of it, so that the shards see the rows the unsharded call sees."""
    if global_rows is None:
        return pytest_uniform((rows, cols), device)
    off, total = global_rows
    return pytest_uniform((int(total), cols), device)[int(off):int(off) + rows].contiguous()


def _density_noise(shape, raw_noise_std, pytest, device, global_rows=None):
    if not raw_noise_std > 0.:
        return None
    if pytest:   # R:290-294: uniform in pytest mode
        return _pytest_rows(shape[0], shape[1], device, global_rows) * raw_noise_std
    return _rows_of_global(lambda n, c: torch.randn(n, c, device=device), shape[0], shape[1], global_rows) * raw_noise_std


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """R:55-67."""
    all_ret = {}
    gr = kwargs.pop("_global_rows", None)
    if kwargs.get("_live") is not None and rays_flat.shape[0] > chunk:
        raise ops.CnerfError(f"_live needs the padded batch ({rays_flat.shape[0]} rows) to fit one chunk ({chunk})")
    if gr is not None and rays_flat.shape[0] > chunk:
        # every chunk would draw the whole batch's stream afresh, while the unsharded call draws one stream PER CHUNK: the
        # "N-rank step sees the 1-rank step's random numbers row for row" guarantee holds for shards that fit one chunk
        # (the training case: N_rand <= chunk)
        raise ops.CnerfError(f"_global_rows needs the shard ({rays_flat.shape[0]} rays) to fit one chunk ({chunk})")
    for i in range(0, rays_flat.shape[0], chunk):
        if gr is not None:     # this chunk's rows of the global batch
            kwargs["_global_rows"] = (gr[0] + i, gr[1])
        ret = render_rays(rays_flat[i:i + chunk], **kwargs)
        for k in ret:
            all_ret.setdefault(k, []).append(ret[k])
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in all_ret.items()}


def render_loss(H, W, K, target_s, chunk=1024 * 32, rays=None, **kwargs):
    """R:764-775 as one call — `rgb, disp, acc, extras = render(H, W, K, chunk=, rays=batch_rays, retraw=True, **render_kwargs_train);
    img_loss = img2mse(rgb, target_s); loss = img_loss (+ img2mse(extras['rgb0'], target_s))` — with the loss folded into the
    compositing launches (_RenderLossFn).  -> (loss, rgb, disp, acc, extras); the same values and, after loss.backward() (or
    run_nerf.backward(loss): no seed fill), the same parameter gradients bit for bit as the lines above, from 6 launches fewer.
    The maps are detached.  Batches larger than `chunk`, or an empty one, take the lines above literally."""
    n = rays[0].reshape(-1, 3).shape[0] if rays is not None else 0
    tgt = target_s.reshape(-1, 3) if torch.is_tensor(target_s) else None
    if (rays is None or n == 0 or n > min(chunk, ops.composite_mse_max_rays()) or tgt is None or not tgt.is_cuda or tgt.dtype != torch.float32 or tgt.shape[0] != n
            or kwargs.get('c2w') is not None):
        rgb, disp, acc, extras = render(H, W, K, chunk=chunk, rays=rays, **kwargs)
        loss = img2mse(rgb, target_s)
        if 'rgb0' in extras:
            loss = loss + img2mse(extras['rgb0'], target_s)
        return loss, rgb, disp, acc, extras
    rgb, disp, acc, extras = render(H, W, K, chunk=chunk, rays=rays, _target=tgt.contiguous(), **kwargs)
    return extras.pop('loss'), rgb, disp, acc, extras


def _ray_batch(H, W, K, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, device):
    """R:97-125 -> (rays [B, 8|11], output leading shape)."""
    coef = ndc_coefficients(H, W, K[0][0]) if ndc else (0., 0.)
    # per-ray tensor bounds are written after the pack (R:111 `near * ones` takes any mix of scalars and tensors)
    near_s = 0. if torch.is_tensor(near) else near
    far_s = 1. if torch.is_tensor(far) else far
    if c2w is not None:
        sh = (H, W)
        if c2w_staticcam is not None and use_viewdirs:
            # viewdirs from c2w, geometry from the static camera (R:104-108)
            vd = ops.gen_rays(H, W, K, c2w, near_s, far_s, True, False, device)[:, 8:11]
            batch = ops.gen_rays(H, W, K, c2w_staticcam, near_s, far_s, True, ndc, device, coef)
            batch[:, 8:11] = vd
        else:
            batch = ops.gen_rays(H, W, K, c2w, near_s, far_s, use_viewdirs, ndc, device, coef)
    else:
        pk = getattr(rays, "_cnerf_packed", None)
        if pk is not None and pk.matches(H, W, K, near, far, use_viewdirs, ndc, device, src=rays if torch.is_tensor(rays) else None):
            # raybank's one-launch sampler already wrote the [B, 8|11] rows this call would assemble (same arithmetic: raygen.hpp)
            return pk.rows, (pk.rows.shape[0],)
        rays_o, rays_d = rays
        sh = tuple(rays_d.shape[:-1])
        batch = ops.pack_rays(rays_o.to(device), rays_d.to(device), near_s, far_s, use_viewdirs, ndc, coef)
    if torch.is_tensor(near):
        batch[:, 6] = near.reshape(-1).to(batch)
    if torch.is_tensor(far):
        batch[:, 7] = far.reshape(-1).to(batch)
    return batch, sh


def _default_device():
    if not torch.cuda.is_available():
        raise ops.CnerfError("consistentnerf_amd needs an MI355X (no CPU execution path)")
    return torch.device("cuda", torch.cuda.current_device())


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """R:70-137 -> [rgb_map, disp_map, acc_map, extras]."""
    return _render(H, W, K, chunk, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, False, kwargs)


def _camera_path_ok(rays, c2w, near, far, use_viewdirs, c2w_staticcam, kwargs):
    """render(c2w=...) of a whole image for inference can generate its rays inside the kernels (cnerf_render_fwd_cam):
    that needs the stock network_query_fn of create_nerf (the kernels ARE that query), scalar near / far, no static
    camera, and no autograd graph to build."""
    net, fine = kwargs.get("network_fn"), kwargs.get("network_fine")
    if rays is not None or c2w is None or c2w_staticcam is not None or torch.is_tensor(near) or torch.is_tensor(far):
        return False
    if not getattr(kwargs.get("network_query_fn"), "_cnerf_stock", False) or not isinstance(net, NeRF):
        return False
    if fine is not None and not isinstance(fine, NeRF):
        return False
    if DEBUG:            # the ray-tensor path carries render_rays' NaN / Inf report (R:417-419)
        return False
    nets = [net] + ([fine] if fine is not None else [])
    if any(bool(n.use_viewdirs) != bool(use_viewdirs) for n in nets):
        return False
    if any(getattr(n, "inference_precision", "fp32") != "fp32" for n in nets):   # the camera path is the exact-fp32 one
        return False
    return not (torch.is_grad_enabled() and any(p.requires_grad for n in nets for p in n.parameters()))


def camera_path_ok(c2w, render_kwargs):
    """True when render(c2w=..., **render_kwargs) would take the in-kernel camera path (distributed.render_image_sharded asks
    before calling render_pixels)."""
    kw = dict(render_kwargs)
    near, far = kw.pop("near", 0.), kw.pop("far", 1.)
    return _camera_path_ok(None, c2w, near, far, kw.pop("use_viewdirs", False), kw.pop("c2w_staticcam", None), kw)


def render_pixels(H, W, K, chunk, c2w, first, count, ndc=True, near=0., far=1., use_viewdirs=False, with_depth=False, **kwargs):
    """The pixels [first, first + count) (row-major) of the image render(H, W, K, c2w=c2w, ...) would produce, as flat
    per-ray maps {rgb_map [count, 3], disp_map [count], acc_map [count], ...}: the row block of one rank of a sharded frame
    (distributed.render_image_sharded).  Inference only, stock network_query_fn only (camera_path_ok); the rays are generated
    inside the kernels, and every pixel equals the one of the full-frame render bit for bit (rays are independent; chunk
    boundaries do not matter, R:79-80) as long as no per-call random stream is involved (perturb = 0, raw_noise_std = 0)."""
    if isinstance(c2w, torch.Tensor):
        c2w = c2w[:3, :4]
    if not _camera_path_ok(None, c2w, near, far, use_viewdirs, None, kwargs):
        raise ops.CnerfError("render_pixels needs the in-kernel camera path (stock network_query_fn, scalar near / far, "
                             "no autograd graph)")
    return _render_camera(H, W, K, chunk, c2w, ndc, near, far, use_viewdirs, with_depth, kwargs, int(first), int(count))


def _render_camera(H, W, K, chunk, c2w, ndc, near, far, use_viewdirs, with_depth, kwargs, first0=0, count=None):
    """batchify_rays (R:55-67) + render_rays (R:311-421) over the image of one camera (or its pixels [first0, first0 +
    count)), rays generated in-kernel: one C call per chunk, the same random streams in the same order as the ray-tensor
    path (bit-identical results)."""
    net, fine = kwargs["network_fn"], kwargs.get("network_fine")
    Nc, Nf = kwargs["N_samples"], kwargs.get("N_importance", 0)
    perturb, pytest = kwargs.get("perturb", 0.), kwargs.get("pytest", False)
    std, dev = kwargs.get("raw_noise_std", 0.), next(net.parameters()).device
    coef = ndc_coefficients(H, W, K[0][0]) if ndc else (0., 0.)
    two_nets = fine is not None and Nf > 0
    if isinstance(c2w, torch.Tensor):     # ONE device-to-host copy of the 12 pose floats per frame, not one per chunk
        c2w = c2w.detach().cpu().numpy()
    all_ret = {}
    end = H * W if count is None else min(H * W, first0 + count)
    for first in range(first0, end, chunk):
        B = min(chunk, end - first)
        t_rand = u = None
        if perturb > 0.:
            if pytest:
                t_rand = pytest_uniform((B, Nc), dev)
            elif _in_kernel_rng():
                # the streams render_rays' kernels generate for the same generator state, materialised (this C call takes tensors)
                rng = ops.rng_draw(dev)
                t_rand = ops.uniform_rng(rng, B, Nc, dev, 0)
                u = ops.uniform_rng(rng, B, Nf, dev, 1) if Nf > 0 else None
            else:
                t_rand, u = _jitter_and_u(B, Nc, Nf, dev, None)     # (the same draw as render_rays: bit-identical paths)
        noise0 = _density_noise((B, Nc), std, pytest, dev)
        if u is None:
            u = sample_u(B, Nf, perturb == 0., pytest, dev) if Nf > 0 else None
        noise1 = _density_noise((B, Nc + Nf), std, pytest, dev) if Nf > 0 else None
        if two_nets:
            _prepack_pair(net, fine)
        o = ops.render_forward_cam(net.spec(), _packed(net), fine.spec() if two_nets else None,
                                   _packed(fine) if two_nets else None, H, W, K, c2w, near, far, use_viewdirs, ndc, coef, first,
                                   B, Nc, Nf, t_rand, u, noise0, noise1, bool(kwargs.get("lindisp", False)),
                                   bool(kwargs.get("white_bkgd", False)), bool(kwargs.get("retraw", False)))
        if not with_depth:
            o.pop("depth_map", None)
            o.pop("depth0", None)
        for k, v in o.items():
            all_ret.setdefault(k, []).append(v)
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in all_ret.items()}


def _render(H, W, K, chunk, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, with_depth, kwargs):
    net = kwargs.get("network_fn")
    device = next(net.parameters()).device if net is not None else _default_device()
    if _camera_path_ok(rays, c2w, near, far, use_viewdirs, c2w_staticcam, kwargs):
        kw = {k: v for k, v in kwargs.items()}
        all_ret = _render_camera(H, W, K, chunk, c2w, ndc, near, far, use_viewdirs, with_depth, kw)
        for k in all_ret:
            all_ret[k] = torch.reshape(all_ret[k], [H, W] + list(all_ret[k].shape[1:]))
        k_extract = ['rgb_map', 'disp_map', 'acc_map'] + (['depth_map'] if with_depth else [])
        return [all_ret[k] for k in k_extract] + [{k: all_ret[k] for k in all_ret if k not in k_extract}]
    batch, sh = _ray_batch(H, W, K, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, device)
    all_ret = batchify_rays(batch, chunk, _with_depth=with_depth, **kwargs)
    for k in all_ret:
        if k not in ('loss', 'loss_terms'):          # (the scalars of render_loss pass through)
            all_ret[k] = torch.reshape(all_ret[k], list(sh) + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'acc_map'] + (['depth_map'] if with_depth else [])
    ret_list = [all_ret[k] for k in k_extract]
    ret_dict = {k: all_ret[k] for k in all_ret if k not in k_extract}
    return ret_list + [ret_dict]


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0):
    """R:140-178 -> (rgbs, disps) numpy."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    rgbs, disps = [], []
    t = time.time()
    for i, c2w in enumerate(render_poses):
        print(i, time.time() - t)
        t = time.time()
        with torch.no_grad():
            rgb, disp, acc, _ = render(H, W, K, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
        rgbs.append(rgb.cpu().numpy())
        disps.append(disp.cpu().numpy())
        if i == 0:
            print(rgb.shape, disp.shape)
        if savedir is not None:
            _save_png(os.path.join(savedir, '{:03d}.png'.format(i)), to8b(rgbs[-1]))
    return np.stack(rgbs, 0), np.stack(disps, 0)


def _save_png(path, img8):
    try:
        import imageio
        imageio.imwrite(path, img8)
    except ImportError:   # image IO is outside the hot path; keep the render usable without imageio
        np.save(path + ".npy", img8)


def save_checkpoint(basedir, expname, global_step, render_kwargs_train, optimizer):
    """The checkpoint write of train() (R:836-845): `{basedir}/{expname}/{global_step:06d}.tar` with the reference's
    keys, loadable by the reference and by create_nerf() here."""
    path = os.path.join(basedir, expname, '{:06d}.tar'.format(global_step))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    fine = render_kwargs_train.get('network_fine')
    ckpt = {'global_step': global_step,
            'network_fn_state_dict': render_kwargs_train['network_fn'].state_dict(),
            'optimizer_state_dict': optimizer.state_dict()}
    if fine is not None:
        ckpt['network_fine_state_dict'] = fine.state_dict()
    torch.save(ckpt, path)
    print('Saved checkpoints at', path)
    return path


def create_nerf(args):
    """R:181-262 -> (render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer)."""
    return _create_nerf(args, NeRF, False)


def _create_nerf(args, model_cls, view_variant):
    device = _default_device()
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    input_ch_views, embeddirs_fn = 0, None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    extra = dict(coarse=True, stable_init=getattr(args, "stable_init", False)) if view_variant else {}
    model = model_cls(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=skips,
                      input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs, **extra).to(device)
    grad_vars = list(model.parameters())
    model_fine = None
    if args.N_importance > 0:
        extra_f = dict(stable_init=getattr(args, "stable_init", False)) if view_variant else {}
        model_fine = model_cls(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch,
                               skips=skips, input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs,
                               **extra_f).to(device)
        grad_vars += list(model_fine.parameters())
    if view_variant:   # V:321: the coarse net starts as a copy of the fine net
        model.load_state_dict(model_fine.state_dict())

    def network_query_fn(inputs, viewdirs, network_fn):
        return run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                           netchunk=args.netchunk)

    network_query_fn._cnerf_stock = True     # render(c2w=...) may replace it by the single-call camera path
    optimizer = FusedAdam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start = 0
    basedir, expname = args.basedir, args.expname
    if args.ft_path is not None and args.ft_path != 'None':
        ckpts = [args.ft_path]
    else:
        d = os.path.join(basedir, expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if 'tar' in f] if os.path.isdir(d) else []
    print('Found ckpts', ckpts)
    if len(ckpts) > 0 and not args.no_reload:
        ckpt_path = ckpts[-1]
        print('Reloading from', ckpt_path)
        ckpt = torch.load(ckpt_path, map_location=device, weights_only=False)
        start = ckpt['global_step']
        if not view_variant:      # V:351 skips the optimizer reload
            optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        sd_c, sd_f = ckpt['network_fn_state_dict'], ckpt.get('network_fine_state_dict')
        if view_variant:          # V:353-358 resets the three scalars
            for sd in (sd_c, sd_f):
                if sd is not None:
                    for k in ('temp_rgb', 'temp_depth', 'depth_scale'):
                        sd[k] = torch.full((1,), 0.1, device=device)
        model.load_state_dict(sd_c)
        if model_fine is not None and sd_f is not None:
            model_fine.load_state_dict(sd_f)
    render_kwargs_train = {
        'network_query_fn': network_query_fn, 'perturb': args.perturb, 'N_importance': args.N_importance,
        'network_fine': model_fine, 'N_samples': args.N_samples, 'network_fn': model,
        'use_viewdirs': args.use_viewdirs, 'white_bkgd': args.white_bkgd, 'raw_noise_std': args.raw_noise_std,
    }
    if args.dataset_type != 'llff' or args.no_ndc:
        print('Not ndc!')
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False,
                _with_depth=False, _debug=False, _global_rows=None, _target=None, _live=None):
    """R:311-421 (V:441-551 when _with_depth).  Returns the same dict (+ the sample depths when _debug).
    `_global_rows = (offset, total)`: `ray_batch` is rows [offset, offset + N_rays) of a global batch of `total` rays sharded over
    ranks — the jitter / resampling / noise streams are drawn for the whole batch and sliced (_rows_of_global).
    `_target` [N_rays, 3] (render_loss): the compositing launches also produce ret['loss'] = img2mse(rgb_map, _target)
    (+ img2mse(rgb0, _target) with two levels) through _RenderLossFn; the maps are then detached values.
    `_live` (device int32 [1]): `ray_batch` is padded to a fixed capacity, only its first _live[0] rows are real rays (the rest are
    valid dummy rays whose loss weight is 0): the MLP training kernels stop at the count, which never visits the host."""
    rays = ray_batch if ray_batch.is_contiguous() else ray_batch.contiguous()
    N_rays, dev = rays.shape[0], rays.device
    if N_rays == 0:   # nothing to launch (the reference's batchify_rays raises on an empty batch; here: empty maps)
        S = N_samples + N_importance
        last = network_fn if (network_fine is None or N_importance <= 0) else network_fine
        e = lambda *sh: torch.empty(*sh, device=dev)  # noqa: E731
        ret = {'rgb_map': e(0, 3), 'disp_map': e(0), 'acc_map': e(0)}
        if _with_depth:
            ret['depth_map'] = e(0)
        if retraw:
            ret['raw'] = e(0, S, last.spec().raw_ch)
        if N_importance > 0:
            ret.update({'rgb0': e(0, 3), 'disp0': e(0), 'acc0': e(0), 'z_std': e(0)})
            if _with_depth:
                ret['depth0'] = e(0)
        return ret
    viewdirs = rays[:, -3:] if rays.shape[-1] > 8 else None
    t_rand = u_drawn = rng = None
    if perturb > 0.:
        if pytest:
            t_rand = _pytest_rows(N_rays, N_samples, dev, _global_rows)
        elif _in_kernel_rng():
            # no generator launch: both streams are generated where they are consumed, indexed by the GLOBAL row (a shard of a
            # batch sees the rows the unsharded call sees without drawing the whole batch's stream)
            rng = ops.rng_draw(dev, 0 if _global_rows is None else int(_global_rows[0]))
        else:
            t_rand, u_drawn = _jitter_and_u(N_rays, N_samples, N_importance, dev, _global_rows)
    z_vals = ops.coarse_z(rays, N_samples, t_rand, lindisp, rng=rng)
    if N_importance > 0:
        _prepack_pair(network_fn, network_fine)
    skip_c = 0
    if (_live is not None and isinstance(_target, ops.ClossSpec) and _target.seg_row and _target.ss_coins is not None and N_importance > 0
            and not (_target.ss_coins[2] or (_target.prior is not None and _target.ss_coins[3]))):
        # VT:959 / VT:966 with both coarse coins 0: no term of the PRIMARY segment depends on the coarse network (its colour term falls
        # back to the fine rgb, its depth term is absent) — the coarse level's backward covers the second segment only
        skip_c = int(_target.seg_row)
    raw = network_query_fn(RayPoints(rays, z_vals, _live, skip_c), viewdirs, network_fn)
    noise = _density_noise((N_rays, N_samples), raw_noise_std, pytest, dev, _global_rows)
    loss = loss_c = terms = None
    if _target is None:
        rgb_map, disp_map, acc_map, weights, depth_map = _CompositeFn.apply(raw, z_vals, rays, noise, bool(white_bkgd))
    elif isinstance(_target, ops.ClossSpec):
        _target = _target.checked(N_rays)
        if N_importance > 0:  # coarse level of two: its partial sums ride in its compositing launch; the autograd node comes below
            rgb_map, disp_map, acc_map, weights, depth_map, ws_c = ops.composite_forward_closs(raw, z_vals, rays, noise,
                                                                                               bool(white_bkgd), _target)
        else:
            loss, terms, rgb_map, disp_map, acc_map, weights, depth_map = _RenderClossFn.apply(
                raw, None, z_vals, None, rays, noise, None, bool(white_bkgd), _target, None, None, None)
    elif N_importance > 0:    # coarse level of two: its loss term rides in its compositing launch; the autograd node comes below
        rgb_map, disp_map, acc_map, weights, depth_map, loss_c = ops.composite_forward_mse(raw, z_vals, rays, noise, bool(white_bkgd),
                                                                                          _target)
    else:
        loss, rgb_map, disp_map, acc_map, weights, depth_map = _RenderLossFn.apply(raw, None, z_vals, None, rays, noise, None,
                                                                                   bool(white_bkgd), _target, None, None)
    z_coarse, noise_coarse = z_vals, noise
    if N_importance > 0:
        rgb_map_0, disp_map_0, acc_map_0, depth_map_0 = rgb_map, disp_map, acc_map, depth_map
        if rng is not None:
            u = None                                     # generated inside resample_k
        elif u_drawn is not None:
            u = u_drawn                                  # drawn together with the jitter above (one generator call)
        elif _global_rows is not None and pytest and perturb != 0.:
            u = _pytest_rows(N_rays, N_importance, dev, _global_rows)
        elif _global_rows is not None and perturb != 0.:    # (perturb < 0: random u, no jitter was drawn — H:221 det = perturb == 0)
            u = _rows_of_global(lambda n, c: torch.rand(n, c, device=dev), N_rays, N_importance, _global_rows)
        else:
            u = sample_u(N_rays, N_importance, perturb == 0., pytest, dev)
        z_vals, z_std = ops.resample(z_vals, weights, u, rng=rng, Nf=N_importance)   # R:395-399 + R:415, no gradient (R:397)
        run_fn = network_fn if network_fine is None else network_fine
        raw_coarse = raw
        raw = network_query_fn(RayPoints(rays, z_vals, _live), viewdirs, run_fn)
        _link_levels(raw_coarse, raw)
        noise = _density_noise((N_rays, N_samples + N_importance), raw_noise_std, pytest, dev, _global_rows)
        if _target is None:
            rgb_map, disp_map, acc_map, weights, depth_map = _CompositeFn.apply(raw, z_vals, rays, noise, bool(white_bkgd))
        elif isinstance(_target, ops.ClossSpec):
            rc = raw_coarse
            if (_target.ss_coins is not None and not _target.seg_row
                    and not (_target.ss_coins[2] or (_target.prior is not None and _target.ss_coins[3]))):
                # VT:959 / VT:966 with both coarse coins 0: no term of this render depends on the coarse network (its colour term
                # falls back to the FINE rgb, its depth term is absent) — as in the reference's graph, the coarse level then has no
                # backward at all (left attached, its dgrad + wgrad would run on zero seeds: 16 % of the step for nothing).  (Not in the
                # one-render form of the step, seg_row > 0: the second render's coarse terms always reach the coarse network.)
                rc = raw_coarse.detach()
            loss, terms, rgb_map, disp_map, acc_map, weights, depth_map = _RenderClossFn.apply(
                raw, rc, z_vals, z_coarse, rays, noise, noise_coarse, bool(white_bkgd), _target, rgb_map_0, depth_map_0, ws_c)
        else:
            loss, rgb_map, disp_map, acc_map, weights, depth_map = _RenderLossFn.apply(
                raw, raw_coarse, z_vals, z_coarse, rays, noise, noise_coarse, bool(white_bkgd), _target, rgb_map_0, loss_c)
    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map}
    if loss is not None:
        ret['loss'] = loss
    if terms is not None:
        ret['loss_terms'] = terms
    if _with_depth:
        ret['depth_map'] = depth_map
    if retraw:
        ret['raw'] = raw
    if N_importance > 0:
        ret['rgb0'], ret['disp0'], ret['acc0'] = rgb_map_0, disp_map_0, acc_map_0
        if _with_depth:
            ret['depth0'] = depth_map_0
        ret['z_std'] = z_std
    if _debug:
        ret['_z_coarse'], ret['_z_vals'], ret['_weights'] = z_coarse, z_vals, weights
        if N_importance > 0:
            ret['_raw_coarse'] = raw_coarse
    if DEBUG:
        for k in ret:
            if torch.isnan(ret[k]).any() or torch.isinf(ret[k]).any():
                print(f"! [Numerical Error] {k} contains nan or inf.")
    return ret
