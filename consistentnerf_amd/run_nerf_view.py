"""Drop-in for the hot-path surface of the reference's run_nerf_view.py (V) — the ConsistentNeRF driver:

  render V:183-249 (4 maps + extras) | render_path V:252-294 (rgbs, disps, accs) | create_nerf V:297-389 |
  render_rays V:441-551 | raw2outputs V:392-438 | get_rays_ref V:553 | get_ref_rays V:576 |
  get_test_label V:630 | the hard-mask precompute of train() V:994-1046 (`compute_hard_masks`) |
  the masked RGB / depth losses V:1645-1648, V:1737, V:1786-1788, V:1865 (`hardmask_losses`) |
  the in-loop consistency block of run_nerf_view_test.py: VT:905-938 (`ss_consistency`) and its consumers VT:941-969
  (`ss_primary_losses`)

The older in-loop variant of the warp (run_nerf_view_test.py VT:451-501: no axis flip, masked points) is
available through `get_ref_rays(..., variant="VT")`.
"""
import os
import time

import numpy as np
import torch

from . import ops
from . import run_nerf as _R
from .run_nerf import (batchify, batchify_rays, raw2outputs, run_network)  # noqa: F401
from .run_nerf_helpers import (NeRF, get_embedder, get_rays, get_rays_np, img2mse, mse2psnr, ndc_coefficients,  # noqa: F401
                               ndc_rays, sample_pdf, to8b)


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """V:183-249 -> [rgb_map, disp_map, acc_map, depth_map, extras]."""
    return _R._render(H, W, K, chunk, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam, True, kwargs)


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, **kw):
    """V:441-551."""
    kw.pop("_with_depth", None)
    return _R.render_rays(ray_batch, network_fn, network_query_fn, N_samples, _with_depth=True, **kw)


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0):
    """V:252-294 -> (rgbs, disps, accs) numpy."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = H // render_factor, W // render_factor, focal / render_factor
    rgbs, disps, accs = [], [], []
    t = time.time()
    for i, c2w in enumerate(render_poses):
        print(i, time.time() - t)
        t = time.time()
        with torch.no_grad():
            rgb, disp, acc, _, _ = render(H, W, K, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
        rgbs.append(rgb.cpu().numpy())
        disps.append(disp.cpu().numpy())
        accs.append(acc.cpu().numpy())
        if i == 0:
            print(rgb.shape, disp.shape)
        if savedir is not None:
            _R._save_png(os.path.join(savedir, 'color_{:03d}.png'.format(i)), to8b(rgbs[-1]))
    return np.stack(rgbs, 0), np.stack(disps, 0), np.stack(accs, 0)


def create_nerf(args):
    """V:297-389 (coarse initialised from fine, stable_init, scalars reset on reload)."""
    if args.N_importance <= 0:
        raise ValueError("run_nerf_view.create_nerf copies the fine net into the coarse one (V:321): "
                         "N_importance must be > 0; use run_nerf.create_nerf otherwise")
    return _R._create_nerf(args, NeRF, True)


# ----------------------------------------------------------------------------- cross-view warp (a12)
def get_rays_ref(directions, c2w):
    """V:553-574: camera-frame directions [..., 3] -> world rays through the reference camera."""
    rays_d = directions @ c2w[:3, :3].T
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    return rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)


def _warp(w2c_ref, intrinsic_ref, point_samples, H, W, flip):
    P = point_samples.reshape(-1, 3)
    return ops.warp_points(P, w2c_ref[0], intrinsic_ref[0].detach().cpu().numpy(), H, W, flip)


def get_ref_rays(w2c_ref, c2w_ref, intrinsic_ref, point_samples, img, depths_h=None, variant="V"):
    """V:576-627 (variant="VT": VT:451-501).  Batch size 1 (the only use in the reference).
    point_samples [1, N_rays, N_samples, 3]; img [1, 3, H, W]; depths_h [1, H, W].
    Returns (rgb_ref [1,3,M], depth_ref [1,1,M], point_samples_cam, rays_o [M,3], rays_d [M,3], mask [1,N])."""
    assert point_samples.shape[0] == 1 and img.shape[0] == 1, "reference only ever uses batch 1"
    _, _, H, W = img.shape
    Xc, px, py, inb = _warp(w2c_ref, intrinsic_ref, point_samples, H, W, variant == "V")
    xs, ys = px[inb], py[inb]
    Kr = intrinsic_ref[0]
    directions = torch.stack([(xs - Kr[0, 2]) / Kr[0, 0], (ys - Kr[1, 2]) / Kr[1, 1], torch.ones_like(xs)], -1)
    rays_o, rays_d = get_rays_ref(directions, c2w_ref[0].to(xs.device))
    yi, xi = ys.long(), xs.long()
    rgb_ref = img[:, :, yi, xi]
    pts_cam = Xc[None] if variant == "V" else Xc[inb]
    mask = inb[None]
    if depths_h is not None:
        return rgb_ref, depths_h.unsqueeze(1)[:, :, yi, xi], pts_cam, rays_o, rays_d, mask
    return rgb_ref, pts_cam, rays_o, rays_d, mask


def get_test_label(w2c_ref, c2w_ref, intrinsic_ref, point_samples, img):
    """V:630-669 -> (pixel_y [1,N], pixel_x [1,N], mask [1,N], cam_z [1,N])."""
    _, _, H, W = img.shape
    Xc, px, py, inb = _warp(w2c_ref, intrinsic_ref, point_samples, H, W, True)
    return py[None], px[None], inb[None], Xc[None, :, 2]


# ----------------------------------------------------------------------------- hard masks (a13)
def compute_hard_masks(H, W, K, poses, depths_cas, i_train, occlusion_threshold=0.1, chunk=5120,
                       device=None, return_thresholds=False):
    """The mask precompute of train() (V:994-1046): for every training view, OR over the other training
    views of (projects in-bounds AND |depth in ref camera - ref depth prior| < threshold), the threshold
    doubled per 5120-pixel chunk until some pixel of the chunk passes.  Non-training views get zeros.
    poses [N,3,4] (c2w), depths_cas [N,H,W].  Returns bool numpy [N,H,W] like `masks_cas`."""
    device = device or _R._default_device()
    poses = np.asarray(poses, np.float32)
    N = poses.shape[0]
    dep = [torch.as_tensor(np.ascontiguousarray(depths_cas[i], np.float32), device=device).reshape(-1)
           for i in range(N)]
    w2c = {}
    for r in i_train:
        c2w = torch.eye(4)
        c2w[:3, :4] = torch.from_numpy(poses[r, :3, :4])
        w2c[r] = torch.inverse(c2w).numpy()          # 4x4 host inverse, as V:1008-1010
    masks, thr = [], {}
    for t in range(N):
        m = torch.zeros(H * W, dtype=torch.uint8, device=device)
        if t in i_train:
            for r in i_train:
                if r == t:
                    continue
                th = ops.hard_mask_pair(H, W, K, poses[t], w2c[r], dep[t], dep[r], occlusion_threshold, chunk, m,
                                        want_thr=return_thresholds)
                if return_thresholds:
                    thr[(t, r)] = th.cpu().numpy()
        masks.append(m.reshape(H, W).bool().cpu().numpy())
    masks = np.stack(masks, 0)
    return (masks, thr) if return_thresholds else masks


# ----------------------------------------------------------------------------- masked losses (a14)
class _MaskedLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, depth, target, prior, mask, far, coef, counts):
        loss, d_rgb, d_depth = ops.masked_loss(rgb, target, depth, prior, mask, far, coef, counts)
        ctx.save_for_backward(d_rgb, d_depth if d_depth is not None else torch.empty(0, device=rgb.device))
        ctx.has_depth = depth is not None
        return loss[0], loss[1]

    @staticmethod
    def backward(ctx, g_rgb_loss, g_depth_loss):
        d_rgb, d_depth = ctx.saved_tensors
        gr = d_rgb * g_rgb_loss if g_rgb_loss is not None else None
        gd = d_depth * g_depth_loss if (ctx.has_depth and g_depth_loss is not None) else None
        return gr, gd, None, None, None, None, None, None


def hardmask_losses(rgb, target, mask, hardmask_coef=0.2, depth=None, depth_prior=None, far=1.0, counts=None):
    """(img_loss, depth_loss) of one level: V:1645-1648 / V:1786-1788 and V:1737 / V:1865.
    mask [B] of 0/1 floats (None = plain img2mse, R:769).  `counts` = (n_masked, n_unmasked) tensor when the
    batch is sharded over ranks (distributed.global_mask_counts) so the means stay global."""
    m = None if mask is None else mask.reshape(-1).to(torch.float32)
    return _MaskedLossFn.apply(rgb, depth, target, depth_prior, m, float(far), float(hardmask_coef), counts)


class _PatchDepthLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth_pred, mono, patch_num, n):
        loss, d = ops.patch_depth_loss(depth_pred, mono, patch_num, n, 1.0, ctx.needs_input_grad[0])
        ctx.save_for_backward(d)
        ctx.shape = depth_pred.shape
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        d, = ctx.saved_tensors
        out = torch.zeros(ctx.shape, device=d.device).reshape(-1)
        out[:d.numel()] = d * g
        return out.reshape(ctx.shape), None, None, None


class _SoftLpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, coef):
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        loss, d_x = ops.soft_lp_loss(x, y, coef, need)
        if need:
            ctx.save_for_backward(d_x)
        return loss

    @staticmethod
    def backward(ctx, g):
        d_x, = ctx.saved_tensors
        gx = d_x * g
        return (gx if ctx.needs_input_grad[0] else None), (-gx if ctx.needs_input_grad[1] else None), None


def img2mse_softLpmask(x, y, coef):
    """V:58, the `--softLpmask` branch of the loss (V:1663-1664 on colours, V:1760-1761 on depths / far): every squared residual
    weighted by |x - y|^coef + 1, normalised by the detached sum of the weights.  One launch (value + gradient seed) for same-shape
    fp32 GPU tensors; the reference's expression on ATen otherwise.  One deliberate difference: where x == y exactly and coef < 1 the
    reference's autograd returns NaN (inf * 0 in the derivative of |d|^coef); the kernel returns the limit, 0."""
    if (torch.is_tensor(x) and torch.is_tensor(y) and x.is_cuda and y.is_cuda and x.shape == y.shape and x.numel() > 0
            and x.dtype == torch.float32 and y.dtype == torch.float32):
        return _SoftLpFn.apply(x.contiguous(), y.contiguous(), float(coef))
    d = x - y
    w = d.abs() ** coef + 1
    return torch.sum(w * d ** 2) / torch.sum(w).detach()


class Temp_Scheduler:
    """V:80-100 — the linear schedule of the pseudo-label noise level (`std_scheduler = Temp_Scheduler(total_iters, 0.2, 0.05,
    temp_min=0.05)`, V:1420): step() -> (1 - epoch / total) (base - min) + min, floored at min; epoch counts the calls (the
    constructor already makes one)."""

    def __init__(self, total_epochs, curr_temp, base_temp, temp_min=0.33, last_epoch=-1):
        self.curr_temp, self.base_temp, self.temp_min = curr_temp, base_temp, temp_min
        self.last_epoch, self.total_epochs = last_epoch, total_epochs
        self.step(last_epoch + 1)

    def step(self, epoch=None):
        # (the reference's step() ignores its argument and always advances by one: kept)
        self.last_epoch += 1
        self.curr_temp = max((1 - self.last_epoch / self.total_epochs) * (self.base_temp - self.temp_min) + self.temp_min,
                             self.temp_min)
        return self.curr_temp


def add_label_noise(rgb, depth_pred, extras, std, far, generator=None):
    """The `--use_noise` block (V:1633-1638): N(0, std) added to the rendered colours of both levels, far * N(0, std) to their depths,
    before the losses.  Returns (rgb, depth_pred, extras) — `extras` is updated in place like the reference's dict.  The draws come
    from the device generator (the reference draws on the CPU and copies: same distribution, another stream)."""
    n = lambda t, s: torch.randn(t.shape, device=t.device, dtype=t.dtype, generator=generator) * s  # noqa: E731
    rgb = rgb + n(rgb, std)
    depth_pred = depth_pred + far * n(depth_pred, std)
    if 'rgb0' in extras:
        extras['rgb0'] = extras['rgb0'] + n(extras['rgb0'], std)
    if 'depth0' in extras:
        extras['depth0'] = extras['depth0'] + far * n(extras['depth0'], std)
    return rgb, depth_pred, extras


def midas_patch_loss(depth_pred, mono_dpt_s, patch_num=4, patch_size=16):
    """`mono_depth_mses` of V:1678-1720 (the monocular-depth patch term; its SSIM / LPIPS neighbours are out of scope):
    the first patch_num * patch_size^2 rays of the batch are the sampled patches (raybank.sample_patch_rays)."""
    return _PatchDepthLossFn.apply(depth_pred, mono_dpt_s, int(patch_num), int(patch_size) * int(patch_size))


# ----------------------------------------------------------------------------- the step's loss as one call
_TERM_NAMES = ("loss", "img_loss", "depth_loss", "patch_loss", "img_loss0", "depth_loss0", "patch_loss0")


def _render_loss_lines(H, W, K, target_s, mask, depth_prior, chunk, rays, coef, far, rgb_w, depth_w, mono, P, ps, patch_w, counts,
                       kwargs):
    """The reference's own sequence (V:1645-1865) on render()'s maps: what render_loss computes, launch by launch."""
    rgb, disp, acc, depth, extras = render(H, W, K, chunk=chunk, rays=rays, **kwargs)
    tgt = target_s.reshape(-1, 3)
    with_depth = depth_prior is not None
    terms = {}

    def level(c, d, suffix):
        il, dl = hardmask_losses(c, tgt, mask, coef, d if with_depth else None, depth_prior, far, counts)
        part = rgb_w * il
        terms["img_loss" + suffix] = il.detach()
        if mono is not None and P > 0:
            pl = midas_patch_loss(d, mono, P, ps)
            part = part + patch_w * pl
            terms["patch_loss" + suffix] = pl.detach()
        if with_depth:
            part = part + depth_w * dl
            terms["depth_loss" + suffix] = dl.detach()
        return part

    loss = level(rgb, depth, "")
    if 'rgb0' in extras:
        loss = loss + level(extras['rgb0'], extras['depth0'], "0")
    terms["loss"] = loss.detach()
    return loss, terms, rgb, disp, acc, depth, extras


def render_loss(H, W, K, target_s, mask=None, depth_prior=None, chunk=1024 * 32, rays=None, hardmask_coef=0.2, depth_far=None,
                rgb_w=1.0, depth_w=1.0, mono=None, patch_num=4, patch_size=16, patch_w=0.001, counts=None, _ss_coins=None, **kwargs):
    """The loss of one run_nerf_view.train() step as ONE call (V:1636-1865 with the terms this package builds):

        rgb, disp, acc, depth_pred, extras = render(H, W, K, chunk=, rays=batch_rays, retraw=True, **render_kwargs_train)
        img_loss   = img2mse(rgb[m == 1], target_s[m == 1]) + hardmask_coef * img2mse(rgb[m == 0], target_s[m == 0])   # V:1645-1648
        loss       = rgb_w * img_loss + patch_w * mono_depth_mses(depth_pred[:P * ps * ps], mono)                   # V:1672-1726
        loss      += depth_w * img2mse(depth_pred[m == 1] / far, depth_prior[m == 1] / far)                         # V:1737
        ... and the same three terms of the coarse level (V:1786-1788, V:1855-1857, V:1865)

    with every term folded into the compositing launches (run_nerf._RenderClossFn): one autograd node from both levels' `raw` to
    the scalar.  mask [B] (0 / 1; None = plain img2mse), depth_prior [B] (None = no depth terms), mono [P * ps * ps] (None = no patch
    term), depth_far = the `far` the depths are divided by (default: the render's far bound), counts = global (n1, n0) for a batch
    sharded over ranks.  -> (loss, terms, rgb, disp, acc, depth, extras); terms: dict of detached 0-d tensors (img_loss, depth_loss,
    patch_loss, img_loss0, ...).  Values equal the lines above (fp64-association round-off on the sums), parameter gradients after
    loss.backward() bit for bit.  Batches beyond one chunk, more than 8 patches, CPU tensors: the lines above, literally."""
    far = float(kwargs.get('far', 1.)) if depth_far is None else float(depth_far)
    n = rays[0].reshape(-1, 3).shape[0] if rays is not None else 0
    P = int(patch_num) if mono is not None else 0
    ps2 = int(patch_size) * int(patch_size)
    tgt = target_s.reshape(-1, 3) if torch.is_tensor(target_s) else None
    ok = (rays is not None and 0 < n <= chunk and tgt is not None and tgt.is_cuda and tgt.dtype == torch.float32 and tgt.shape[0] == n
          and kwargs.get('c2w') is None and P <= 8 and P * ps2 <= n and not torch.is_tensor(kwargs.get('near'))
          and not torch.is_tensor(kwargs.get('far')))
    if _ss_coins is not None:       # the in-loop consistency step's primary terms (ss_step_loss): VT:941-969 on render()'s maps
        if not ok or mask is None:
            rgb, disp, acc, depth, extras = render(H, W, K, chunk=chunk, rays=rays, **kwargs)
            loss, il, il0 = ss_primary_losses(rgb, depth, extras, target_s, depth_prior, None, None,
                                              with_depth_loss=depth_prior is not None, coins=_ss_coins_lines(_ss_coins, depth_prior is not None),
                                              sel=mask)
            return loss, dict(loss=loss.detach(), img_loss=il.detach(), img_loss0=None if il0 is None else il0.detach()), rgb, disp, acc, depth, extras
        spec = ops.ClossSpec(tgt, mask, depth_prior, 1.0, 0.0, 1.0, 1.0, 0.0, None, 0, ps2, None, tuple(_ss_coins))
        rgb, disp, acc, depth, extras = render(H, W, K, chunk=chunk, rays=rays, _target=spec, **kwargs)
        loss, t = extras.pop('loss'), extras.pop('loss_terms')
        return loss, {k: t[i] for i, k in enumerate(_TERM_NAMES)}, rgb, disp, acc, depth, extras
    if not ok:
        return _render_loss_lines(H, W, K, target_s, mask, depth_prior, chunk, rays, hardmask_coef, far, rgb_w, depth_w, mono, P,
                                  patch_size, patch_w, counts, kwargs)
    spec = ops.ClossSpec(tgt, mask, depth_prior, far, hardmask_coef, rgb_w, depth_w, patch_w, mono, P, ps2, counts)
    rgb, disp, acc, depth, extras = render(H, W, K, chunk=chunk, rays=rays, _target=spec, **kwargs)
    loss, t = extras.pop('loss'), extras.pop('loss_terms')
    terms = {k: t[i] for i, k in enumerate(_TERM_NAMES)}
    return loss, terms, rgb, disp, acc, depth, extras


# ----------------------------------------------------------------------------- in-loop consistency (a15)
def _ss_rays(rays_o, rays_d, depth_cas_s, pose_ref, K, image_ref, depth_ref, H, W, render_kwargs, occlusion_threshold):
    """VT:905-925 as ONE launch (ops.ss_ref_rays) + one 16-byte read-back -> the dictionary ss_consistency returns, minus the
    second render."""
    dev = rays_o.device
    c2w_ref, w2c_ref = _ss_pose(pose_ref)
    img = torch.as_tensor(image_ref, dtype=torch.float32).to(dev)
    dep = torch.as_tensor(depth_ref, dtype=torch.float32).to(dev)
    # the warp, the compaction of the in-bounds points (`x[mask]` in the reference: a host sync per boolean index), the reference
    # rays + the rows render() packs from them, the gathered colours / depth priors, |z - D_ref|, the doubling rule's threshold and
    # both masks
    near, far = render_kwargs.get('near', 0.), render_kwargs.get('far', 1.)
    vd, ndc = bool(render_kwargs.get('use_viewdirs', False)), bool(render_kwargs.get('ndc', True))
    scalar_bounds = not (torch.is_tensor(near) or torch.is_tensor(far))
    o = ops.ss_ref_rays(rays_o, rays_d, depth_cas_s, w2c_ref.numpy(), c2w_ref.numpy(), K, H, W, img, dep, float(occlusion_threshold),
                        float(near) if scalar_bounds else 0., float(far) if scalar_bounds else 1., vd, ndc,
                        ndc_coefficients(H, W, K[0][0]) if ndc else (0., 0.), flip=False, want_rows=scalar_bounds)
    if o["M"] == 0:
        raise ops.CnerfError("ss_consistency: no point of the batch projects into the reference view "
                             "(the reference loops forever here)")
    batch_rays_ref = o["rays_od"]
    if o["rows"] is not None:
        from .raybank import PackedRays
        batch_rays_ref._cnerf_packed = PackedRays(o["rows"], H, W, K[0][0], near, far, vd, ndc, src=batch_rays_ref)
    tgt = o["target"]
    return dict(mask_bound=o["inb"].view(torch.bool)[None], mask=o["mask"].view(torch.bool)[:, None], sel=o["sel"],
                threshold=torch.tensor(o["thr"], dtype=torch.float32), batch_rays_ref=batch_rays_ref,
                rgb_target_ref=tgt.t()[None],                      # [1, 3, M] like img[:, :, yi, xi] (a view)
                rays_depth_ref=o["depth_tgt"][None, None],         # [1, 1, M]
                _tgt=tgt, _depth_tgt=o["depth_tgt"])


def _ss_second_render(info, H, W, K, render_kwargs, chunk, with_depth_loss):
    """The second render + its loss terms (VT:927-938): img2mse(rgb_ref, tgt) [+ img2mse(depth_pred_ref, rays_depth_ref)] on both
    levels = render_loss with no mask, un-normalised depths (far 1) and unit weights: every term rides in the compositing launches."""
    loss, _terms, rgb_ref, disp_ref, acc_ref, depth_pred_ref, extras_ref = render_loss(
        H, W, K, info["_tgt"], mask=None, depth_prior=info["_depth_tgt"] if with_depth_loss else None, chunk=chunk,
        rays=info["batch_rays_ref"], hardmask_coef=0.0, depth_far=1.0, rgb_w=1.0, depth_w=1.0, mono=None,
        **dict(render_kwargs, retraw=True))
    info.update(rgb_ref=rgb_ref, depth_pred_ref=depth_pred_ref, extras_ref=extras_ref)
    return loss


def ss_consistency(rays_o, rays_d, depth_cas_s, pose_ref, K, image_ref, depth_ref, H, W, render_kwargs, chunk=1024 * 32,
                   occlusion_threshold=0.1, with_depth_loss=False):
    """The `args.ss_loss` block of run_nerf_view_test.train() (VT:905-938): the batch's depth-prior points
    `rays_o + depth_cas_s * rays_d` are warped into the reference view (`get_ref_rays`, VT variant), the in-bounds ones
    define rays of the reference camera through the snapped pixels; those rays are rendered (a second full render pass)
    and compared with the reference view's colours (and depth prior).  The occlusion mask |z_ref_cam - D_ref| < thr uses
    the reference's doubling rule (thr = occlusion_threshold * 2^k, smallest k that lets some point pass) evaluated on
    the device — the reference's `while mask.sum() == 0` loop costs a host sync per iteration.  Round 5: everything in front of
    the second render is ONE launch (cnerf_ss_ref_rays) and one 16-byte read-back.

    rays_o, rays_d [N, 3], depth_cas_s [N], pose_ref [3, 4] (c2w), image_ref [H, W, 3], depth_ref [H, W].
    Returns a dict: loss (the four VT:930-938 terms), mask_bound [1, N], mask [M, 1], sel [N] (1.0 where mask_bound AND the
    occlusion mask hold: the selection `x[mask_bound][mask]` as a ray weight, for ss_primary_losses), threshold (0-d tensor, the one
    that produced `mask`), batch_rays_ref [2, M, 3], rgb_target_ref [1, 3, M], rays_depth_ref [1, 1, M], and the second
    render's rgb_ref, depth_pred_ref, extras_ref."""
    info = _ss_rays(rays_o, rays_d, depth_cas_s, pose_ref, K, image_ref, depth_ref, H, W, render_kwargs, occlusion_threshold)
    info["loss"] = _ss_second_render(info, H, W, K, render_kwargs, chunk, with_depth_loss)
    return info


def ss_primary_losses(rgb, depth_pred, extras, target_s, depth_cas_s, mask_bound, mask, with_depth_loss=False, coins=None,
                      sel=None):
    """The primary render's loss terms under `args.ss_loss` (VT:941-969), consumers of `ss_consistency`'s masks: each term
    is restricted to the rays `[mask_bound.squeeze()][mask.squeeze()]` (projected into the reference view AND passing the
    occlusion test) when its `random.randint(0, 1)` coin is 1, and is the plain mean (rgb) / absent (depth) otherwise.
    `coins` = the draws in the reference's call order — (rgb, depth, rgb0, depth0) with the depth loss, (rgb, rgb0) without
    — or None to draw them here with `random.randint` like the reference.  The reference's fallback of the COARSE rgb
    term to the FINE rgb when its coin is 0 (VT:959) is kept.  Depth terms are un-normalised MSEs (no /far), as in VT.
    Returns (loss, img_loss, img_loss0); img_loss0 is None when the render has no coarse outputs.

    The double boolean selection becomes one 0/1 ray weight (no host sync, no gathers) fed to the masked-loss kernel; pass
    `sel=ss_consistency(...)['sel']` to take the one its launch wrote instead of re-deriving it from the two masks."""
    import random
    draw = (lambda: random.randint(0, 1)) if coins is None else iter(list(coins)).__next__
    if sel is not None:    # ss_consistency's own launch already formed it (`sel` of its result)
        sel = sel.reshape(-1).to(torch.float32)
    else:
        mb, mk = mask_bound.reshape(-1).bool(), mask.reshape(-1).bool()
        if mk.numel() == 0:
            sel = torch.zeros(mb.shape, device=rgb.device, dtype=torch.float32)
        else:   # sel[i] = mask_bound[i] and mask[rank of i among the in-bounds rays]
            pos = (torch.cumsum(mb.long(), 0) - 1).clamp_(min=0, max=mk.numel() - 1)
            sel = (mb & mk[pos]).to(torch.float32)

    def masked(c, d):      # (masked rgb mse, masked depth mse) of one level in one launch
        return hardmask_losses(c, target_s, sel, 0.0, d, depth_cas_s if d is not None else None, 1.0)

    c_rgb = draw()
    c_dep = draw() if with_depth_loss else 0
    lm = masked(rgb, depth_pred if c_dep else None) if (c_rgb or c_dep) else None
    img_loss = lm[0] if c_rgb else img2mse(rgb, target_s)
    loss = img_loss
    if c_dep:
        loss = loss + lm[1]
    img_loss0 = None
    if 'rgb0' in extras:
        c_rgb0 = draw()
        c_dep0 = draw() if with_depth_loss else 0
        lm0 = masked(extras['rgb0'], extras['depth0'] if c_dep0 else None) if (c_rgb0 or c_dep0) else None
        img_loss0 = lm0[0] if c_rgb0 else img2mse(rgb, target_s)
        loss = loss + img_loss0
        if c_dep0:
            loss = loss + lm0[1]
    return loss, img_loss, img_loss0


def _ss_coins_lines(coins4, with_depth):
    """(rgb, depth, rgb0, depth0) -> the draws in ss_primary_losses' call order."""
    c = [int(bool(x)) for x in coins4]
    return c if with_depth else [c[0], c[2]]


_SS_TERM_NAMES = ("loss", "img_loss", "depth_loss", None, "img_loss0", "depth_loss0", None, "M", "img_loss_ref", "depth_loss_ref",
                  "img_loss0_ref", "depth_loss0_ref")


def _ss_pose(pose_ref):
    c2w_ref = torch.eye(4)
    c2w_ref[:3, :4] = torch.as_tensor(np.asarray(pose_ref.cpu() if isinstance(pose_ref, torch.Tensor) else pose_ref),
                                      dtype=torch.float32)[:3, :4]
    return c2w_ref, torch.inverse(c2w_ref)                 # 4x4 on the host, as VT:910


def _ss_one_render_ok(batch_rays, target_s, depth_cas_s, render_kwargs, chunk):
    """The one-render form of the step needs: a single chunk for the 2N-row batch, N a multiple of the 8 rays of a compositing
    workgroup (the loss partials of a workgroup belong to one segment), scalar bounds, device fp32 inputs, the stock query."""
    rays_o = batch_rays[0]
    n = rays_o.reshape(-1, 3).shape[0]
    t = target_s if torch.is_tensor(target_s) else None
    near, far = render_kwargs.get('near', 0.), render_kwargs.get('far', 1.)
    return (n > 0 and n % 8 == 0 and 2 * n <= chunk and rays_o.is_cuda and t is not None and t.is_cuda and t.dtype == torch.float32
            and t.reshape(-1, 3).shape[0] == n and torch.is_tensor(depth_cas_s) and depth_cas_s.is_cuda
            and not torch.is_tensor(near) and not torch.is_tensor(far) and render_kwargs.get('c2w') is None
            and isinstance(render_kwargs.get('network_fn'), NeRF))


def ss_global_stats(meta, n_local, group=None):
    """The two exchanges a SHARDED `--ss_loss` step needs (VT:917-921 and VT:941-966 are batch-global): from the `meta` of
    this rank's ops.ss_batch call -> (amin_global [1] float: all-reduce MIN of the ranks' minimum |z - D_ref|, the input of the
    threshold-doubling rule), and a function counts3(meta2) -> [3] float: all-reduce SUM of (selected primary rays, primary rays,
    warped rays) once the global threshold has been applied (the `meta` of the second ops.ss_batch call)."""
    import torch.distributed as dist
    amin = meta[4:5].view(torch.float32).clone()
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if on:
        dist.all_reduce(amin, op=dist.ReduceOp.MIN, group=group)

    def counts3(meta2):
        c = torch.stack([meta2[5].to(torch.float32), torch.tensor(float(n_local), device=meta2.device), meta2[0].to(torch.float32)])
        if on:
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        return c
    return amin, counts3


def ss_host_view(info):
    """`ss_step_loss`'s one-render `info` in the reference's shapes (a HOST SYNCHRONISATION: reads M and the threshold back): the
    dictionary of the two-render route — mask [M, 1], batch_rays_ref [2, M, 3], rgb_target_ref [1, 3, M], rays_depth_ref [1, 1, M],
    rgb_ref [M, 3], depth_pred_ref [M], threshold (0-d tensor) ... — for logging, tests and callers written against VT:905-938."""
    if "meta" not in info:
        return info
    meta = info["meta"].cpu()
    M, N = int(meta[0]), info["N"]
    out = dict(info)
    out.update(M=M, threshold=torch.tensor(float(np.int32(int(meta[2])).view(np.float32)), dtype=torch.float32),
               mask=info["occ"][:M].view(torch.bool)[:, None], batch_rays_ref=info["rays_od"][:, :M],
               rgb_target_ref=info["_target2"][N:N + M].t()[None], rays_depth_ref=info["_prior2"][N:N + M][None, None],
               rgb_ref=info["rgb_ref"][:M], depth_pred_ref=info["depth_pred_ref"][:M],
               extras_ref={k: v[:M] for k, v in info["extras_ref"].items()})
    return out


def _ss_step_one_render(H, W, K, batch_rays, target_s, depth_cas_s, pose_ref, image_ref, depth_ref, render_kwargs, chunk,
                        occlusion_threshold, with_depth_loss, coins, global_stats, group):
    """VT:899-969 as ONE render of 2N rows (ops.ss_batch + the two-segment loss tail): no host synchronisation anywhere — the second
    render's ray count M exists only in device memory (the MLP launches are sized for the capacity and stop at the device-side
    count), so the step can be recorded as a hipGraph; 15 launches, all own kernels."""
    rays_o, rays_d = batch_rays[0], batch_rays[1]
    dev = rays_o.device
    c2w_ref, w2c_ref = _ss_pose(pose_ref)
    img = torch.as_tensor(image_ref, dtype=torch.float32).to(dev)
    dep = torch.as_tensor(depth_ref, dtype=torch.float32).to(dev)
    near, far = float(render_kwargs.get('near', 0.)), float(render_kwargs.get('far', 1.))
    vd, ndc = bool(render_kwargs.get('use_viewdirs', False)), bool(render_kwargs.get('ndc', True))
    coef = ndc_coefficients(H, W, K[0][0]) if ndc else (0., 0.)
    args = (rays_o, rays_d, depth_cas_s, target_s, w2c_ref.numpy(), c2w_ref.numpy(), K, H, W, img, dep, float(occlusion_threshold),
            near, far, vd, ndc, coef)
    counts3 = None
    if global_stats is not None:          # the caller ran the exchanges (tests: the shards of one batch on one GPU)
        amin_g, counts3 = global_stats
        b = ops.ss_batch(*args, flip=False, amin_global=amin_g)
    elif group is not None:
        b0 = ops.ss_batch(*args, flip=False)
        amin_g, counts_fn = ss_global_stats(b0["meta"], b0["N"], group)
        b = ops.ss_batch(*args, flip=False, amin_global=amin_g)
        counts3 = counts_fn(b["meta"])
    else:
        b = ops.ss_batch(*args, flip=False)
    N = b["N"]
    spec = ops.ClossSpec(b["target"], b["mask"], b["prior"] if with_depth_loss else None, 1.0, 0.0, 1.0, 1.0, 0.0, None, 0, 256, None,
                         tuple(coins), N, counts3)
    kw = {k: v for k, v in render_kwargs.items() if k not in ('near', 'far', 'ndc', 'use_viewdirs', 'c2w_staticcam', 'c2w')}
    ret = batchify_rays(b["rows"], chunk, _with_depth=True, _target=spec, _live=b["live"], **dict(kw, retraw=True))
    loss, t = ret.pop('loss'), ret.pop('loss_terms')
    terms = {k: t[i] for i, k in enumerate(_SS_TERM_NAMES) if k is not None}
    cut = lambda x: (x[:N], x[N:])  # noqa: E731
    rgb, rgb_ref = cut(ret['rgb_map'])
    disp, _ = cut(ret['disp_map'])
    acc, _ = cut(ret['acc_map'])
    depth, depth_ref_pred = cut(ret['depth_map'])
    extras = {k: v[:N] for k, v in ret.items() if k not in ('rgb_map', 'disp_map', 'acc_map', 'depth_map')}
    extras_ref = {k: v[N:] for k, v in ret.items() if k not in ('rgb_map', 'disp_map', 'acc_map', 'depth_map')}
    info = dict(N=N, loss=loss, terms=terms, live=b["live"], meta=b["meta"], mask_bound=b["inb"].view(torch.bool)[None], occ=b["occ"],
                sel=b["sel"], rank=b["rank"], rays_od=b["rays_od"], depth_diff=b["depth_diff"], rows=b["rows"], _target2=b["target"],
                _prior2=b["prior"], _mask2=b["mask"], rgb=rgb, disp=disp, acc=acc, depth_pred=depth, extras=extras, rgb_ref=rgb_ref,
                depth_pred_ref=depth_ref_pred, extras_ref=extras_ref, img_loss=terms["img_loss"], img_loss0=terms["img_loss0"],
                coins=tuple(coins), route="one_render")
    return loss, info


def ss_step_loss(H, W, K, batch_rays, target_s, depth_cas_s, pose_ref, image_ref, depth_ref, render_kwargs, chunk=1024 * 32,
                 occlusion_threshold=0.1, with_depth_loss=False, coins=None, route=None, global_stats=None, group=None):
    """The whole loss of one `--ss_loss` step of run_nerf_view_test.train() (VT:899-969) as ONE call, at the standard of
    render_loss: the warp / compaction / reference-ray launch FIRST (it depends on the batch only), then the primary render with its
    terms — each behind its `random.randint(0, 1)` coin, restricted to the rays `[mask_bound][mask]` — folded into its compositing
    launches (mask = the launch's `sel`; cnerf_closs_finish_ss), then the second render on the warped rays with its four terms
    folded the same way.  coins = (rgb, depth, rgb0, depth0) or None to draw them here with `random.randint` in the reference's
    order (two draws without the depth loss).  -> (loss, info): info carries ss_consistency's dictionary (`mask_bound`, `mask`,
    `sel`, `threshold`, `batch_rays_ref`, the second render's maps) plus the primary render's `rgb`, `disp`, `acc`, `depth_pred`,
    `extras`, `img_loss`, `img_loss0` and the two partial losses `loss_primary`, `loss_ref`.
    Values: the reference's lines up to summation order (the loss adds the second render's terms first there; here last).

    route (round 6): "one_render" — the primary rays and the warped rays are rendered as ONE batch of 2N rows (ops.ss_batch builds
    it in the warp launch; rows past N + M are padding) with a two-segment loss tail (cnerf_closs_finish_ss2), the loss accumulated
    in the reference's own order: half the MFMA launches, 15 launches in all, and NO host synchronisation — M lives on the device
    only, so `info` holds capacity-N tensors (`ss_host_view(info)` gives the reference's shapes at the cost of a sync) and the step
    can be recorded as a hipGraph (a batch of which NOTHING projects into the reference view — M = 0, where the reference never
    leaves its threshold loop and the two-render route raises — yields the un-masked primary terms only; `info["terms"]["M"]` says
    so); "two_renders" — round 5's form described above (one 32-byte read-back).  None: one_render when
    the batch qualifies (_ss_one_render_ok), else two_renders.
    Sharded batches: `group` = the process group whose ranks each hold a slice of the batch (two small all-reduces: MIN
    of the minimum |z - D_ref| for the threshold rule, SUM of the three ray counts the means divide by), or `global_stats` =
    (amin_global, counts3) when the caller ran them; the per-rank losses then ADD UP to the unsharded loss (GradReducer(mean=False))."""
    import random
    rays_o, rays_d = batch_rays[0], batch_rays[1]
    if coins is None:
        c_rgb = random.randint(0, 1)
        c_dep = random.randint(0, 1) if with_depth_loss else 0
        c_rgb0 = random.randint(0, 1) if render_kwargs.get('N_importance', 0) > 0 else 0
        c_dep0 = random.randint(0, 1) if (with_depth_loss and render_kwargs.get('N_importance', 0) > 0) else 0
        coins = (c_rgb, c_dep, c_rgb0, c_dep0)
    ok = _ss_one_render_ok(batch_rays, target_s, depth_cas_s, render_kwargs, chunk)
    if route == "one_render" and not ok:
        raise ops.CnerfError("ss_step_loss(route='one_render'): the batch does not qualify (N % 8, 2 N <= chunk, scalar bounds, "
                             "device fp32 inputs)")
    if (global_stats is not None or group is not None) and not (ok and route != "two_renders"):
        raise ops.CnerfError("ss_step_loss: sharded batches take the one-render route")
    if ok and route != "two_renders":
        return _ss_step_one_render(H, W, K, batch_rays, target_s, depth_cas_s, pose_ref, image_ref, depth_ref, render_kwargs, chunk,
                                   occlusion_threshold, with_depth_loss, coins, global_stats, group)
    info = _ss_rays(rays_o, rays_d, depth_cas_s, pose_ref, K, image_ref, depth_ref, H, W, render_kwargs, occlusion_threshold)
    lp, terms, rgb, disp, acc, depth, extras = render_loss(
        H, W, K, target_s, mask=info["sel"], depth_prior=depth_cas_s.reshape(-1) if with_depth_loss else None, chunk=chunk,
        rays=batch_rays, depth_far=1.0, mono=None, _ss_coins=tuple(coins), **dict(render_kwargs, retraw=True))
    ls = _ss_second_render(info, H, W, K, render_kwargs, chunk, with_depth_loss)
    loss = ls + lp
    info.update(loss=loss, loss_primary=lp, loss_ref=ls, rgb=rgb, disp=disp, acc=acc, depth_pred=depth, extras=extras,
                img_loss=terms.get("img_loss"), img_loss0=terms.get("img_loss0"), coins=tuple(coins), route="two_renders")
    return loss, info
