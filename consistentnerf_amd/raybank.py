"""Training-ray supply of the reference's train() loops, on the device.

  RayBank        R:677-701 + R:720-729  all rays of the training views + their colours, shuffled; N_rand-row batches;
                                        reshuffle when an epoch ends
  sample_image_rays   R:730-757         the `--no_batching` branch: N_rand random pixels of one image, optional centre
                                        pre-crop for the first iterations

The reference builds the bank with numpy on the host (get_rays_np per pose, concatenate, transpose, reshape, one
np.random.shuffle) and ships 3 x 3 x 4 bytes per ray to the GPU; here the rays come out of `cnerf_gen_rays` on the
device and only the images cross PCIe.  The random choices stay the caller's: pass the permutation / pixel indices
the reference's RNG would have produced (tests do) or let the device draw them.
"""
import numpy as np
import torch

from . import ops
from .run_nerf_helpers import ndc_coefficients


class PackedRays:
    """The [B, 8|11] rows render() would assemble from a (rays_o, rays_d) batch (R:100-125), written by the sampler's own launch
    for the camera / bounds named here; attached to the `batch_rays` tensor as `_cnerf_packed`, taken by run_nerf._ray_batch when
    the render() call asks for exactly these bounds (otherwise the rays are packed again, as for any caller's batch)."""
    __slots__ = ("rows", "H", "W", "focal", "near", "far", "use_viewdirs", "ndc", "src_ptr", "src_version")

    def __init__(self, rows, H, W, focal, near, far, use_viewdirs, ndc, src=None):
        self.rows, self.H, self.W, self.focal = rows, int(H), int(W), float(focal)
        self.near, self.far, self.use_viewdirs, self.ndc = float(near), float(far), bool(use_viewdirs), bool(ndc)
        # the tensor the rows were packed FROM, as it was then: an in-place change of it afterwards (buf.copy_(new_rays), a scaled or
        # perturbed direction) bumps its version counter and the packed rows are no longer taken
        self.src_ptr = None if src is None else src.data_ptr()
        self.src_version = None if src is None else src._version

    def matches(self, H, W, K, near, far, use_viewdirs, ndc, device, src=None):
        if torch.is_tensor(near) or torch.is_tensor(far):
            return False
        if src is not None and self.src_ptr is not None and (src.data_ptr() != self.src_ptr or src._version != self.src_version):
            return False         # the batch tensor was modified in place since its rows were packed: pack its current contents
        return (self.rows.device == torch.device(device) and float(near) == self.near and float(far) == self.far
                and bool(use_viewdirs) == self.use_viewdirs and bool(ndc) == self.ndc
                and (not self.ndc or (int(H) == self.H and int(W) == self.W and float(K[0][0]) == self.focal)))


def _pack_args(render_kwargs):
    """(near, far, use_viewdirs, ndc) of the render() call the batch is for (render's own defaults, R:70-71)."""
    if render_kwargs is None:
        return None
    return (float(render_kwargs.get("near", 0.)), float(render_kwargs.get("far", 1.)),
            bool(render_kwargs.get("use_viewdirs", False)), bool(render_kwargs.get("ndc", True)))


def _sample(target, pose, H, W, K, N_rand, patch_starts, patch_size, select_inds, precrop_frac, extras, render_kwargs):
    """One launch (cnerf_sample_pixels) for everything the two samplers below return."""
    dev = target.device if isinstance(target, torch.Tensor) and target.is_cuda else torch.device(
        "cuda", torch.cuda.current_device())
    image = torch.as_tensor(target, dtype=torch.float32).to(dev)
    H, W = int(H), int(W)
    if precrop_frac is not None:
        dH, dW = int(H // 2 * precrop_frac), int(W // 2 * precrop_frac)
        crop = (H // 2 - dH, W // 2 - dW, 2 * dH, 2 * dW)
    else:
        crop = (0, 0, H, W)
    pk = _pack_args(render_kwargs)
    near, far, vd, ndc = pk if pk is not None else (0., 1., False, False)
    coef = ndc_coefficients(H, W, K[0][0]) if ndc else (0., 0.)
    sel = None
    if select_inds is not None:
        # caller-supplied pixel indices go straight into image / prior addresses: range-checked here (the kernel clamps as well) —
        # on the host when they are host data, by a device-side assertion (no synchronisation) when they already live on the GPU
        n_grid = crop[2] * crop[3]
        if isinstance(select_inds, torch.Tensor) and select_inds.is_cuda:
            sel = select_inds.to(device=dev, dtype=torch.long)
            if sel.numel():
                torch._assert_async(((sel >= 0) & (sel < n_grid)).all())
        else:
            arr = np.asarray(select_inds.cpu() if isinstance(select_inds, torch.Tensor) else select_inds).astype(np.int64).reshape(-1)
            if arr.size and (arr.min() < 0 or arr.max() >= n_grid):
                raise IndexError(f"select_inds must lie in [0, {n_grid}) (the {crop[2]} x {crop[3]} pixel grid); got "
                                 f"[{arr.min()}, {arr.max()}]")
            sel = torch.as_tensor(arr, device=dev, dtype=torch.long)
    n_rand = int(N_rand) if sel is None else int(sel.numel())
    rng = ops.rng_draw(dev) if sel is None and n_rand > 0 else None
    ex = [torch.as_tensor(e, dtype=torch.float32).to(dev) for e in extras]
    rows, od, tgt, ex_out, coords = ops.sample_pixels(H, W, K, np.asarray(pose.cpu() if isinstance(pose, torch.Tensor) else pose)[:3, :4],
                                                      near, far, vd, ndc, coef, crop, patch_starts, patch_size, n_rand, sel, rng,
                                                      image, ex, want_rows=pk is not None)
    if rows is not None:
        od._cnerf_packed = PackedRays(rows, H, W, K[0][0], near, far, vd, ndc, src=od)
    return od, tgt, coords, ([ex_out[i] for i in range(len(ex))] if ex else [])


def _perm_like_numpy_shuffle(n, seed):
    """The row permutation `np.random.seed(seed); np.random.shuffle(rows)` applies (Fisher-Yates on the row index)."""
    idx = np.arange(n)
    np.random.seed(seed)
    np.random.shuffle(idx)
    return idx


class RayBank:
    """rays_rgb [(len(i_train) * H * W), 3, 3] = (rays_o, rays_d, rgb) per training pixel (R:683-690)."""

    def __init__(self, images, poses, H, W, K, i_train, device=None, perm=None, seed=None):
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.H, self.W = int(H), int(W)
        imgs = torch.as_tensor(np.asarray(images)[..., :3] if not isinstance(images, torch.Tensor) else images[..., :3],
                               dtype=torch.float32)
        rows = []
        for i in i_train:
            c2w = torch.as_tensor(np.asarray(poses[i])[:3, :4] if not isinstance(poses, torch.Tensor)
                                  else poses[i, :3, :4], dtype=torch.float32)
            r = ops.gen_rays(self.H, self.W, K, c2w, 0., 1., False, False, dev)          # [H*W, 11] o, d, near, far, vd
            rgb = imgs[i].reshape(-1, 3).to(dev)
            rows.append(torch.stack([r[:, 0:3], r[:, 3:6], rgb], 1))                        # [H*W, 3, 3]
        self.rays_rgb = torch.cat(rows, 0)
        n = self.rays_rgb.shape[0]
        if perm is None:
            perm = _perm_like_numpy_shuffle(n, seed) if seed is not None else None          # R:692
        if perm is not None:
            self.rays_rgb = self.rays_rgb[torch.as_tensor(perm, device=dev, dtype=torch.long)]
        else:
            self.rays_rgb = self.rays_rgb[torch.randperm(n, device=dev)]
        self.i_batch = 0
        self.epochs = 0

    def __len__(self):
        return self.rays_rgb.shape[0]

    def next_batch(self, N_rand, rand_idx=None):
        """R:720-729 -> (batch_rays [2, B, 3], target_s [B, 3]); `rand_idx` = the epoch-end permutation to use."""
        batch = self.rays_rgb[self.i_batch:self.i_batch + N_rand].transpose(0, 1)
        batch_rays, target_s = batch[:2], batch[2]
        self.i_batch += N_rand
        if self.i_batch >= self.rays_rgb.shape[0]:
            if rand_idx is None:
                rand_idx = torch.randperm(self.rays_rgb.shape[0], device=self.rays_rgb.device)
            self.rays_rgb = self.rays_rgb[torch.as_tensor(rand_idx, device=self.rays_rgb.device, dtype=torch.long)]
            self.i_batch = 0
            self.epochs += 1
        return batch_rays, target_s


def crop_coords(H, W, precrop_frac=None):
    """Pixel grid the sampler draws from (R:741-753): centre crop of 2*dH x 2*dW or the whole image -> [n, 2] (row, col)."""
    if precrop_frac is not None:
        dH, dW = int(H // 2 * precrop_frac), int(W // 2 * precrop_frac)
        rr = torch.arange(H // 2 - dH, H // 2 + dH)
        cc = torch.arange(W // 2 - dW, W // 2 + dW)
    else:
        rr, cc = torch.arange(H), torch.arange(W)
    return torch.stack(torch.meshgrid(rr, cc, indexing="ij"), -1).reshape(-1, 2)


_COORDS = {}


def _coords_on(dev, H, W, precrop_frac):
    """The (cropped) pixel grid on the device, built once per (size, crop): the reference rebuilds it every step on the
    host (R:741-753), which here would be a 3 MB host-to-device copy per step at LLFF size."""
    key = (str(dev), int(H), int(W), precrop_frac)
    if key not in _COORDS:
        _COORDS[key] = crop_coords(int(H), int(W), precrop_frac).to(dev)
    return _COORDS[key]


def sample_image_rays(target, pose, H, W, K, N_rand, precrop_frac=None, select_inds=None, render_kwargs=None):
    """R:730-757: N_rand distinct pixels of one image -> (batch_rays [2, B, 3], target_s [B, 3]).
    `select_inds` = indices into the (cropped) pixel grid (the reference's own `np.random.choice`); default: a device-side draw
    without replacement inside the sampling launch (a keyed permutation of the grid, csrc/sampler.hip) — one launch either way.
    `render_kwargs` (the dict the batch will be rendered with): the launch also writes the [B, 8|11] rows render() would pack."""
    od, tgt, _, _ = _sample(target, pose, H, W, K, N_rand, None, 1, select_inds, precrop_frac, (), render_kwargs)
    return od, tgt


# ---- patch sampler of run_nerf_view.train() (V:1471-1517) ------------------------------------------------------------
def draw_patch_starts(H, W, n_patches=4, patch_size=16, precrop=None):
    """Top-left corners of the `n_patches` patches, drawn from numpy's global RNG exactly as V:1477-1488 does (two
    randint per patch; the white-background test at V:1497 can never reject a 256-pixel patch).  `precrop` = (dH, dW)
    while i < precrop_iters; the column start is bounded below by H//2 - dH there, as in the reference."""
    H, W = int(H), int(W)
    starts = np.empty((n_patches, 2), dtype=np.int64)
    for k in range(n_patches):
        if precrop is not None:
            dH, dW = precrop
            starts[k, 0] = np.random.randint(H // 2 - dH, H // 2 + dH - patch_size, size=(1, 1, 1)).item()
            starts[k, 1] = np.random.randint(H // 2 - dH, W // 2 + dW - patch_size, size=(1, 1, 1)).item()
        else:
            starts[k, 0] = np.random.randint(0, H - patch_size + 1, size=(1, 1, 1)).item()
            starts[k, 1] = np.random.randint(0, W - patch_size + 1, size=(1, 1, 1)).item()
    return starts


def _host_to_device(a, dtype, device):
    """A small host array to the device WITHOUT stalling the host: staged through pinned memory and copied stream-ordered
    (a pageable-memory copy makes the host wait for everything queued on the stream — one full GPU drain per training step when the
    array is the step's patch corners)."""
    t = torch.as_tensor(a, dtype=dtype)
    if device is None or torch.device(device).type != "cuda" or t.is_cuda:
        return t if device is None else t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def patch_coords(starts, patch_size=16, device=None):
    """[P, 2] corners -> [P * ps * ps, 2] (row, col); within a patch the ROW index runs fastest (V:1490-1494)."""
    s = _host_to_device(np.asarray(starts), torch.long, device).reshape(-1, 1, 2)
    k = torch.arange(patch_size * patch_size, device=device)
    off = torch.stack([k % patch_size, k // patch_size], -1)[None]
    return (s + off).reshape(-1, 2)


def sample_patch_rays(target, pose, H, W, K, N_rand, patch_starts, select_inds=None, precrop_frac=None, patch_size=16,
                      extras=(), render_kwargs=None):
    """V:1452-1517: the P patches' pixels first, then N_rand distinct random pixels of the (cropped) grid ->
    (batch_rays [2, P*ps*ps + N_rand, 3], target_s, select_coords [.., 2], [e[select_coords] for e in extras])
    — `extras` are per-pixel maps sampled at the same pixels (depth priors, hard mask, monocular depth).  ONE launch
    (cnerf_sample_pixels): the patch corners travel as kernel arguments, the random pixels are `select_inds` or drawn in the kernel,
    rays / colours / maps are produced for the chosen pixels only (round 4: a full-image ray generation, a randperm over the grid
    and ~15 ATen index / cat / stack kernels per step).  `render_kwargs`: see sample_image_rays."""
    return _sample(target, pose, H, W, K, N_rand, patch_starts, patch_size, select_inds, precrop_frac, extras, render_kwargs)
