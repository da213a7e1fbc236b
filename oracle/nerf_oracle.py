"""CPU oracle for the ConsistentNeRF render / train hot path.  TEST INFRASTRUCTURE ONLY.

This file is an independent restatement (stock PyTorch CPU ops, functional style, weights passed as
plain dicts) of the reference algorithm, used as the *checker* for the HIP kernels and as the timed
`cpu_baseline` ("port") in bench.py.  Only `tests/`, `__graft_entry__.smoke()` and bench.py's
`cpu_baseline` leg may import it; the product package `consistentnerf_amd/` never does and fails
loudly when its HIP library is missing.

Parity is PINNED: `tests/test_oracle_golden.py` checks every function here against the fixtures in
`tests/golden/*.npz`, which `tests/golden/make_golden.py` produced by running the reference's own
Python (imported from /root/reference in the build container) on the same seeded inputs.

Reference citations use H = nerf-pytorch-master/run_nerf_helpers.py, R = run_nerf.py,
V = run_nerf_view.py, VT = run_nerf_view_test.py (all under /root/reference).

All arithmetic is fp32, same operation order as the reference wherever order is observable
(scans, the 1e-10 / 1e-5 guards, division-before-round in the warp).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# a5  positional encoding                                                     H:15-63
# ----------------------------------------------------------------------------------------------
def embed(x: Tensor, n_freqs: int) -> Tensor:
    """gamma(x) = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]; each term spans
    all input coords before the next (H:24-45).  Frequencies are exact powers of two."""
    if n_freqs < 0:
        return x
    parts = [x]
    for k in range(n_freqs):
        xf = x * float(2 ** k)
        parts.append(torch.sin(xf))
        parts.append(torch.cos(xf))
    return torch.cat(parts, dim=-1)


def embed_dim(n_freqs: int) -> int:
    return 3 if n_freqs < 0 else 3 + 6 * n_freqs


# ----------------------------------------------------------------------------------------------
# a6  the MLP                                                                  H:67-130
# ----------------------------------------------------------------------------------------------
@dataclass
class NetCfg:
    D: int = 8
    W: int = 256
    multires: int = 10
    multires_views: int = 4
    use_viewdirs: bool = True
    output_ch: int = 4
    skips: tuple = (4,)

    @property
    def input_ch(self):
        return embed_dim(self.multires)

    @property
    def input_ch_views(self):
        return embed_dim(self.multires_views) if self.use_viewdirs else 0


def _lin(sd: Dict[str, Tensor], name: str, x: Tensor) -> Tensor:
    return torch.addmm(sd[name + ".bias"], x, sd[name + ".weight"].t())


def mlp_forward(sd: Dict[str, Tensor], x_pts: Tensor, x_dir: Optional[Tensor], cfg: NetCfg,
                masks=None, flips: Optional[list] = None) -> Tensor:
    """[M, input_ch] (+ [M, input_ch_views]) -> [M, 4] (or output_ch without viewdirs).
    Trunk of D ReLU layers; after trunk layer i in `skips` the encoded point is concatenated IN FRONT
    of the hidden state (H:110-114); heads per H:116-128: sigma from the trunk, rgb through the
    W/2-wide view branch fed [feature | dir-encoding]; output order [rgb, sigma].

    `masks` (teacher-forced gradient tests only): a list of D (+1 with viewdirs) bool tensors [M, width] — the ReLU pattern to
    APPLY instead of this function's own (z > 0), i.e. h = z * mask.  With the kernel's sign bits passed in, the oracle's
    backward differentiates exactly the piecewise-linear branch the kernel took; wherever the patterns differ the value changes by
    |z| only, and `flips` (a list) receives per ReLU layer (number of units whose own (z > 0) differs from the mask, max |z| there)
    so that a test can assert such units only exist within round-off of zero."""
    def act(z, k):
        if masks is None:
            return torch.relu(z)
        m = masks[k]
        if flips is not None:
            with torch.no_grad():
                diff = (z > 0) != m
                n = int(diff.sum())
                flips.append((n, float(z[diff].abs().max()) if n else 0.0))
        return z * m.to(z.dtype)

    h = x_pts
    for i in range(cfg.D):
        h = act(_lin(sd, f"pts_linears.{i}", h), i)
        if i in cfg.skips:
            h = torch.cat([x_pts, h], dim=-1)
    if not cfg.use_viewdirs:
        return _lin(sd, "output_linear", h)
    sigma = _lin(sd, "alpha_linear", h)
    feat = _lin(sd, "feature_linear", h)
    hv = act(_lin(sd, "views_linears.0", torch.cat([feat, x_dir], dim=-1)), cfg.D)
    rgb = _lin(sd, "rgb_linear", hv)
    return torch.cat([rgb, sigma], dim=-1)


def query(sd: Dict[str, Tensor], pts: Tensor, viewdirs: Optional[Tensor], cfg: NetCfg, masks=None,
          flips: Optional[list] = None) -> Tensor:
    """a4 run_network (R:37-52): pts [B,S,3], viewdirs [B,3] (one per ray, broadcast over samples)
    -> raw [B,S,C].  (The reference's netchunk loop does not change results.)  masks / flips: see mlp_forward."""
    B, S = pts.shape[:2]
    xp = embed(pts.reshape(-1, 3), cfg.multires)
    xd = None
    if cfg.use_viewdirs:
        xd = embed(viewdirs[:, None, :].expand(B, S, 3).reshape(-1, 3), cfg.multires_views)
    out = mlp_forward(sd, xp, xd, cfg, masks, flips)
    return out.reshape(B, S, out.shape[-1])


# ----------------------------------------------------------------------------------------------
# a7  alpha compositing                                                        R:265-308
# ----------------------------------------------------------------------------------------------
def composite(raw: Tensor, z: Tensor, rays_d: Tensor, noise: Optional[Tensor] = None,
              white_bkgd: bool = False):
    """raw [B,S,>=4], z [B,S], rays_d [B,3] -> (rgb_map[B,3], disp_map[B], acc_map[B], weights[B,S],
    depth_map[B]).  Interval widths: z_{i+1}-z_i with a 1e10 tail, scaled by |rays_d| (R:280-283);
    alpha = 1-exp(-relu(sigma+noise)*dist); T_i = prod_{j<i}(1-alpha_j+1e-10) (R:298);
    disp = 1/max(1e-10, depth/acc) (NaN where acc==0, reference behaviour R:302)."""
    B, S = z.shape
    delta = torch.cat([z[:, 1:] - z[:, :-1], torch.full((B, 1), 1e10, dtype=z.dtype)], dim=-1)
    delta = delta * torch.norm(rays_d[:, None, :], dim=-1)
    sigma = raw[..., 3]
    if noise is not None:
        sigma = sigma + noise
    alpha = 1.0 - torch.exp(-torch.relu(sigma) * delta)
    trans = torch.cumprod(torch.cat([torch.ones((B, 1), dtype=z.dtype), 1.0 - alpha + 1e-10], dim=-1), dim=-1)[:, :-1]
    w = alpha * trans
    color = torch.sigmoid(raw[..., :3])
    rgb_map = torch.sum(w[..., None] * color, dim=-2)
    depth_map = torch.sum(w * z, dim=-1)
    acc_map = torch.sum(w, dim=-1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / torch.sum(w, dim=-1))
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    return rgb_map, disp_map, acc_map, w, depth_map


# ----------------------------------------------------------------------------------------------
# a8  inverse-CDF resampling                                                   H:206-250
# ----------------------------------------------------------------------------------------------
def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor):
    """bins [B,Nb], weights [B,Nb-1], u [B,Nf] in [0,1] -> (samples [B,Nf], inds [B,Nf] int64).
    inds = first k with cdf[k] > u (searchsorted right=True, H:234); below/above clamped to
    [0, Nb-1]; flat segments (cdf gap < 1e-5) fall back to denom 1 (H:246-247)."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, dim=-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, dim=-1)], dim=-1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    lo = (inds - 1).clamp(min=0)
    hi = inds.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    b_lo, b_hi = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
    gap = c_hi - c_lo
    gap = torch.where(gap < 1e-5, torch.ones_like(gap), gap)
    t = (u - c_lo) / gap
    return b_lo + t * (b_hi - b_lo), inds


def pytest_uniform(shape):
    """The reference's deterministic RNG hook: np.random.seed(0); np.random.rand(*shape) -> fp32
    (R:376-380, H:221-229, R:290-294).  Re-seeded on every call, like the reference."""
    np.random.seed(0)
    return torch.from_numpy(np.random.rand(*shape).astype(np.float32))


# ----------------------------------------------------------------------------------------------
# a3  per-chunk renderer                                                       R:311-421, V:441-551
# ----------------------------------------------------------------------------------------------
@dataclass
class RenderCfg:
    N_samples: int = 64
    N_importance: int = 0
    perturb: float = 0.0
    lindisp: bool = False
    white_bkgd: bool = False
    raw_noise_std: float = 0.0


def coarse_z(near: Tensor, far: Tensor, Nc: int, lindisp: bool, t_rand: Optional[Tensor]) -> Tensor:
    """R:360-382. near/far [B,1]; t_rand [B,Nc] or None."""
    t = torch.linspace(0.0, 1.0, steps=Nc)
    if lindisp:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    else:
        z = near * (1.0 - t) + far * t
    z = z.expand(near.shape[0], Nc)
    if t_rand is not None:
        mid = 0.5 * (z[:, 1:] + z[:, :-1])
        hi = torch.cat([mid, z[:, -1:]], dim=-1)
        lo = torch.cat([z[:, :1], mid], dim=-1)
        z = lo + (hi - lo) * t_rand
    return z


def render_rays(ray_batch: Tensor, sd_coarse, sd_fine, net: NetCfg, cfg: RenderCfg,
                t_rand: Optional[Tensor] = None, u: Optional[Tensor] = None,
                noise0: Optional[Tensor] = None, noise1: Optional[Tensor] = None,
                retraw: bool = True, net_fine: Optional[NetCfg] = None, z_fine: Optional[Tensor] = None,
                masks_coarse=None, masks_fine=None, flips: Optional[list] = None):
    """ray_batch [B, 8|11] = o, d, near, far, (viewdirs).  Randoms are passed IN (t_rand [B,Nc] for
    the stratified jitter when perturb>0; u [B,Nf] for sample_pdf; noise0/noise1 already scaled by
    raw_noise_std) so oracle and kernels consume identical streams.  If perturb==0, u defaults to
    linspace(0,1,Nf) (det=True, R:396).  Returns the V-style dict (superset of R's: +depth_map,
    +depth0), plus 'z_vals' / 'weights' of the last level for diagnostics."""
    B = ray_batch.shape[0]
    o, d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    vd = ray_batch[:, -3:] if ray_batch.shape[-1] > 8 else None
    near, far = ray_batch[:, 6:7], ray_batch[:, 7:8]
    z = coarse_z(near, far, cfg.N_samples, cfg.lindisp, t_rand if cfg.perturb > 0 else None)
    pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
    raw = query(sd_coarse, pts, vd, net, masks_coarse, flips)
    rgb, disp, acc, w, depth = composite(raw, z, d, noise0, cfg.white_bkgd)
    out = {}
    if cfg.N_importance > 0:
        out.update(rgb0=rgb, disp0=disp, acc0=acc, depth0=depth)
        z_mid = 0.5 * (z[:, 1:] + z[:, :-1])
        if u is None:
            assert cfg.perturb == 0.0
            u = torch.linspace(0.0, 1.0, steps=cfg.N_importance).expand(B, cfg.N_importance)
        z_new, _ = sample_pdf(z_mid, w[:, 1:-1], u)
        z_new = z_new.detach()
        z, _ = torch.sort(torch.cat([z, z_new], dim=-1), dim=-1)
        z_own = z
        if z_fine is not None:   # "teacher forcing" for kernel tests: evaluate the fine level at given depths
            z = z_fine
        pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
        sd2, net2 = (sd_coarse, net) if sd_fine is None else (sd_fine, net_fine or net)
        out["z_own"] = z_own       # this function's own sorted depths (== z unless z_fine overrides them)
        raw = query(sd2, pts, vd, net2, masks_fine, flips)
        rgb, disp, acc, w, depth = composite(raw, z, d, noise1, cfg.white_bkgd)
        out["z_std"] = torch.std(z_new, dim=-1, unbiased=False)
    out.update(rgb_map=rgb, disp_map=disp, acc_map=acc, depth_map=depth, z_vals=z, weights=w)
    if retraw:
        out["raw"] = raw
    return out


def render_rays_pytest(ray_batch, sd_coarse, sd_fine, net: NetCfg, cfg: RenderCfg, **kw):
    """render_rays with the reference's pytest=True RNG (each stream re-seeded with 0)."""
    B = ray_batch.shape[0]
    t_rand = pytest_uniform((B, cfg.N_samples)) if cfg.perturb > 0 else None
    u = None
    if cfg.N_importance > 0:
        if cfg.perturb == 0:  # det=True: np.linspace in float64, cast to fp32 (H:224-226)
            u = torch.from_numpy(np.broadcast_to(np.linspace(0., 1., cfg.N_importance),
                                                 (B, cfg.N_importance)).astype(np.float32).copy())
        else:
            u = pytest_uniform((B, cfg.N_importance))
    n0 = n1 = None
    if cfg.raw_noise_std > 0:
        n0 = pytest_uniform((B, cfg.N_samples)) * cfg.raw_noise_std
        n1 = pytest_uniform((B, cfg.N_samples + cfg.N_importance)) * cfg.raw_noise_std
    return render_rays(ray_batch, sd_coarse, sd_fine, net, cfg, t_rand, u, n0, n1, **kw)


# ----------------------------------------------------------------------------------------------
# ray generation (rows "next" f-2, needed by render(c2w=...))                  H:164-202
# ----------------------------------------------------------------------------------------------
def get_rays(H: int, W: int, K, c2w: Tensor):
    """Pinhole rays, OpenGL camera (looks down -z, +y up).  rays_d = R @ dir computed as a
    broadcast-multiply + sum over the last axis (H:170), NOT a matmul (summation order differs)."""
    jj, ii = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
    dirs = torch.stack([(ii - K[0][2]) / K[0][0], -(jj - K[1][2]) / K[1][1], -torch.ones_like(ii)], dim=-1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], dim=-1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def get_rays_np(H: int, W: int, K, c2w: np.ndarray):
    """numpy twin (H:176-183)."""
    ii, jj = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    dirs = np.stack([(ii - K[0][2]) / K[0][0], -(jj - K[1][2]) / K[1][1], -np.ones_like(ii)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def ndc_rays(H: int, W: int, focal: float, near: float, rays_o: Tensor, rays_d: Tensor):
    """Forward-facing NDC warp (H:186-202)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    ax, ay = -1.0 / (W / (2.0 * focal)), -1.0 / (H / (2.0 * focal))
    o0 = ax * o[..., 0] / o[..., 2]
    o1 = ay * o[..., 1] / o[..., 2]
    o2 = 1.0 + 2.0 * near / o[..., 2]
    d0 = ax * (rays_d[..., 0] / rays_d[..., 2] - o[..., 0] / o[..., 2])
    d1 = ay * (rays_d[..., 1] / rays_d[..., 2] - o[..., 1] / o[..., 2])
    d2 = -2.0 * near / o[..., 2]
    return torch.stack([o0, o1, o2], dim=-1), torch.stack([d0, d1, d2], dim=-1)


def build_ray_batch(rays_o: Tensor, rays_d: Tensor, near: float, far: float, use_viewdirs: bool,
                    ndc: bool = False, H: int = 0, W: int = 0, focal: float = 0.0) -> Tensor:
    """a1 (R:100-125): viewdirs from the PRE-NDC directions, then the optional NDC warp, then the
    [B, 8|11] pack."""
    vd = None
    if use_viewdirs:
        vd = (rays_d / torch.norm(rays_d, dim=-1, keepdim=True)).reshape(-1, 3).float()
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, focal, 1.0, rays_o, rays_d)
    o = rays_o.reshape(-1, 3).float()
    d = rays_d.reshape(-1, 3).float()
    cols = [o, d, near * torch.ones_like(d[:, :1]), far * torch.ones_like(d[:, :1])]
    if vd is not None:
        cols.append(vd)
    return torch.cat(cols, dim=-1)


# ----------------------------------------------------------------------------------------------
# a12  cross-view warp                                    V:576-669 (flip) / VT:451-501 (no flip)
# ----------------------------------------------------------------------------------------------
def warp_points(P: Tensor, w2c: Tensor, K: Tensor, H: int, W: int, flip: bool = True):
    """World points P [N,3] -> reference camera.  Returns (Xc [N,3], x [N], y [N], inb [N] bool):
    Xc = (P R^T + T) (*diag(1,-1,-1) when flip: OpenGL->OpenCV, V:596-597); pixel = round-half-even
    of (K Xc).xy / (K Xc).z; in-bounds is STRICT on x/(W-1), y/(H-1) in (0,1) (V:611-613) so the
    border pixels are excluded."""
    R, T = w2c[:3, :3], w2c[:3, 3]
    Xc = P @ R.t() + T.reshape(1, 3)
    if flip:
        Xc = Xc @ torch.tensor([[1., 0, 0], [0, -1, 0], [0, 0, -1]])
    pix = Xc @ K.t()
    x = (pix[:, 0] / pix[:, 2] + 0.0).round()
    y = (pix[:, 1] / pix[:, 2] + 0.0).round()
    xn, yn = x / (W - 1), y / (H - 1)
    inb = (xn > 0.0) & (xn < 1.0) & (yn > 0.0) & (yn < 1.0)
    return Xc, x, y, inb


def get_ref_rays(w2c: Tensor, c2w: Tensor, K: Tensor, P: Tensor, img: Tensor, depth: Optional[Tensor],
                 flip: bool = True, masked_points: bool = False):
    """img [3,H,W], depth [H,W].  Mirrors the 6-tuple of V:576-627 for a single batch element:
    (rgb_ref [3,M], depth_ref [M], Xc [N,3] (or [M,3] when masked_points, the VT variant), rays_o
    [M,3], rays_d [M,3], inb [N])."""
    H, W = img.shape[-2:]
    Xc, x, y, inb = warp_points(P, w2c, K, H, W, flip)
    xs, ys = x[inb], y[inb]
    dirs = torch.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], torch.ones_like(xs)], dim=-1)
    rays_d = dirs @ c2w[:3, :3].t()
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    yi, xi = ys.long(), xs.long()
    rgb_ref = img[:, yi, xi]
    depth_ref = depth[yi, xi] if depth is not None else None
    return rgb_ref, depth_ref, (Xc[inb] if masked_points else Xc), rays_o, rays_d, inb


def ss_block(rays_o: Tensor, rays_d: Tensor, depth_cas_s: Tensor, c2w_ref: Tensor, K: Tensor, img: Tensor, depth: Tensor,
             occlusion_threshold: float = 0.1):
    """The `args.ss_loss` block in front of the second render (run_nerf_view_test.py VT:905-925): the batch's depth-prior points
    (VT:905) -> get_ref_rays of the VT variant (no axis flip, masked points, VT:451-501) -> the occlusion mask with the threshold
    doubled until one point passes (VT:921-925).  img [3,H,W], depth [H,W], c2w_ref [3,4].  Returns a dict: mask_bound [1,N],
    mask [M,1], thr (the threshold that produced the mask), thr_next (the variable's value when the loop exits: 2 thr),
    rays_ref [2,M,3], rgb_target_ref [1,3,M], rays_depth_ref [1,1,M], sel [N] (float: the rays `x[mask_bound][mask]` selects,
    VT:941-969).  M = 0 loops forever in the reference; here it raises."""
    P = rays_o + depth_cas_s[:, None] * rays_d
    c2w = torch.eye(4)
    c2w[:3, :4] = c2w_ref[:3, :4]
    w2c = torch.inverse(c2w)
    rgb_ref, depth_ref, Xc_m, ro, rd, inb = get_ref_rays(w2c, c2w, K, P, img, depth, flip=False, masked_points=True)
    if ro.shape[0] == 0:
        raise ValueError("no point of the batch projects into the reference view")
    thr = float(occlusion_threshold)
    mask = torch.ones(ro.shape[0], 1) < 0
    thr_used = thr
    while mask.sum() == 0:
        diff = Xc_m[..., -1].unsqueeze(-1) - depth_ref.reshape(-1)[:, None]
        mask = diff.abs() < thr
        thr_used, thr = thr, 2 * thr
    sel = torch.zeros(inb.shape[0])
    sel[inb] = mask.reshape(-1).to(torch.float32)
    return dict(mask_bound=inb[None], mask=mask, thr=thr_used, thr_next=thr, rays_ref=torch.stack([ro, rd], 0),
                rgb_target_ref=rgb_ref[None], rays_depth_ref=depth_ref[None, None], sel=sel)


# ----------------------------------------------------------------------------------------------
# a13  hard-mask assembly                                                      V:994-1046
# ----------------------------------------------------------------------------------------------
def hard_masks(Hh: int, Ww: int, K: np.ndarray, poses: np.ndarray, depths: np.ndarray, i_train,
               thr0: float = 0.1, chunk: int = 5120):
    """Per-pixel mask = OR over reference views of (in-bounds AND |Xc_z - D_ref[y,x]| < thr) with
    thr = thr0 * 2^k, the smallest k>=0 for which at least one pixel OF THAT 5120-PIXEL CHUNK passes
    (the chunking is semantic).  Non-train views get all-zero masks.  Returns (masks [N,H,W] bool,
    thr log [(tgt, ref, chunk, thr)])."""
    N = poses.shape[0]
    Kt = torch.from_numpy(np.asarray(K, np.float32))
    masks, log = [], []
    for t in range(N):
        if t not in i_train:
            masks.append(np.zeros((Hh, Ww), bool))
            continue
        ro, rd = get_rays(Hh, Ww, Kt, torch.from_numpy(poses[t]))
        ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
        dt = torch.from_numpy(depths[t]).reshape(-1)
        P = ro + dt[:, None] * rd
        agg = torch.zeros(Hh * Ww, dtype=torch.bool)
        for r in i_train:
            if r == t:
                continue
            c2w = torch.eye(4)
            c2w[:3, :4] = torch.from_numpy(poses[r])
            w2c = torch.inverse(c2w)
            dref = torch.from_numpy(depths[r])
            for c in range((P.shape[0] + chunk - 1) // chunk):
                sl = slice(c * chunk, (c + 1) * chunk)
                Xc, x, y, inb = warp_points(P[sl], w2c, Kt, Hh, Ww, flip=True)
                thr = float("nan")
                if bool(inb.any()):
                    diff = (Xc[inb][:, 2] - dref[y[inb].long(), x[inb].long()]).abs()
                    thr = thr0
                    while not bool((diff < thr).any()):
                        thr = 2 * thr
                    ok = inb.clone()
                    ok[inb] = diff < thr
                    agg[sl] |= ok
                log.append((t, r, c, thr))
        masks.append(agg.reshape(Hh, Ww).numpy())
    return np.stack(masks), np.array(log, np.float64)


# ----------------------------------------------------------------------------------------------
# a14  masked losses                                                   V:1645-1648, 1737, 1786, 1865
# ----------------------------------------------------------------------------------------------
def mse(a: Tensor, b: Tensor) -> Tensor:
    return torch.mean((a - b) ** 2)


def psnr_from_mse(m: Tensor) -> Tensor:
    return -10.0 * torch.log(m) / torch.log(torch.tensor([10.0]))


def mse_soft_lp(x: Tensor, y: Tensor, coef: float) -> Tensor:
    """img2mse_softLpmask (run_nerf_view.py V:58): squared residuals weighted by |d|^coef + 1, over the DETACHED weight sum."""
    d = x - y
    w = torch.pow(torch.abs(d), coef) + 1
    return torch.sum(w * (d * d)) / torch.sum(w).detach()


def noise_level(total_iters: int, calls: int, base: float = 0.05, floor: float = 0.05) -> float:
    """What the `calls`-th step() of `Temp_Scheduler(total_iters, 0.2, base, temp_min=floor)` returns (V:80-100, V:1420): the
    constructor itself advances the counter once (to epoch 0), every step() by one more."""
    epoch = calls            # epoch 0 is consumed by the constructor: the first step() sees epoch 1
    return max((1 - epoch / total_iters) * (base - floor) + floor, floor)


def masked_rgb_loss(rgb: Tensor, target: Tensor, m: Tensor, coef: float) -> Tensor:
    loss = mse(rgb[m == 1], target[m == 1])
    if m.sum() != m.shape[0]:
        loss = loss + coef * mse(rgb[m == 0], target[m == 0])
    return loss


def masked_depth_loss(depth: Tensor, prior: Tensor, m: Tensor, far: float) -> Tensor:
    return mse(depth[m == 1] / far, prior[m == 1] / far)


def ss_primary_losses(rgb: Tensor, depth_pred: Tensor, rgb0: Optional[Tensor], depth0: Optional[Tensor], target_s: Tensor,
                      depth_cas_s: Tensor, mask_bound: Tensor, mask: Tensor, with_depth_loss: bool, coins):
    """VT:941-969 (the `args.ss_loss` consumers of a15's masks): each primary-render term is restricted to
    `[mask_bound][mask]` when its coin is 1.  coins = the `random.randint(0, 1)` draws in call order: (rgb, depth, rgb0,
    depth0) with the depth loss, (rgb, rgb0) without.  The coarse rgb term falls back to the FINE rgb when its coin is 0
    (VT:959, kept).  Returns (loss, img_loss, img_loss0)."""
    coins = list(coins)
    mb, mk = mask_bound.reshape(-1), mask.reshape(-1)
    pick = lambda x: x[mb][mk]  # noqa: E731
    img_loss = mse(pick(rgb), pick(target_s)) if coins.pop(0) == 1 else mse(rgb, target_s)
    loss = img_loss
    if with_depth_loss:
        loss = loss + (mse(pick(depth_pred), pick(depth_cas_s)) if coins.pop(0) == 1 else 0)
    img_loss0 = None
    if rgb0 is not None:
        img_loss0 = mse(pick(rgb0), pick(target_s)) if coins.pop(0) == 1 else mse(rgb, target_s)
        loss = loss + img_loss0
        if with_depth_loss:
            loss = loss + (mse(pick(depth0), pick(depth_cas_s)) if coins.pop(0) == 1 else 0)
    return loss, img_loss, img_loss0


# ----------------------------------------------------------------------------------------------
# optimiser tail ("next" f-1)                                   R:210, R:780-788, V:1983
# ----------------------------------------------------------------------------------------------
def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, clip: float = 0.0,
              b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    """One torch.optim.Adam step (no weight decay / amsgrad), optional clip_grad_value_ first."""
    if clip > 0:
        g = g.clamp(-clip, clip)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p


def lr_at(lrate: float, global_step: int, lrate_decay: int) -> float:
    """R:784-786."""
    return lrate * (0.1 ** (global_step / (lrate_decay * 1000)))


def as_tensors(sd_np: Dict[str, np.ndarray], requires_grad: bool = False) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(requires_grad)
            for k, v in sd_np.items()}


# ----------------------------------------------------------------------------------------------
# training-ray supply: ray bank + batching (R:677-701, R:720-729), --no_batching sampler (R:730-757)
def ray_bank(images: np.ndarray, poses: np.ndarray, H: int, W: int, K, i_train, perm=None) -> np.ndarray:
    """rays_rgb [(len(i_train)*H*W), 3, 3] = (rays_o, rays_d, rgb) per training pixel, rows permuted by `perm`
    (the permutation np.random.shuffle applies, R:692)."""
    rows = []
    for i in i_train:
        ro, rd = get_rays_np(H, W, K, np.asarray(poses[i])[:3, :4])
        rows.append(np.stack([ro.reshape(-1, 3), rd.reshape(-1, 3), np.asarray(images[i])[..., :3].reshape(-1, 3)], 1))
    bank = np.concatenate(rows, 0).astype(np.float32)
    return bank if perm is None else bank[np.asarray(perm)]


def numpy_shuffle_perm(n: int, seed: int) -> np.ndarray:
    """Row permutation of `np.random.seed(seed); np.random.shuffle(x)` for an x with n rows."""
    idx = np.arange(n)
    np.random.seed(seed)
    np.random.shuffle(idx)
    return idx


class BankBatches:
    """R:720-729: consecutive N_rand-row slices; when the cursor passes the end, permute by rand_idx and restart."""

    def __init__(self, bank: np.ndarray):
        self.bank, self.i_batch = bank, 0

    def next(self, N_rand: int, rand_idx=None):
        batch = np.transpose(self.bank[self.i_batch:self.i_batch + N_rand], (1, 0, 2))
        self.i_batch += N_rand
        if self.i_batch >= self.bank.shape[0]:
            self.bank = self.bank[np.asarray(rand_idx)]
            self.i_batch = 0
        return batch[:2], batch[2]


def crop_coords(H: int, W: int, precrop_frac=None) -> np.ndarray:
    """(row, col) grid the --no_batching sampler draws from (R:741-753), row-major."""
    if precrop_frac is not None:
        dH, dW = int(H // 2 * precrop_frac), int(W // 2 * precrop_frac)
        rr, cc = np.arange(H // 2 - dH, H // 2 + dH), np.arange(W // 2 - dW, W // 2 + dW)
    else:
        rr, cc = np.arange(H), np.arange(W)
    return np.stack(np.meshgrid(rr, cc, indexing="ij"), -1).reshape(-1, 2)


def sample_image_rays(target: np.ndarray, pose: np.ndarray, H: int, W: int, K, select_inds, precrop_frac=None):
    """R:730-757 with the random pixel indices given."""
    ro, rd = get_rays_np(H, W, K, np.asarray(pose)[:3, :4])
    sc = crop_coords(H, W, precrop_frac)[np.asarray(select_inds)]
    return np.stack([ro[sc[:, 0], sc[:, 1]], rd[sc[:, 0], sc[:, 1]]], 0), np.asarray(target)[sc[:, 0], sc[:, 1], :3]


# ---- f-2 / f-5: patch sampler and the monocular-depth patch term of run_nerf_view.train() -----------------------------
def patch_coords(starts, patch_size: int = 16) -> np.ndarray:
    """V:1471-1503: pixel (row, col) lists of the sampled patches, in the reference's order — the FIRST index runs
    fastest within a patch (its meshgrid is 'xy' and the pair is used as (row, col)) -> [P * ps * ps, 2]."""
    out = []
    for x0, y0 in np.asarray(starts).reshape(-1, 2):
        k = np.arange(patch_size * patch_size)
        out.append(np.stack([x0 + k % patch_size, y0 + k // patch_size], -1))
    return np.concatenate(out, 0)


def draw_patch_starts(H: int, W: int, n_patches: int = 4, patch_size: int = 16, precrop=None) -> np.ndarray:
    """The np.random draws of V:1477-1488 (global numpy RNG, two randint per patch; the 'fewer than 257 white pixels'
    test at V:1497 always passes for a 256-pixel patch, so no draw is ever rejected).  `precrop` = (dH, dW) during the
    pre-crop iterations — the reference bounds the column start by H//2 - dH there (sic)."""
    s = []
    for _ in range(n_patches):
        if precrop is not None:
            dH, dW = precrop
            x0 = np.random.randint(H // 2 - dH, H // 2 + dH - patch_size, size=(1, 1, 1))
            y0 = np.random.randint(H // 2 - dH, W // 2 + dW - patch_size, size=(1, 1, 1))
        else:
            x0 = np.random.randint(0, H - patch_size + 1, size=(1, 1, 1))
            y0 = np.random.randint(0, W - patch_size + 1, size=(1, 1, 1))
        s.append((int(x0.item()), int(y0.item())))
    return np.asarray(s, dtype=np.int64)


def patch_depth_loss(depth_pred: Tensor, mono: Tensor, patch_num: int = 4, n: int = 256) -> Tensor:
    """V:1678-1720 without the SSIM / LPIPS lines: sum over patches of the shift-aligned, min-max normalised
    squared difference between clipped inverse rendered depth and the monocular prior, / patch_num / 2."""
    one = torch.ones(1)
    clip = 1 / torch.where(depth_pred <= 0, 0.0001 * one, depth_pred)
    total = 0.0
    for p in range(patch_num):
        pr = torch.nan_to_num(clip[p * n:(p + 1) * n])
        gt = torch.nan_to_num(mono[p * n:(p + 1) * n])
        m = torch.where(gt > 0, one, torch.zeros(1))
        lo = torch.where(gt > 0, gt, one * 10 ** 5).min()
        gt = m * (gt - lo) / (gt.max() - lo + 0.0001)
        lo = torch.where(m * pr > 0, pr, one * 10 ** 5).min()
        pr = m * (pr - lo) / ((m * pr).max() - lo + 0.0001)
        alpha = (pr - gt).mean()
        total = total + ((gt - pr + alpha) ** 2).mean() / patch_num / 2
    return total
