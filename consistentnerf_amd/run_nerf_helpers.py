"""Drop-in for the reference's run_nerf_helpers.py (H): same names, argument meaning and return
structure, backed by the HIP kernels of libcnerf_hip.so.

  Embedder / get_embedder  H:15-63     NeRF  H:67-130      get_rays / get_rays_np / ndc_rays  H:164-202
  sample_pdf  H:206-250                img2mse / mse2psnr / to8b  H:9-11

Differences a caller can observe (by design):
  * tensors must live on an MI355X; there is no CPU execution path (CnerfError otherwise);
  * `NeRF` parameters keep the reference names/shapes (checkpoints interchange); the module is evaluated by the
    fused encoding+MLP kernel — through `run_nerf.run_network` on raw points (render path, encodings never
    materialised) or through `model(x)` on a pre-embedded [M, 90] batch like the reference.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .ops import NetSpec

# Misc (H:9-11)
class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        loss, d_x = ops.mse(x, y, need)
        if need:
            ctx.save_for_backward(d_x)
        return loss

    @staticmethod
    def backward(ctx, g):
        d_x, = ctx.saved_tensors
        gx = d_x * g
        return (gx if ctx.needs_input_grad[0] else None), (-gx if ctx.needs_input_grad[1] else None)


def img2mse(x, y):
    """H:9 `torch.mean((x - y) ** 2)`.  Same-shape fp32 GPU tensors (every use on the hot path: [N_rays, 3] colours,
    [N_rays] depths) take one fused kernel that also emits the gradient seed; anything else (broadcasting, other dtypes)
    is the reference's expression on ATen's GPU kernels."""
    if (torch.is_tensor(x) and torch.is_tensor(y) and x.is_cuda and y.is_cuda and x.shape == y.shape and x.numel() > 0
            and x.dtype == torch.float32 and y.dtype == torch.float32):
        return _MseFn.apply(x, y)
    return torch.mean((x - y) ** 2)


mse2psnr = lambda x: -10. * torch.log(x) / torch.log(torch.tensor([10.], device=x.device))  # noqa: E731
to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)  # noqa: E731


class Embedder:
    """Positional encoding descriptor (H:15-45).  Only the configuration create_nerf uses is supported:
    include_input, log-sampled power-of-two bands, [sin, cos]."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d = kwargs.get('input_dims', 3)
        if d != 3 or not kwargs.get('include_input', True) or not kwargs.get('log_sampling', True):
            raise NotImplementedError("only the create_nerf embedder configuration is compiled")
        self.multires = int(kwargs['num_freqs'])
        if int(kwargs['max_freq_log2']) != self.multires - 1:
            raise NotImplementedError("bands must be 2**linspace(0, L-1, L)")
        self.out_dim = 3 + 6 * self.multires

    def embed(self, inputs):
        return ops.embed(inputs, self.multires)

    __call__ = embed


class _Identity(nn.Identity):
    multires = -1
    out_dim = 3


def get_embedder(multires, i=0):
    if i == -1:
        return _Identity(), 3
    e = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                 log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return e, e.out_dim


def _multires_of(dim: int) -> int:
    if dim == 3:
        return -1
    if dim < 3 or (dim - 3) % 6:
        raise ValueError(f"input width {dim} is not 3+6L")
    return (dim - 3) // 6


class NeRF(nn.Module):
    """Same constructor, parameter names and shapes as H:67-104 (incl. the three ConsistentNeRF
    scalars temp_rgb / temp_depth / depth_scale, H:79-84)."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False,
                 coarse=False, stable_init=False):
        super().__init__()
        self.D, self.W = D, W
        self.input_ch, self.input_ch_views = input_ch, input_ch_views
        self.skips, self.use_viewdirs, self.coarse = skips, use_viewdirs, coarse
        self.output_ch = output_ch
        self.temp_rgb = nn.Parameter(torch.full((1,), -0.7))
        self.temp_depth = nn.Parameter(torch.full((1,), -0.7))
        self.depth_scale = nn.Parameter(torch.full((1,), 1.0))
        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)] + [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + input_ch, W)
                                        for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)
        if stable_init:
            nn.init.uniform_(self.alpha_linear.bias)
        if len(skips) > 1:
            raise NotImplementedError("one skip connection (create_nerf uses skips=[4])")

    # ---- kernel-facing view of the module
    def spec(self) -> NetSpec:
        skip = self.skips[0] if len(self.skips) else -1
        return NetSpec(D=self.D, W=self.W, multires=_multires_of(self.input_ch),
                       multires_views=_multires_of(self.input_ch_views) if self.use_viewdirs else 4,
                       use_viewdirs=bool(self.use_viewdirs), output_ch=self.output_ch, skip=skip)

    def kernel_tensors(self):
        """Parameters in the C-ABI order of include/cnerf.h (state-dict order minus the scalars)."""
        ts = []
        for l in self.pts_linears:
            ts += [l.weight, l.bias]
        ts += [self.views_linears[0].weight, self.views_linears[0].bias]
        if self.use_viewdirs:
            for l in (self.feature_linear, self.alpha_linear, self.rgb_linear):
                ts += [l.weight, l.bias]
        else:
            ts += [self.output_linear.weight, self.output_linear.bias]
        return ts

    # OPT-IN: "bf16" | "bf16x2" | "bf16x3" run the INFERENCE forward (no autograd graph) on the bf16 matrix cores with the
    # operands split into 1 / 2 / 3 bf16 planes and fp32 accumulation (csrc/mlp_fwd_bf.hip).  Training and everything
    # under autograd always use the exact-fp32 kernels; so does this module unless the attribute is set.
    inference_precision = "fp32"

    def invalidate_packed(self):
        """Forget the kernel-layout copy of the weights.  The cache key (run_nerf._packed) sees optimizer steps,
        load_state_dict() and every autograd-visible in-place edit, but NOT writes through `p.data` (manual EMA /
        re-initialisation code: `p.data.mul_()`, `p.data.copy_()`), which bump no version counter — call this after such
        an edit."""
        self.__dict__.pop("_cnerf_packed", None)
        self.__dict__.pop("_cnerf_packed_bf", None)

    def forward(self, x):
        """H:107-130 on an already-embedded batch x[..., input_ch + input_ch_views] -> [..., 4 | output_ch]
        (differentiable w.r.t. the parameters).  The render path does not come through here: run_network feeds
        the kernel raw points and the encodings never exist in HBM."""
        from .run_nerf import _MlpFn
        lead = x.shape[:-1]
        width = self.input_ch + (self.input_ch_views if self.use_viewdirs else 0)
        x2 = x.reshape(-1, x.shape[-1])[:, :width].contiguous()
        params = self.kernel_tensors()
        if not torch.is_grad_enabled():     # (see run_network: no stash for inference)
            params = [p.detach() for p in params]
        out = _MlpFn.apply(self, x2.shape[0], 1, None, None, None, None, x2, None, 0, *params)
        return out.reshape(*lead, out.shape[-1])

    def load_weights_from_keras(self, weights):
        """H:132-159."""
        assert self.use_viewdirs, "Not implemented if use_viewdirs=False"
        dev = self.pts_linears[0].weight.device

        def put(lin, iw):
            lin.weight.data = torch.from_numpy(np.transpose(weights[iw])).to(dev)
            lin.bias.data = torch.from_numpy(np.transpose(weights[iw + 1])).to(dev)
        for i in range(self.D):
            put(self.pts_linears[i], 2 * i)
        put(self.feature_linear, 2 * self.D)
        put(self.views_linears[0], 2 * self.D + 2)
        put(self.rgb_linear, 2 * self.D + 4)
        put(self.alpha_linear, 2 * self.D + 6)


# Ray helpers
def _dev(t=None):
    if isinstance(t, torch.Tensor) and t.is_cuda:
        return t.device
    return torch.device("cuda", torch.cuda.current_device())


def get_rays(H, W, K, c2w):
    """H:164-173 -> (rays_o [H,W,3], rays_d [H,W,3]) on the GPU."""
    rays = ops.gen_rays(H, W, K, c2w, 0., 1., False, False, _dev(c2w))
    return rays[:, 0:3].reshape(H, W, 3), rays[:, 3:6].reshape(H, W, 3)


def get_rays_np(H, W, K, c2w):
    """H:176-183 (host-side numpy in the reference as well: it builds the CPU ray bank, R:680-684)."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def ndc_coefficients(H, W, focal):
    """The two Python-scalar coefficients of H:193-199, evaluated with the caller's own scalar types."""
    return -1. / (W / (2. * focal)), -1. / (H / (2. * focal))


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """H:186-202 (near plane 1, the only value render() passes, R:116)."""
    if float(near) != 1.0:
        raise NotImplementedError("ndc_rays is compiled for the near plane render() uses (1.0)")
    sh = rays_d.shape
    r = ops.pack_rays(rays_o, rays_d, 0., 1., False, True, ndc_coefficients(H, W, focal))
    return r[:, 0:3].reshape(sh), r[:, 3:6].reshape(sh)


def pytest_uniform(shape, device):
    """The reference's deterministic hook: np.random.seed(0); np.random.rand(*shape) (R:376-380,
    H:221-229, R:290-294)."""
    np.random.seed(0)
    return torch.from_numpy(np.random.rand(*shape).astype(np.float32)).to(device)


_DET_U = {}


def sample_u(B, N_samples, det, pytest, device):
    """u for sample_pdf (H:214-229)."""
    if pytest:
        if det:
            u = np.broadcast_to(np.linspace(0., 1., N_samples), (B, N_samples)).astype(np.float32)
            return torch.from_numpy(u[:1].copy()).to(device)   # identical rows: broadcast in the kernel
        return pytest_uniform((B, N_samples), device)
    if det:   # evaluated on the CPU like the reference's constant, cached per device (no H2D copy per chunk; graph-capturable)
        key = (N_samples, str(device))
        if key not in _DET_U:
            _DET_U[key] = torch.linspace(0., 1., steps=N_samples).to(device)[None]
        return _DET_U[key]
    return torch.rand(B, N_samples, device=device)


# Hierarchical sampling (section 5.2)
def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    """H:206-250."""
    lead = bins.shape[:-1]
    b2, w2 = bins.reshape(-1, bins.shape[-1]), weights.reshape(-1, weights.shape[-1])
    u = sample_u(b2.shape[0], N_samples, det, pytest, bins.device)
    return ops.sample_pdf(b2, w2.detach(), u).reshape(*lead, N_samples)
