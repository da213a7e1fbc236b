"""Tensor-level wrappers over the C ABI (include/cnerf.h).  Device memory, streams and allocation are
PyTorch-ROCm plumbing; every computation here is a HIP kernel of libcnerf_hip.so.  No fallbacks."""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import Closs, ClossTail, CnerfError, Net, PixelBatch, Ptrs, RayGen, RenderCfg, RenderGrads, RenderOut, Rng, SsWarp

Tensor = torch.Tensor


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-kernel timing with HIP events recorded on the stream the kernels are launched on
# (bench.py's live roofline measurement).  PROFILE = None disables it (zero overhead).
PROFILE = None


class _timed:
    def __init__(self, name, units):
        self.name, self.units = name, units

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e1.record(torch.cuda.current_stream())
            PROFILE.append((self.name, self.units, self.e0, self.e1))
        return False


def _p(t: Optional[Tensor]):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _chk(t: Optional[Tensor], name: str, dtype=torch.float32) -> Optional[Tensor]:
    if t is None:
        return None
    if not t.is_cuda:
        raise CnerfError(f"{name} must live on the GPU: consistentnerf_amd has no CPU path (got {t.device})")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


@dataclass(frozen=True)
class NetSpec:
    """Architecture of one MLP (mirror of struct cnerf_net)."""
    D: int = 8
    W: int = 256
    multires: int = 10
    multires_views: int = 4
    use_viewdirs: bool = True
    output_ch: int = 4
    skip: int = 4

    def c(self) -> Net:
        return Net(self.D, self.W, self.multires, self.multires_views, int(self.use_viewdirs), self.output_ch,
                   self.skip)

    @property
    def raw_ch(self) -> int:
        return 4 if self.use_viewdirs else self.output_ch

    def num_tensors(self) -> int:
        return _lib.load().cnerf_num_tensors(C.byref(self.c()))

    def tensor_shapes(self):
        lib, net, out = _lib.load(), self.c(), []
        r, c = C.c_int64(), C.c_int64()
        for i in range(self.num_tensors()):
            _lib.check(lib.cnerf_tensor_shape(C.byref(net), i, C.byref(r), C.byref(c)), "cnerf_tensor_shape")
            out.append((r.value,) if i & 1 else (r.value, c.value))
        return out


def _ptrs(tensors: Sequence[Optional[Tensor]]) -> Ptrs:
    p = Ptrs()
    for i, t in enumerate(tensors):
        p.p[i] = None if t is None else t.data_ptr()
    return p


def device_info(dev: int = 0):
    lib = _lib.load()
    name = C.create_string_buffer(64)
    cus, lds = C.c_int(), C.c_int()
    rc = lib.cnerf_device_info(dev, name, C.byref(cus), C.byref(lds))
    return rc == 0, name.value.decode(), cus.value, lds.value


# ------------------------------------------------------------------------------------------ weights
def pack_weights(spec: NetSpec, params: Sequence[Tensor], out: Optional[Tensor] = None) -> Tensor:
    lib, net = _lib.load(), spec.c()
    params = [_chk(p, f"param{i}") for i, p in enumerate(params)]
    n = lib.cnerf_packed_floats(C.byref(net))
    if n < 0:
        raise CnerfError(f"unsupported network {spec}")
    if out is None:
        out = torch.empty(n, device=params[0].device, dtype=torch.float32)
    ptrs = _ptrs(params)
    _lib.check(lib.cnerf_pack_weights(C.byref(net), C.byref(ptrs), _p(out), _stream()), "cnerf_pack_weights")
    return out


def pack_weights_pair(spec0: NetSpec, params0: Sequence[Tensor], out0: Optional[Tensor], spec1: NetSpec, params1: Sequence[Tensor],
                      out1: Optional[Tensor]):
    """cnerf_pack_weights_pair: both networks of a render_rays call re-packed by one launch -> (packed0, packed1)."""
    lib, n0, n1 = _lib.load(), spec0.c(), spec1.c()
    params0 = [_chk(p, f"param0[{i}]") for i, p in enumerate(params0)]
    params1 = [_chk(p, f"param1[{i}]") for i, p in enumerate(params1)]
    outs = []
    for net, spec, out, ps in ((n0, spec0, out0, params0), (n1, spec1, out1, params1)):
        n = lib.cnerf_packed_floats(C.byref(net))
        if n < 0:
            raise CnerfError(f"unsupported network {spec}")
        outs.append(out if out is not None else torch.empty(n, device=ps[0].device, dtype=torch.float32))
    p0, p1 = _ptrs(params0), _ptrs(params1)
    _lib.check(lib.cnerf_pack_weights_pair(C.byref(n0), C.byref(p0), _p(outs[0]), C.byref(n1), C.byref(p1), _p(outs[1]),
                                           _stream()), "cnerf_pack_weights_pair")
    return outs[0], outs[1]


# ------------------------------------------------------------------------------------------ sampling
_TVALS = {}


def _t_vals(n: int, device) -> Tensor:
    """torch.linspace(0,1,n) evaluated on the CPU (the reference's constants), cached per device."""
    key = (n, str(device))
    if key not in _TVALS:
        _TVALS[key] = torch.linspace(0., 1., steps=n).to(device)
    return _TVALS[key]


# In-kernel uniform streams (csrc/rng.hpp).  A stream is named by (seed, offset): both come from torch's OWN generator of the device
# — seed = its current seed, offset = its philox offset, which every draw advances by RNG_STRIDE (host-side integers: no launch, no
# sync) — so torch.manual_seed() / get_rng_state() / set_rng_state() govern these streams exactly like they govern torch.rand, and
# torch.rand calls in between keep their own numbers.  Under hipGraph capture the pair lives in device memory instead
# (RngCapture: graph.GraphedStep uploads {seed, offset} before every replay and advances the generator by what a replay consumes).
RNG_STRIDE = 4          # (torch's generator wants offsets in multiples of 4)
IN_KERNEL_RNG = True    # False: render_rays draws t_rand / u with torch.rand and hands the tensors to the kernels (round-3 form)


class RngCapture:
    """Device-resident {seed, base offset} of the recording in progress; `used` = offsets one replay consumes."""
    active = None

    def __init__(self, device):
        self.state = torch.zeros(2, device=device, dtype=torch.int64)
        self.used = 0


@dataclass
class RngStream:
    seed: int
    offset: int
    state: Optional[Tensor] = None     # device int64[2] (capture mode)
    row0: int = 0

    def c(self, offset_add: int = 0, row0: Optional[int] = None) -> Rng:
        return Rng(self.seed & 0xFFFFFFFFFFFFFFFF, (self.offset + offset_add) & 0xFFFFFFFFFFFFFFFF,
                   None if self.state is None else self.state.data_ptr(), self.row0 if row0 is None else row0)


def rng_draw(device, row0: int = 0) -> RngStream:
    """Reserve RNG_STRIDE consecutive stream offsets (one render_rays call: +0 jitter, +1 resampling) from the device's generator."""
    cap = RngCapture.active
    if cap is not None:
        st = RngStream(0, cap.used, cap.state, row0)
        cap.used += RNG_STRIDE
        return st
    if torch.cuda.is_current_stream_capturing():
        raise CnerfError("in-kernel random streams inside a hipGraph recording need graph.GraphedStep (or ops.IN_KERNEL_RNG = False)")
    gen = torch.cuda.default_generators[torch.device(device).index if torch.device(device).index is not None
                                        else torch.cuda.current_device()]
    off = gen.get_offset()
    gen.set_offset(off + RNG_STRIDE)
    return RngStream(gen.initial_seed(), off, None, row0)


def uniform_rng(rng: RngStream, rows: int, cols: int, device, offset_add: int = 0) -> Tensor:
    out = torch.empty(rows, cols, device=device, dtype=torch.float32)
    r = rng.c(offset_add)
    _lib.check(_lib.load().cnerf_uniform_rng(C.byref(r), rows, cols, _p(out), _stream()), "cnerf_uniform_rng")
    return out


def coarse_z(rays: Tensor, Nc: int, t_rand: Optional[Tensor], lindisp: bool, rng: Optional[RngStream] = None) -> Tensor:
    rays = _chk(rays, "rays")
    B = rays.shape[0]
    t_rand = _chk(t_rand, "t_rand")
    z = torch.empty(B, Nc, device=rays.device, dtype=torch.float32)
    if rng is not None:        # jitter generated in the kernel: stream offset + 0
        r = rng.c(0)
        _lib.check(_lib.load().cnerf_coarse_z_rng(_p(rays), rays.shape[1], B, Nc, _p(_t_vals(Nc, rays.device)), C.byref(r),
                                                  int(lindisp), _p(z), _stream()), "cnerf_coarse_z_rng")
        return z
    _lib.check(_lib.load().cnerf_coarse_z(_p(rays), rays.shape[1], B, Nc, _p(_t_vals(Nc, rays.device)), _p(t_rand),
                                          int(lindisp), _p(z), _stream()), "cnerf_coarse_z")
    return z


def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor, want_inds: bool = False):
    bins, weights, u = _chk(bins, "bins"), _chk(weights, "weights"), _chk(u, "u")
    B, Nb = bins.shape
    Nf = u.shape[-1]
    stride = 0 if u.dim() == 1 or u.shape[0] == 1 else Nf
    samples = torch.empty(B, Nf, device=bins.device, dtype=torch.float32)
    inds = torch.empty(B, Nf, device=bins.device, dtype=torch.int64) if want_inds else None
    _lib.check(_lib.load().cnerf_sample_pdf(_p(bins), _p(weights), _p(u), stride, B, Nb, Nf, _p(samples), _p(inds),
                                            _stream()), "cnerf_sample_pdf")
    return (samples, inds) if want_inds else samples


def resample(z: Tensor, weights: Tensor, u: Optional[Tensor], want_samples: bool = False, rng: Optional[RngStream] = None,
             Nf: Optional[int] = None):
    z, weights, u = _chk(z, "z"), _chk(weights, "weights"), _chk(u, "u")
    B, Nc = z.shape
    if rng is not None:        # u generated in the kernel: stream offset + 1
        z_fine = torch.empty(B, Nc + Nf, device=z.device, dtype=torch.float32)
        z_std = torch.empty(B, device=z.device, dtype=torch.float32)
        samples = torch.empty(B, Nf, device=z.device, dtype=torch.float32) if want_samples else None
        inds = torch.empty(B, Nf, device=z.device, dtype=torch.int64) if want_samples else None
        r = rng.c(1)
        _lib.check(_lib.load().cnerf_resample_rng(_p(z), _p(weights), C.byref(r), B, Nc, Nf, _p(z_fine), _p(z_std), _p(samples),
                                                  _p(inds), _stream()), "cnerf_resample_rng")
        return (z_fine, z_std, samples, inds) if want_samples else (z_fine, z_std)
    Nf = u.shape[-1]
    stride = 0 if u.dim() == 1 or u.shape[0] == 1 else Nf
    z_fine = torch.empty(B, Nc + Nf, device=z.device, dtype=torch.float32)
    z_std = torch.empty(B, device=z.device, dtype=torch.float32)
    samples = torch.empty(B, Nf, device=z.device, dtype=torch.float32) if want_samples else None
    inds = torch.empty(B, Nf, device=z.device, dtype=torch.int64) if want_samples else None
    _lib.check(_lib.load().cnerf_resample(_p(z), _p(weights), _p(u), stride, B, Nc, Nf, _p(z_fine), _p(z_std),
                                          _p(samples), _p(inds), _stream()), "cnerf_resample")
    return (z_fine, z_std, samples, inds) if want_samples else (z_fine, z_std)


# ------------------------------------------------------------------------------------------ MLP
def embed(x: Tensor, L: int) -> Tensor:
    x = _chk(x, "x")
    lead = x.shape[:-1]
    x2 = x.reshape(-1, 3)
    out = torch.empty(x2.shape[0], 3 + 6 * L, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().cnerf_embed(_p(x2), x2.shape[0], L, _p(out), _stream()), "cnerf_embed")
    return out.reshape(*lead, 3 + 6 * L)


def mlp_forward(spec: NetSpec, packed: Tensor, B: int, S: int, *, pts: Optional[Tensor] = None,
                rays: Optional[Tensor] = None, z: Optional[Tensor] = None, dirs: Optional[Tensor] = None,
                want_stash: bool = False, live: Optional[Tensor] = None):
    """`live` (device int32 [1], training + rays form only): the batch is padded to the capacity B and only its first live[0] rays
    are real (cnerf_mlp_fwd_live): the launch is sized for B, tiles past the count leave zero raw outputs."""
    lib, net = _lib.load(), spec.c()
    pts, rays, z, dirs = _chk(pts, "pts"), _chk(rays, "rays"), _chk(z, "z"), _chk(dirs, "dirs")
    dev = packed.device
    raw = torch.empty(B, S, spec.raw_ch, device=dev, dtype=torch.float32)
    stash = None
    if want_stash:
        stash = torch.empty(lib.cnerf_mlp_stash_floats(C.byref(net), B * S), device=dev, dtype=torch.float32)
    rs = rays.shape[1] if rays is not None else 0
    if live is not None:
        if not want_stash or rays is None or pts is not None or dirs is not None or live.dtype != torch.int32 or not live.is_cuda:
            raise CnerfError("mlp_forward(live=...) is the training forward on ray rows (no explicit points / directions); "
                             "live must be a device int32 tensor")
        with _timed("mlp_fwd_train", B * S):
            _lib.check(lib.cnerf_mlp_fwd_live(C.byref(net), _p(packed), _p(rays), rs, _p(z), B, S, _p(raw), _p(stash), _p(live),
                                              _stream()), "cnerf_mlp_fwd_live")
        return raw, stash
    with _timed("mlp_fwd_train" if want_stash else "mlp_fwd", B * S):
        _lib.check(lib.cnerf_mlp_fwd(C.byref(net), _p(packed), _p(pts), _p(rays), rs, _p(dirs), _p(z), B, S,
                                     _p(raw), _p(stash), _stream()), "cnerf_mlp_fwd")
    return raw, stash


PRECISION_PLANES = {"bf16": 1, "bf16x2": 2, "bf16x3": 3}
# which kernels of a bf16x3 training step run in that arithmetic (ablation switches; all on by default)
import os as _os
DGRAD_BF3 = _os.environ.get("CNERF_BF3_DGRAD", "1") != "0"
WGRAD_BF3 = _os.environ.get("CNERF_BF3_WGRAD", "1") != "0"


def pack_weights_bf(spec: NetSpec, params: Sequence[Tensor], planes: int, out: Optional[Tensor] = None) -> Tensor:
    """cnerf_pack_weights_bf: the bf16-plane panels of the opt-in reduced-precision inference forward (a byte buffer)."""
    lib, net = _lib.load(), spec.c()
    params = [_chk(p, f"param{i}") for i, p in enumerate(params)]
    n = lib.cnerf_packed_bf_bytes(C.byref(net), int(planes))
    if n < 0:
        raise CnerfError(f"reduced-precision inference is not compiled for {spec} with {planes} plane(s)")
    if out is None:
        out = torch.empty(n, device=params[0].device, dtype=torch.uint8)
    ptrs = _ptrs(params)
    _lib.check(lib.cnerf_pack_weights_bf(C.byref(net), C.byref(ptrs), int(planes), _p(out), _stream()), "cnerf_pack_weights_bf")
    return out


def mlp_forward_bf(spec: NetSpec, packed_bf: Tensor, planes: int, B: int, S: int, *, pts: Optional[Tensor] = None,
                   rays: Optional[Tensor] = None, z: Optional[Tensor] = None, dirs: Optional[Tensor] = None) -> Tensor:
    """cnerf_mlp_fwd_bf: inference forward on the bf16 matrix cores (opt-in)."""
    lib, net = _lib.load(), spec.c()
    pts, rays, z, dirs = _chk(pts, "pts"), _chk(rays, "rays"), _chk(z, "z"), _chk(dirs, "dirs")
    raw = torch.empty(B, S, spec.raw_ch, device=packed_bf.device, dtype=torch.float32)
    rs = rays.shape[1] if rays is not None else 0
    with _timed("mlp_fwd_bf%d" % planes, B * S):
        _lib.check(lib.cnerf_mlp_fwd_bf(C.byref(net), _p(packed_bf), int(planes), _p(pts), _p(rays), rs, _p(dirs), _p(z), B, S,
                                        _p(raw), _stream()), "cnerf_mlp_fwd_bf")
    return raw


def mlp_forward_bf_train(spec: NetSpec, packed_bf: Tensor, B: int, S: int, *, pts: Optional[Tensor] = None,
                         rays: Optional[Tensor] = None, z: Optional[Tensor] = None, dirs: Optional[Tensor] = None):
    """cnerf_mlp_fwd_bf_train: the OPT-IN bf16x3 training forward (three bf16 planes per operand, fp32 accumulation) ->
    (raw, stash); the stash is the fp32 kernel's (cnerf_mlp_dgrad / cnerf_mlp_wgrad consume it unchanged)."""
    lib, net = _lib.load(), spec.c()
    pts, rays, z, dirs = _chk(pts, "pts"), _chk(rays, "rays"), _chk(z, "z"), _chk(dirs, "dirs")
    dev = packed_bf.device
    raw = torch.empty(B, S, spec.raw_ch, device=dev, dtype=torch.float32)
    stash = torch.empty(lib.cnerf_mlp_stash_floats(C.byref(net), B * S), device=dev, dtype=torch.float32)
    rs = rays.shape[1] if rays is not None else 0
    with _timed("mlp_fwd_train_bf3", B * S):
        _lib.check(lib.cnerf_mlp_fwd_bf_train(C.byref(net), _p(packed_bf), _p(pts), _p(rays), rs, _p(dirs), _p(z), B, S,
                                              _p(raw), _p(stash), _stream()), "cnerf_mlp_fwd_bf_train")
    return raw, stash


def mlp_forward_embedded(spec: NetSpec, packed: Tensor, x: Tensor, want_stash: bool = False):
    """NeRF.forward on pre-embedded inputs x[M, in_ch + in_ch_views]."""
    lib, net = _lib.load(), spec.c()
    x = _chk(x, "x")
    M = x.shape[0]
    raw = torch.empty(M, spec.raw_ch, device=x.device, dtype=torch.float32)
    stash = None
    if want_stash:
        stash = torch.empty(lib.cnerf_mlp_stash_floats(C.byref(net), M), device=x.device, dtype=torch.float32)
    with _timed("mlp_fwd_train" if want_stash else "mlp_fwd", M):
        _lib.check(lib.cnerf_mlp_fwd_embedded(C.byref(net), _p(packed), _p(x), M, _p(raw), _p(stash), _stream()),
                   "cnerf_mlp_fwd_embedded")
    return raw, stash


def mlp_backward(spec: NetSpec, packed: Tensor, d_raw: Tensor, B: int, S: int, stash: Tensor,
                 grads: Optional[List[Tensor]] = None, accumulate: bool = False, packed_bf: Optional[Tensor] = None,
                 live: Optional[Tensor] = None) -> List[Tensor]:
    """packed_bf (the three-plane buffer of pack_weights_bf): the dgrad runs in the opt-in bf16x3 arithmetic.
    live: the forward ran through mlp_forward(live=...) — the backward stops at the same device-side row count (exact fp32)."""
    lib, net = _lib.load(), spec.c()
    d_raw = _chk(d_raw, "d_raw")
    dev = packed.device
    if grads is None:
        grads = [torch.empty(s, device=dev, dtype=torch.float32) for s in spec.tensor_shapes()]
        accumulate = False
    ws = torch.empty(lib.cnerf_mlp_bwd_ws_floats(C.byref(net), B * S), device=dev, dtype=torch.float32)
    ptrs = _ptrs(grads)
    if live is not None:
        if packed_bf is not None:
            raise CnerfError("mlp_backward(live=...) is an exact-fp32 path")
        with _timed("mlp_bwd_live", B * S):
            _lib.check(lib.cnerf_mlp_bwd_live(C.byref(net), _p(packed), _p(d_raw), B, S, _p(stash), _p(ws), C.byref(ptrs),
                                              int(accumulate), _p(live), _stream()), "cnerf_mlp_bwd_live")
        return grads
    if packed_bf is not None and DGRAD_BF3:
        with _timed("mlp_dgrad_bf3", B * S):
            _lib.check(lib.cnerf_mlp_dgrad_bf(C.byref(net), _p(packed_bf), _p(d_raw), B, S, _p(stash), _p(ws), _stream()),
                       "cnerf_mlp_dgrad_bf")
    else:
      with _timed("mlp_dgrad", B * S):
        _lib.check(lib.cnerf_mlp_dgrad(C.byref(net), _p(packed), _p(d_raw), B, S, _p(stash), _p(ws), _stream()),
                   "cnerf_mlp_dgrad")
    bf_w = packed_bf is not None and WGRAD_BF3
    with _timed("mlp_wgrad_bf3" if bf_w else "mlp_wgrad", B * S):
        _lib.check((lib.cnerf_mlp_wgrad_bf if bf_w else lib.cnerf_mlp_wgrad)(C.byref(net), B, S, _p(stash), _p(ws), C.byref(ptrs),
                                                                             int(accumulate), _stream()), "cnerf_mlp_wgrad")
    return grads


def mlp_backward_pair(spec0: NetSpec, packed0: Tensor, d_raw0: Tensor, B0: int, S0: int, stash0: Tensor, grads0: List[Tensor],
                      spec1: NetSpec, packed1: Tensor, d_raw1: Tensor, B1: int, S1: int, stash1: Tensor, grads1: List[Tensor],
                      accumulate: bool = False, packed_bf0: Optional[Tensor] = None, packed_bf1: Optional[Tensor] = None,
                      live: Optional[Tensor] = None, first0: int = 0, first1: int = 0):
    """cnerf_mlp_bwd_pair: the backward of two independent networks (coarse / fine) as one dgrad grid, one wgrad grid and
    one reduction; gradients are written (or accumulated) into grads0 / grads1.  packed_bf0 AND packed_bf1: the dgrad grid runs
    in the opt-in bf16x3 arithmetic.  live: both levels belong to one ray batch whose live row count sits on the device
    (mlp_forward(live=...)): cnerf_mlp_bwd_pair_live (exact fp32); first0 / first1: that level's first rays carry zero seeds and are
    left out of its backward."""
    lib = _lib.load()
    n0, n1 = spec0.c(), spec1.c()
    d_raw0, d_raw1 = _chk(d_raw0, "d_raw0"), _chk(d_raw1, "d_raw1")
    dev = packed0.device
    ws0 = torch.empty(lib.cnerf_mlp_bwd_ws_floats(C.byref(n0), B0 * S0), device=dev, dtype=torch.float32)
    ws1 = torch.empty(lib.cnerf_mlp_bwd_ws_floats(C.byref(n1), B1 * S1), device=dev, dtype=torch.float32)
    p0, p1 = _ptrs(grads0), _ptrs(grads1)
    if live is not None:
        if packed_bf0 is not None or packed_bf1 is not None:
            raise CnerfError("mlp_backward_pair(live=...) is an exact-fp32 path")
        # (the halves of cnerf_mlp_bwd_pair_live, launched separately so that bench.py's HIP events bracket each kernel; `units` is
        #  the launch CAPACITY — the live point count is on the device — the bench rescales by the step's live rows)
        with _timed("mlp_dgrad", B0 * S0 + B1 * S1):
            _lib.check(lib.cnerf_mlp_dgrad_pair_live(C.byref(n0), _p(packed0), _p(d_raw0), B0, S0, _p(stash0), _p(ws0), C.byref(n1),
                                                     _p(packed1), _p(d_raw1), B1, S1, _p(stash1), _p(ws1), _p(live), int(first0),
                                                     int(first1), _stream()), "cnerf_mlp_dgrad_pair_live")
        with _timed("mlp_wgrad", B0 * S0 + B1 * S1):
            _lib.check(lib.cnerf_mlp_wgrad_pair_live(C.byref(n0), B0, S0, _p(stash0), _p(ws0), C.byref(p0), C.byref(n1), B1, S1,
                                                     _p(stash1), _p(ws1), C.byref(p1), int(accumulate), _p(live), int(first0),
                                                     int(first1), _stream()), "cnerf_mlp_wgrad_pair_live")
        return
    if packed_bf0 is not None and packed_bf1 is not None and DGRAD_BF3:
        with _timed("mlp_dgrad_bf3", B0 * S0 + B1 * S1):
            _lib.check(lib.cnerf_mlp_dgrad_bf_pair(C.byref(n0), _p(packed_bf0), _p(d_raw0), B0, S0, _p(stash0), _p(ws0),
                                                   C.byref(n1), _p(packed_bf1), _p(d_raw1), B1, S1, _p(stash1), _p(ws1), _stream()),
                       "cnerf_mlp_dgrad_bf_pair")
    else:
      with _timed("mlp_dgrad", B0 * S0 + B1 * S1):
        _lib.check(lib.cnerf_mlp_dgrad_pair(C.byref(n0), _p(packed0), _p(d_raw0), B0, S0, _p(stash0), _p(ws0),
                                            C.byref(n1), _p(packed1), _p(d_raw1), B1, S1, _p(stash1), _p(ws1), _stream()),
                   "cnerf_mlp_dgrad_pair")
    bf_w = packed_bf0 is not None and packed_bf1 is not None and WGRAD_BF3
    with _timed("mlp_wgrad_bf3" if bf_w else "mlp_wgrad", B0 * S0 + B1 * S1):
        _lib.check((lib.cnerf_mlp_wgrad_bf_pair if bf_w else lib.cnerf_mlp_wgrad_pair)(
            C.byref(n0), B0, S0, _p(stash0), _p(ws0), C.byref(p0), C.byref(n1), B1, S1, _p(stash1), _p(ws1), C.byref(p1),
            int(accumulate), _stream()), "cnerf_mlp_wgrad_pair")


# ------------------------------------------------------------------------------------------ render_rays as one call
class RenderState:
    """What cnerf_render_bwd needs from the forward call it follows: the workspace (z, raw, weights, stashes) and the
    call's arguments."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def render_forward(spec_c: NetSpec, packed_c: Tensor, spec_f: Optional[NetSpec], packed_f: Optional[Tensor], rays: Tensor,
                   Nc: int, Nf: int, t_rand: Optional[Tensor] = None, u: Optional[Tensor] = None,
                   noise0: Optional[Tensor] = None, noise1: Optional[Tensor] = None, lindisp: bool = False,
                   white_bkgd: bool = False, train: bool = False, retraw: bool = False):
    """cnerf_render_fwd: render_rays (R:311-421 / V:441-551) of one ray batch in one C call.  Returns (dict with the
    reference's keys + depth maps, RenderState for render_backward)."""
    lib = _lib.load()
    rays, t_rand, u = _chk(rays, "rays"), _chk(t_rand, "t_rand"), _chk(u, "u")
    noise0, noise1 = _chk(noise0, "noise0"), _chk(noise1, "noise1")
    B, dev = rays.shape[0], rays.device
    cfg = RenderCfg(int(Nc), int(Nf), int(lindisp), int(white_bkgd), int(rays.shape[1]), int(train))
    nc, nf = spec_c.c(), (spec_f.c() if spec_f is not None else None)
    nfp = C.byref(nf) if nf is not None else None
    n = lib.cnerf_render_ws_floats(C.byref(nc), nfp, C.byref(cfg), B)
    if n < 0:
        raise CnerfError("cnerf_render_ws_floats: inconsistent arguments")
    ws = torch.empty(n, device=dev, dtype=torch.float32)
    S = Nc + Nf
    ch = (spec_f if (spec_f is not None and Nf > 0) else spec_c).raw_ch
    o = {k: torch.empty(B, 3, device=dev) if k.startswith("rgb") else torch.empty(B, device=dev)
         for k in ("rgb_map", "disp_map", "acc_map", "depth_map")}
    if Nf > 0:
        o.update({k: torch.empty(B, 3, device=dev) if k.startswith("rgb") else torch.empty(B, device=dev)
                  for k in ("rgb0", "disp0", "acc0", "depth0", "z_std")})
    if retraw:
        o["raw"] = torch.empty(B, S, ch, device=dev)
    out = RenderOut(**{k: v.data_ptr() for k, v in o.items()})
    stride = 0 if (u is None or u.dim() == 1 or u.shape[0] == 1) else Nf
    _lib.check(lib.cnerf_render_fwd(C.byref(nc), _p(packed_c), nfp, _p(packed_f), _p(rays), B, C.byref(cfg),
                                    _p(_t_vals(Nc, dev)), _p(t_rand), _p(u), stride, _p(noise0), _p(noise1), C.byref(out),
                                    _p(ws), _stream()), "cnerf_render_fwd")
    st = RenderState(spec_c=spec_c, packed_c=packed_c, spec_f=spec_f, packed_f=packed_f, rays=rays, B=B, cfg=cfg,
                     noise0=noise0, noise1=noise1, ws=ws)
    return o, st


def render_forward_cam(spec_c: NetSpec, packed_c: Tensor, spec_f: Optional[NetSpec], packed_f: Optional[Tensor], H: int, W: int,
                       K, c2w, near: float, far: float, use_viewdirs: bool, ndc: bool, ndc_coef, first: int, B: int, Nc: int,
                       Nf: int, t_rand: Optional[Tensor] = None, u: Optional[Tensor] = None, noise0: Optional[Tensor] = None,
                       noise1: Optional[Tensor] = None, lindisp: bool = False, white_bkgd: bool = False, retraw: bool = False):
    """cnerf_render_fwd_cam: render_rays (inference) of the image chunk [first, first + B) of a camera whose rays are
    generated inside the kernels — no [H*W, 11] ray tensor.  Returns the dict of cnerf_render_fwd."""
    lib = _lib.load()
    t_rand, u, noise0, noise1 = _chk(t_rand, "t_rand"), _chk(u, "u"), _chk(noise0, "noise0"), _chk(noise1, "noise1")
    dev = packed_c.device
    cfg = RenderCfg(int(Nc), int(Nf), int(lindisp), int(white_bkgd), 11, 0)
    nc, nf = spec_c.c(), (spec_f.c() if spec_f is not None else None)
    nfp = C.byref(nf) if nf is not None else None
    n = lib.cnerf_render_ws_floats(C.byref(nc), nfp, C.byref(cfg), B)
    if n < 0:
        raise CnerfError("cnerf_render_ws_floats: inconsistent arguments")
    ws = torch.empty(n, device=dev, dtype=torch.float32)
    if isinstance(c2w, torch.Tensor):
        c2w = c2w.detach().cpu().numpy()
    cam = RayGen(int(H), int(W), float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]), _f4(c2w), float(near),
                 float(far), int(use_viewdirs), int(ndc), float(ndc_coef[0]), float(ndc_coef[1]), int(first))
    S = Nc + Nf
    ch = (spec_f if (spec_f is not None and Nf > 0) else spec_c).raw_ch
    o = {k: torch.empty(B, 3, device=dev) if k.startswith("rgb") else torch.empty(B, device=dev)
         for k in ("rgb_map", "disp_map", "acc_map", "depth_map")}
    if Nf > 0:
        o.update({k: torch.empty(B, 3, device=dev) if k.startswith("rgb") else torch.empty(B, device=dev)
                  for k in ("rgb0", "disp0", "acc0", "depth0", "z_std")})
    if retraw:
        o["raw"] = torch.empty(B, S, ch, device=dev)
    out = RenderOut(**{k: v.data_ptr() for k, v in o.items()})
    stride = 0 if (u is None or u.dim() == 1 or u.shape[0] == 1) else Nf
    with _timed("render_fwd_cam", B * (Nc + (S if Nf > 0 else 0))):
        _lib.check(lib.cnerf_render_fwd_cam(C.byref(nc), _p(packed_c), nfp, _p(packed_f), C.byref(cam), B, C.byref(cfg),
                                            _p(_t_vals(Nc, dev)), _p(t_rand), _p(u), stride, _p(noise0), _p(noise1),
                                            C.byref(out), _p(ws), _stream()), "cnerf_render_fwd_cam")
    return o


def render_backward(st: RenderState, grads_in: dict, grads_c: List[Tensor], grads_f: Optional[List[Tensor]],
                    accumulate: bool = False):
    """cnerf_render_bwd: upstream gradients of the maps (dict with any of rgb_map, disp_map, acc_map, depth_map, rgb0,
    disp0, acc0, depth0) -> parameter gradients of the coarse / fine network, written into the given tensors."""
    lib = _lib.load()
    keep = {k: _chk(v, k) for k, v in grads_in.items() if v is not None}
    g = RenderGrads(**{"g_" + k: v.data_ptr() for k, v in keep.items()})
    nc, nf = st.spec_c.c(), (st.spec_f.c() if st.spec_f is not None else None)
    pc, pf = _ptrs(grads_c), (_ptrs(grads_f) if grads_f is not None else None)
    _lib.check(lib.cnerf_render_bwd(C.byref(nc), _p(st.packed_c), C.byref(nf) if nf is not None else None,
                                    _p(st.packed_f), _p(st.rays), st.B, C.byref(st.cfg), _p(st.noise0), _p(st.noise1),
                                    C.byref(g), _p(st.ws), C.byref(pc), C.byref(pf) if pf is not None else None,
                                    int(accumulate), _stream()), "cnerf_render_bwd")


# ------------------------------------------------------------------------------------------ compositing
def composite_forward(raw: Tensor, z: Tensor, rays: Tensor, noise: Optional[Tensor], white_bkgd: bool):
    raw, z, rays, noise = _chk(raw, "raw"), _chk(z, "z"), _chk(rays, "rays"), _chk(noise, "noise")
    B, S = z.shape
    dev = raw.device
    rgb = torch.empty(B, 3, device=dev)
    disp, acc, depth = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)
    weights = torch.empty(B, S, device=dev)
    _lib.check(_lib.load().cnerf_composite_fwd(_p(raw), raw.shape[-1], _p(z), _p(rays), rays.shape[1], _p(noise), B, S,
                                               int(white_bkgd), _p(rgb), _p(disp), _p(acc), _p(depth), _p(weights),
                                               _stream()), "cnerf_composite_fwd")
    return rgb, disp, acc, weights, depth


_MSE_COUNTERS = {}


def _mse_counter(device) -> Tensor:
    """The ticket counters of cnerf_composite_fwd_mse: one zeroed block per (device, stream) — the kernel leaves it zero.  Under a
    hipGraph recording: the block the recorder allocated BEFORE the recording and owns (prepare_capture: one per GraphedStep, so
    two recorded steps replayed concurrently on different streams never share tickets; memory allocated inside a recording
    belongs to that graph's pool and must not be cached); without a recorder, a throw-away zeroed block owned by the graph."""
    if torch.cuda.is_current_stream_capturing():
        t = _MSE_COUNTERS.get((str(device), "graph"))
        return t if t is not None else torch.zeros(_mse_counter_words(), device=device, dtype=torch.int32)
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    if key not in _MSE_COUNTERS:
        _MSE_COUNTERS[key] = torch.zeros(_mse_counter_words(), device=device, dtype=torch.int32)
    return _MSE_COUNTERS[key]


def _mse_counter_words() -> int:
    return int(_lib.load().cnerf_composite_mse_counter_words())


def composite_mse_max_rays() -> int:
    return int(_lib.load().cnerf_composite_mse_max_rays())


def prepare_capture(device) -> Tensor:
    """Allocate, outside the recording, what the kernels of a recorded step keep across replays: a ticket-counter block of the
    recording's OWN (returned: the recorder keeps it alive as long as its graph) — installed as the block launches recorded on this
    device use until end_capture()."""
    t = torch.zeros(_mse_counter_words(), device=device, dtype=torch.int32)
    _MSE_COUNTERS[(str(torch.device(device)), "graph")] = t
    return t


def end_capture(device):
    _MSE_COUNTERS.pop((str(torch.device(device)), "graph"), None)


def composite_forward_mse(raw: Tensor, z: Tensor, rays: Tensor, noise: Optional[Tensor], white_bkgd: bool, target: Tensor,
                          loss_add: Optional[Tensor] = None):
    """cnerf_composite_fwd_mse -> (rgb, disp, acc, weights, depth, loss[1]): raw2outputs + img2mse(rgb_map, target) (+ loss_add)."""
    raw, z, rays, noise = _chk(raw, "raw"), _chk(z, "z"), _chk(rays, "rays"), _chk(noise, "noise")
    target, loss_add = _chk(target, "target"), _chk(loss_add, "loss_add")
    B, S = z.shape
    if B == 0 or tuple(target.shape) != (B, 3):
        raise CnerfError(f"composite_forward_mse: target must be [{B}, 3] with B > 0, got {tuple(target.shape)}")
    dev = raw.device
    rgb = torch.empty(B, 3, device=dev)
    disp, acc, depth = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)
    weights = torch.empty(B, S, device=dev)
    loss = torch.empty(1, device=dev)
    lib = _lib.load()
    ws = torch.empty(lib.cnerf_composite_mse_ws_floats(B) // 2, device=dev, dtype=torch.float64)
    _lib.check(lib.cnerf_composite_fwd_mse(_p(raw), raw.shape[-1], _p(z), _p(rays), rays.shape[1], _p(noise), B, S,
                                           int(white_bkgd), _p(target), _p(loss_add), _p(rgb), _p(disp), _p(acc), _p(depth),
                                           _p(weights), _p(loss), _p(ws), _p(_mse_counter(dev)), _stream()),
               "cnerf_composite_fwd_mse")
    return rgb, disp, acc, weights, depth, loss


def composite_backward_mse(raw, z, rays, noise, white_bkgd, rgb, target, g_loss) -> Tensor:
    """cnerf_composite_bwd_mse: d_raw of mean((rgb_map - target)^2) * g_loss (a device scalar, or None = 1)."""
    raw, z, rays, noise = _chk(raw, "raw"), _chk(z, "z"), _chk(rays, "rays"), _chk(noise, "noise")
    rgb, target, g_loss = _chk(rgb, "rgb"), _chk(target, "target"), _chk(g_loss, "g_loss")
    B, S = z.shape
    d_raw = torch.empty_like(raw)
    _lib.check(_lib.load().cnerf_composite_bwd_mse(_p(raw), raw.shape[-1], _p(z), _p(rays), rays.shape[1], _p(noise), B, S,
                                                   int(white_bkgd), _p(rgb), _p(target), _p(g_loss), _p(d_raw), _stream()),
               "cnerf_composite_bwd_mse")
    return d_raw


# ---- the ConsistentNeRF losses folded into compositing (cnerf_composite_fwd_closs / cnerf_closs_finish / cnerf_composite_bwd_closs)
@dataclass
class ClossSpec:
    """The loss of one ConsistentNeRF training batch (V:1645-1865): masked rgb + depth terms on every level (mask [B] 0/1 floats or
    None; prior [B] or None = no depth term; depths compared after / far), the monocular patch term on the first P * n rays (mono
    [P * n] or None), weights of the three kinds of term in the step's loss, optional GLOBAL (n1, n0) for a sharded batch."""
    target: Tensor
    mask: Optional[Tensor] = None
    prior: Optional[Tensor] = None
    far: float = 1.0
    coef: float = 0.2
    rgb_w: float = 1.0
    depth_w: float = 1.0
    patch_w: float = 0.001
    mono: Optional[Tensor] = None
    P: int = 0
    n: int = 256
    counts: Optional[Tensor] = None
    ss_coins: Optional[tuple] = None      # (rgb, depth, rgb0, depth0) draws of VT:941-969: the in-loop consistency step's primary terms
    seg_row: int = 0                      # > 0 (with ss_coins): the batch is [primary rays | warped rays + padding] cut here (ss_batch)
    counts3: Optional[Tensor] = None      # (with seg_row) GLOBAL (selected, primary, warped) ray counts of a batch sharded over ranks

    def c(self) -> Closs:
        return Closs(self.target.data_ptr(), None if self.mask is None else self.mask.data_ptr(),
                     None if self.prior is None else self.prior.data_ptr(), float(self.far), int(self.seg_row))

    def checked(self, B: int) -> "ClossSpec":
        t = _chk(self.target.reshape(-1, 3), "target")
        if t.shape[0] != B or B == 0:
            raise CnerfError(f"closs: target must be [{B}, 3] with B > 0, got {tuple(self.target.shape)}")
        m = None if self.mask is None else _chk(self.mask.reshape(-1).to(torch.float32), "mask")
        pr = None if self.prior is None else _chk(self.prior.reshape(-1), "prior")
        mono = None if (self.mono is None or self.P <= 0) else _chk(self.mono.reshape(-1), "mono")
        for x, nme in ((m, "mask"), (pr, "prior")):
            if x is not None and x.numel() != B:
                raise CnerfError(f"closs: {nme} must have {B} elements, got {x.numel()}")
        P = int(self.P) if mono is not None else 0
        if P > 0 and (mono.numel() < P * self.n or P * self.n > B or P > 8):
            raise CnerfError(f"closs: the patch term needs P <= 8 patches of n rays inside the batch (P={P}, n={self.n}, B={B})")
        coins = None if self.ss_coins is None else tuple(int(bool(c)) for c in self.ss_coins)
        if coins is not None and (len(coins) != 4 or P > 0 or self.counts is not None or m is None):
            raise CnerfError("closs: ss_coins takes 4 draws, a selection mask, no patch term and no global counts")
        seg = int(self.seg_row)
        if seg and (coins is None or seg % 8 != 0 or not 0 < seg < B):
            raise CnerfError(f"closs: seg_row {seg} needs ss_coins, a multiple of 8 and 0 < seg_row < B = {B}")
        return ClossSpec(t, m, pr, float(self.far), float(self.coef), float(self.rgb_w), float(self.depth_w), float(self.patch_w),
                         mono, P, int(self.n), _chk(self.counts, "counts"), coins, seg, _chk(self.counts3, "counts3") if seg else None)


def composite_forward_closs(raw: Tensor, z: Tensor, rays: Tensor, noise: Optional[Tensor], white_bkgd: bool, L: ClossSpec):
    """-> (rgb, disp, acc, weights, depth, ws): raw2outputs + the level's five masked-loss partial sums per workgroup in `ws`."""
    raw, z, rays, noise = _chk(raw, "raw"), _chk(z, "z"), _chk(rays, "rays"), _chk(noise, "noise")
    B, S = z.shape
    dev = raw.device
    rgb = torch.empty(B, 3, device=dev)
    disp, acc, depth = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)
    weights = torch.empty(B, S, device=dev)
    lib = _lib.load()
    ws = torch.empty(lib.cnerf_closs_ws_floats(B) // 2, device=dev, dtype=torch.float64)
    c = L.c()
    _lib.check(lib.cnerf_composite_fwd_closs(_p(raw), raw.shape[-1], _p(z), _p(rays), rays.shape[1], _p(noise), B, S,
                                             int(white_bkgd), C.byref(c), _p(rgb), _p(disp), _p(acc), _p(depth), _p(weights),
                                             _p(ws), _stream()), "cnerf_composite_fwd_closs")
    return rgb, disp, acc, weights, depth, ws


def closs_finish(L: ClossSpec, B: int, ws_last: Tensor, ws_coarse: Optional[Tensor], depth_last: Optional[Tensor],
                 depth_coarse: Optional[Tensor], want_grad: bool = True):
    """cnerf_closs_finish -> (terms[8], stats[8], patch_d[levels, P * n] | None); with L.seg_row (the one-render in-loop consistency
    step, cnerf_closs_finish_ss2): terms[12], stats[16] = per level [2 segments][4]."""
    dev = ws_last.device
    terms, stats = torch.empty(12 if L.seg_row else 8, device=dev), torch.empty(16 if L.seg_row else 8, device=dev)
    levels = 2 if ws_coarse is not None else 1
    patch_d = torch.empty(levels, L.P * L.n, device=dev) if (L.P > 0 and want_grad) else None
    a = lambda x: None if x is None else x.data_ptr()  # noqa: E731
    t = ClossTail(a(ws_last), a(ws_coarse), int(B), a(L.counts), L.coef, L.far, L.rgb_w, L.depth_w, L.patch_w,
                  int(L.prior is not None), a(depth_last) if L.P > 0 else None,
                  a(depth_coarse) if (L.P > 0 and levels == 2) else None, a(L.mono) if L.P > 0 else None, L.P, L.n)
    if L.ss_coins is not None and L.seg_row:
        coins = (C.c_int32 * 4)(*L.ss_coins)
        _lib.check(_lib.load().cnerf_closs_finish_ss2(C.byref(t), coins, int(L.seg_row), _p(L.counts3), _p(terms), _p(stats), _stream()),
                   "cnerf_closs_finish_ss2")
    elif L.ss_coins is not None:
        coins = (C.c_int32 * 4)(*L.ss_coins)
        _lib.check(_lib.load().cnerf_closs_finish_ss(C.byref(t), coins, _p(terms), _p(stats), _stream()), "cnerf_closs_finish_ss")
    else:
        _lib.check(_lib.load().cnerf_closs_finish(C.byref(t), _p(terms), _p(stats), _p(patch_d), _stream()), "cnerf_closs_finish")
    return terms, stats, patch_d


def composite_backward_closs(raw, z, rays, noise, white_bkgd, L: ClossSpec, rgb, depth, stats4, g_loss, patch_d) -> Tensor:
    raw, z, rays, noise = _chk(raw, "raw"), _chk(z, "z"), _chk(rays, "rays"), _chk(noise, "noise")
    B, S = z.shape
    d_raw = torch.empty_like(raw)
    c = L.c()
    _lib.check(_lib.load().cnerf_composite_bwd_closs(_p(raw), raw.shape[-1], _p(z), _p(rays), rays.shape[1], _p(noise), B, S,
                                                     int(white_bkgd), C.byref(c), _p(rgb), _p(depth), _p(stats4), _p(g_loss),
                                                     L.rgb_w, L.depth_w, L.patch_w, _p(patch_d),
                                                     L.P * L.n if patch_d is not None else 0, _p(d_raw), _stream()),
               "cnerf_composite_bwd_closs")
    return d_raw


def composite_backward(raw, z, rays, noise, white_bkgd, g_rgb, g_disp, g_acc, g_depth) -> Tensor:
    raw, z, rays, noise = _chk(raw, "raw"), _chk(z, "z"), _chk(rays, "rays"), _chk(noise, "noise")
    g_rgb, g_disp, g_acc, g_depth = (_chk(g, "grad") for g in (g_rgb, g_disp, g_acc, g_depth))
    B, S = z.shape
    d_raw = torch.empty_like(raw)
    _lib.check(_lib.load().cnerf_composite_bwd(_p(raw), raw.shape[-1], _p(z), _p(rays), rays.shape[1], _p(noise), B, S,
                                               int(white_bkgd), _p(g_rgb), _p(g_disp), _p(g_acc), _p(g_depth),
                                               _p(d_raw), _stream()), "cnerf_composite_bwd")
    return d_raw


# ------------------------------------------------------------------------------------------ rays
def _f4(a) -> "C.Array":
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32)[:3, :4]).reshape(-1)
    return (C.c_float * 12)(*a.tolist())


def gen_rays(H: int, W: int, K, c2w, near: float, far: float, use_viewdirs: bool, ndc: bool, device,
             ndc_coef=(0.0, 0.0)) -> Tensor:
    rays = torch.empty(H * W, 11 if use_viewdirs else 8, device=device, dtype=torch.float32)
    if isinstance(c2w, torch.Tensor):
        c2w = c2w.detach().cpu().numpy()
    _lib.check(_lib.load().cnerf_gen_rays(H, W, float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]),
                                          _f4(c2w), float(near), float(far), int(use_viewdirs), int(ndc),
                                          float(ndc_coef[0]), float(ndc_coef[1]), _p(rays), _stream()),
               "cnerf_gen_rays")
    return rays


def pack_rays(rays_o: Tensor, rays_d: Tensor, near: float, far: float, use_viewdirs: bool, ndc: bool,
              ndc_coef=(0.0, 0.0)) -> Tensor:
    rays_o, rays_d = _chk(rays_o.reshape(-1, 3), "rays_o"), _chk(rays_d.reshape(-1, 3), "rays_d")
    B = rays_o.shape[0]
    rays = torch.empty(B, 11 if use_viewdirs else 8, device=rays_o.device, dtype=torch.float32)
    _lib.check(_lib.load().cnerf_pack_rays(_p(rays_o), _p(rays_d), B, float(near), float(far), int(use_viewdirs),
                                           int(ndc), float(ndc_coef[0]), float(ndc_coef[1]), _p(rays), _stream()),
               "cnerf_pack_rays")
    return rays


def sample_pixels(H: int, W: int, K, c2w, near: float, far: float, use_viewdirs: bool, ndc: bool, ndc_coef, crop, patch_starts,
                  patch_size: int, n_rand: int, select_inds: Optional[Tensor], rng: Optional[RngStream], image: Tensor,
                  extras: Sequence[Tensor] = (), want_rows: bool = True, want_od: bool = True, want_coords: bool = True):
    """cnerf_sample_pixels: the training batch of one image in one launch -> (rays [B, 8|11] | None, rays_od [2, B, 3] | None,
    target [B, 3], extras_out [n_extras, B], coords [B, 2] | None).  crop = (r0, c0, h, w) of the grid the random pixels come
    from; patch_starts [P, 2] host integers (or None); select_inds: device int64 [n_rand] or None (device draw keyed by rng)."""
    image = _chk(image, "image")
    if image.dim() != 3 or image.shape[0] != H or image.shape[1] != W or image.shape[2] < 3:
        raise CnerfError(f"sample_pixels: image must be [{H}, {W}, >=3], got {tuple(image.shape)}")
    dev = image.device
    ex = [_chk(e, "extra") for e in extras]
    for e in ex:
        if tuple(e.shape) != (H, W):
            raise CnerfError(f"sample_pixels: per-pixel maps must be [{H}, {W}], got {tuple(e.shape)}")
    starts = np.zeros((0, 2), np.int64) if patch_starts is None else np.asarray(patch_starts, np.int64).reshape(-1, 2)
    P = starts.shape[0]
    cfg = PixelBatch()
    cfg.H, cfg.W, cfg.fx, cfg.fy, cfg.cx, cfg.cy = H, W, float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    if isinstance(c2w, torch.Tensor):
        c2w = c2w.detach().cpu().numpy()
    cfg.c2w = _f4(c2w)
    cfg.near, cfg.far, cfg.use_viewdirs, cfg.ndc = float(near), float(far), int(use_viewdirs), int(ndc)
    cfg.ndc_ax, cfg.ndc_ay = float(ndc_coef[0]), float(ndc_coef[1])
    cfg.crop_r0, cfg.crop_c0, cfg.crop_h, cfg.crop_w = (int(v) for v in crop)
    cfg.n_patches, cfg.patch_size = P, int(patch_size)
    for q in range(P):
        cfg.patch_start[q][0], cfg.patch_start[q][1] = int(starts[q, 0]), int(starts[q, 1])
    cfg.n_rand, cfg.image_ch, cfg.n_extras = int(n_rand), int(image.shape[2]), len(ex)
    B = P * patch_size * patch_size + int(n_rand)
    if select_inds is not None:
        select_inds = _chk(select_inds, "select_inds", dtype=torch.int64)
        if select_inds.numel() != n_rand:
            raise CnerfError(f"sample_pixels: select_inds must have {n_rand} elements")
    rows = torch.empty(B, 11 if use_viewdirs else 8, device=dev) if want_rows else None
    od = torch.empty(2, B, 3, device=dev) if want_od else None
    target = torch.empty(B, 3, device=dev)
    ex_out = torch.empty(len(ex), B, device=dev) if ex else None
    coords = torch.empty(B, 2, device=dev, dtype=torch.int64) if want_coords else None
    ptrs = (C.c_void_p * max(1, len(ex)))(*[e.data_ptr() for e in ex])
    r = rng.c(0) if (rng is not None and select_inds is None) else None
    _lib.check(_lib.load().cnerf_sample_pixels(C.byref(cfg), _p(select_inds), C.byref(r) if r is not None else None, _p(image), ptrs,
                                               _p(rows), _p(od), _p(target), _p(ex_out), _p(coords), _stream()),
               "cnerf_sample_pixels")
    return rows, od, target, ex_out, coords


# ------------------------------------------------------------------------------------------ warp / masks
def warp_points(P: Tensor, w2c, K, H: int, W: int, flip: bool):
    P = _chk(P.reshape(-1, 3), "P")
    N = P.shape[0]
    dev = P.device
    Xc = torch.empty(N, 3, device=dev)
    px, py = torch.empty(N, device=dev), torch.empty(N, device=dev)
    inb = torch.empty(N, device=dev, dtype=torch.uint8)
    if isinstance(w2c, torch.Tensor):
        w2c = w2c.detach().cpu().numpy()
    _lib.check(_lib.load().cnerf_warp_points(_p(P), N, _f4(w2c), float(K[0][0]), float(K[1][1]), float(K[0][2]),
                                             float(K[1][2]), H, W, int(flip), _p(Xc), _p(px), _p(py), _p(inb),
                                             _stream()), "cnerf_warp_points")
    return Xc, px, py, inb.bool()

_PINNED_META = {}


def _pinned_meta(dev):
    """A small ring of pinned int32[8] buffers per device (pinned allocations cost ~100 us each: never per step)."""
    ring = _PINNED_META.get(dev.index)
    if ring is None:
        ring = _PINNED_META[dev.index] = [[torch.empty(8, dtype=torch.int32).pin_memory() for _ in range(8)], 0]
    ring[1] = (ring[1] + 1) % len(ring[0])
    return ring[0][ring[1]]


def ss_ref_rays(rays_o: Tensor, rays_d: Tensor, depth: Tensor, w2c_ref, c2w_ref, K, H: int, W: int, image: Tensor, depth_ref: Tensor,
                thr0: float, near: float, far: float, use_viewdirs: bool, ndc: bool, ndc_coef=(0.0, 0.0), flip: bool = False,
                want_rows: bool = True):
    """cnerf_ss_ref_rays (VT:905-925): one launch + ONE 32-byte read-back (the ray count M of the second render; the reference
    synchronises at every boolean index and every threshold doubling) -> dict(M, k, thr (python float), rows [M, 8|11] | None,
    rays_od [2, M, 3] (views of a [2, N, 3] buffer), target [M, 3], depth_tgt [M], depth_diff [M], inb [N] uint8, mask [M] uint8,
    sel [N] float, rank [N] int32).  image [H, W, >=3] and depth_ref [H, W] live on the device; w2c_ref / c2w_ref are host 3x4|4x4."""
    rays_o, rays_d, depth = _chk(rays_o.reshape(-1, 3), "rays_o"), _chk(rays_d.reshape(-1, 3), "rays_d"), _chk(depth.reshape(-1), "depth")
    image, depth_ref = _chk(image, "image"), _chk(depth_ref, "depth_ref")
    N, dev = rays_o.shape[0], rays_o.device
    if image.dim() != 3 or image.shape[0] != H or image.shape[1] != W or image.shape[2] < 3 or depth_ref.numel() != H * W:
        raise CnerfError(f"ss_ref_rays: image must be [{H}, {W}, >=3] and depth_ref [{H}, {W}]")
    if rays_d.shape[0] != N or depth.shape[0] != N or N == 0:
        raise CnerfError("ss_ref_rays: rays_o / rays_d / depth must describe the same (non-empty) batch")
    cfg = SsWarp()
    r = cfg.ref
    r.H, r.W, r.fx, r.fy, r.cx, r.cy = int(H), int(W), float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    r.c2w = _f4(c2w_ref.detach().cpu().numpy() if isinstance(c2w_ref, torch.Tensor) else c2w_ref)
    r.near, r.far, r.use_viewdirs, r.ndc = float(near), float(far), int(use_viewdirs), int(ndc)
    r.ndc_ax, r.ndc_ay, r.first = float(ndc_coef[0]), float(ndc_coef[1]), 0
    cfg.w2c = _f4(w2c_ref.detach().cpu().numpy() if isinstance(w2c_ref, torch.Tensor) else w2c_ref)
    cfg.flip, cfg.image_ch, cfg.thr0 = int(flip), int(image.shape[2]), float(thr0)
    rows = torch.empty(N, 11 if use_viewdirs else 8, device=dev) if want_rows else None
    od = torch.empty(2, N, 3, device=dev)
    target, dtgt, diff = torch.empty(N, 3, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)
    inb, mask = torch.empty(N, device=dev, dtype=torch.uint8), torch.empty(N, device=dev, dtype=torch.uint8)
    sel, rank = torch.empty(N, device=dev), torch.empty(N, device=dev, dtype=torch.int32)
    meta = torch.empty(8, device=dev, dtype=torch.int32)
    _lib.check(_lib.load().cnerf_ss_ref_rays(C.byref(cfg), _p(rays_o), _p(rays_d), _p(depth), N, _p(image), _p(depth_ref), _p(rows),
                                             _p(od), _p(target), _p(dtgt), _p(diff), _p(inb), _p(mask), _p(sel), _p(rank), _p(meta),
                                             _stream()), "cnerf_ss_ref_rays")
    host = _pinned_meta(dev)
    host.copy_(meta, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    ev.synchronize()                      # the ONE host wait of the block: the second render's launch dimensions need M
    M, k = int(host[0]), int(host[1])
    thr = float(np.int32(int(host[2])).view(np.float32))
    return dict(M=M, k=k, thr=thr, rows=None if rows is None else rows[:M], rays_od=od[:, :M], target=target[:M], depth_tgt=dtgt[:M],
                depth_diff=diff[:M], inb=inb, mask=mask[:M], sel=sel, rank=rank)


def _ss_cfg(w2c_ref, c2w_ref, K, H, W, image, thr0, near, far, use_viewdirs, ndc, ndc_coef, flip) -> "SsWarp":
    cfg = SsWarp()
    r = cfg.ref
    r.H, r.W, r.fx, r.fy, r.cx, r.cy = int(H), int(W), float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    r.c2w = _f4(c2w_ref.detach().cpu().numpy() if isinstance(c2w_ref, torch.Tensor) else c2w_ref)
    r.near, r.far, r.use_viewdirs, r.ndc = float(near), float(far), int(use_viewdirs), int(ndc)
    r.ndc_ax, r.ndc_ay, r.first = float(ndc_coef[0]), float(ndc_coef[1]), 0
    cfg.w2c = _f4(w2c_ref.detach().cpu().numpy() if isinstance(w2c_ref, torch.Tensor) else w2c_ref)
    cfg.flip, cfg.image_ch, cfg.thr0 = int(flip), int(image.shape[2]), float(thr0)
    return cfg


def ss_batch(rays_o: Tensor, rays_d: Tensor, depth: Tensor, target_s: Tensor, w2c_ref, c2w_ref, K, H: int, W: int, image: Tensor,
             depth_ref: Tensor, thr0: float, near: float, far: float, use_viewdirs: bool, ndc: bool, ndc_coef=(0.0, 0.0),
             flip: bool = False, amin_global: Optional[Tensor] = None):
    """cnerf_ss_batch (VT:899-925 for the one-render step): ONE launch, NO read-back -> dict of DEVICE tensors:
    rows [2N, 8|11], target [2N, 3], prior [2N], mask [2N] (the combined batch: N primary rays | M warped rays | N - M padding),
    live int32 [1] = N + M, meta int32 [8] (M, k, bits of thr, NaN flag, bits of the local min |diff|, number of selected primary rays),
    rays_od [2, N, 3], depth_diff [N], inb [N] uint8, occ [N] uint8 (the occlusion mask over the first M), sel [N], rank [N] int32 —
    the per-M outputs keep their capacity N (the host does not know M; `ss_host_view` slices them after a synchronisation)."""
    rays_o, rays_d, depth = _chk(rays_o.reshape(-1, 3), "rays_o"), _chk(rays_d.reshape(-1, 3), "rays_d"), _chk(depth.reshape(-1), "depth")
    target_s = _chk(target_s.reshape(-1, 3), "target_s")
    image, depth_ref, amin_global = _chk(image, "image"), _chk(depth_ref, "depth_ref"), _chk(amin_global, "amin_global")
    N, dev = rays_o.shape[0], rays_o.device
    if image.dim() != 3 or image.shape[0] != H or image.shape[1] != W or image.shape[2] < 3 or depth_ref.numel() != H * W:
        raise CnerfError(f"ss_batch: image must be [{H}, {W}, >=3] and depth_ref [{H}, {W}]")
    if rays_d.shape[0] != N or depth.shape[0] != N or target_s.shape[0] != N or N == 0:
        raise CnerfError("ss_batch: rays_o / rays_d / depth / target_s must describe the same (non-empty) batch")
    cfg = _ss_cfg(w2c_ref, c2w_ref, K, H, W, image, thr0, near, far, use_viewdirs, ndc, ndc_coef, flip)
    rows = torch.empty(2 * N, 11 if use_viewdirs else 8, device=dev)
    target, prior, mask2 = torch.empty(2 * N, 3, device=dev), torch.empty(2 * N, device=dev), torch.empty(2 * N, device=dev)
    live, meta = torch.empty(1, device=dev, dtype=torch.int32), torch.empty(8, device=dev, dtype=torch.int32)
    od, diff = torch.empty(2, N, 3, device=dev), torch.empty(N, device=dev)
    inb, occ = torch.empty(N, device=dev, dtype=torch.uint8), torch.empty(N, device=dev, dtype=torch.uint8)
    sel, rank = torch.empty(N, device=dev), torch.empty(N, device=dev, dtype=torch.int32)
    _lib.check(_lib.load().cnerf_ss_batch(C.byref(cfg), _p(rays_o), _p(rays_d), _p(depth), _p(target_s), N, _p(image), _p(depth_ref),
                                          _p(amin_global), _p(rows), _p(target), _p(prior), _p(mask2), _p(live), _p(od), _p(diff),
                                          _p(inb), _p(occ), _p(sel), _p(rank), _p(meta), _stream()), "cnerf_ss_batch")
    return dict(N=N, rows=rows, target=target, prior=prior, mask=mask2, live=live, meta=meta, rays_od=od, depth_diff=diff, inb=inb,
                occ=occ, sel=sel, rank=rank)


def hard_mask_pair(H, W, K, c2w_tgt, w2c_ref, depth_tgt: Tensor, depth_ref: Tensor, thr0: float, chunk: int,
                   mask: Tensor, want_thr: bool = False):
    depth_tgt, depth_ref = _chk(depth_tgt, "depth_tgt"), _chk(depth_ref, "depth_ref")
    nchunks = (H * W + chunk - 1) // chunk
    thr = torch.empty(nchunks, device=mask.device) if want_thr else None
    _lib.check(_lib.load().cnerf_hard_mask_pair(H, W, float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]),
                                                _f4(c2w_tgt), _f4(w2c_ref), _p(depth_tgt), _p(depth_ref), float(thr0),
                                                chunk, _p(mask), _p(thr), _stream()), "cnerf_hard_mask_pair")
    return thr


# ------------------------------------------------------------------------------------------ loss / optimiser
MASKED_LOSS_SINGLE_WORKGROUP = 16384     # rays up to which cnerf_masked_loss runs as one workgroup


def masked_loss(rgb, target, depth, prior, mask, far: float, coef: float, counts=None, g_scale: float = 1.0,
                want_grads: bool = True):
    rgb, target = _chk(rgb, "rgb"), _chk(target, "target")
    depth, prior, mask, counts = _chk(depth, "depth"), _chk(prior, "prior"), _chk(mask, "mask"), _chk(counts, "counts")
    B = rgb.shape[0]
    dev = rgb.device
    loss = torch.empty(2, device=dev)
    d_rgb = torch.empty_like(rgb) if want_grads else None
    d_depth = torch.empty(B, device=dev) if (want_grads and depth is not None) else None
    lib = _lib.load()
    # beyond one workgroup's comfortable size: one workgroup per 16384 rays + a fixed-order second stage (needs the workspace)
    ws = torch.empty(lib.cnerf_loss_ws_floats() // 2, device=dev, dtype=torch.float64) if B > MASKED_LOSS_SINGLE_WORKGROUP else None
    _lib.check(lib.cnerf_masked_loss(_p(rgb), _p(target), _p(depth), _p(prior), _p(mask), B, float(far), float(coef), _p(counts),
                                     float(g_scale), _p(loss), _p(d_rgb), _p(d_depth), _p(ws), _stream()), "cnerf_masked_loss")
    return loss, d_rgb, d_depth


MSE_SINGLE_WORKGROUP = 65536    # elements up to which cnerf_mse's one workgroup is the faster form (launch-bound)


def mse(x: Tensor, y: Tensor, want_grad: bool = True):
    """cnerf_mse: (mean((x - y)^2) as a 0-d tensor, d loss / d x | None)."""
    x, y = _chk(x, "x"), _chk(y, "y")
    loss = torch.empty(1, device=x.device)
    d_x = torch.empty_like(x) if want_grad else None
    lib, n = _lib.load(), x.numel()
    if n > MSE_SINGLE_WORKGROUP:     # whole images: one workgroup per 16384 elements + a fixed-order second stage
        ws = torch.empty(lib.cnerf_mse_ws_floats(n) // 2, device=x.device, dtype=torch.float64)
        _lib.check(lib.cnerf_mse_ws(_p(x), _p(y), n, _p(loss), _p(d_x), _p(ws), _stream()), "cnerf_mse_ws")
    else:
        _lib.check(lib.cnerf_mse(_p(x), _p(y), n, _p(loss), _p(d_x), _stream()), "cnerf_mse")
    return loss[0], d_x


def soft_lp_loss(x: Tensor, y: Tensor, coef: float, want_grad: bool = True):
    """cnerf_soft_lp_loss (V:58): (sum(w d^2) / sum(w) with w = |d|^coef + 1 and a detached denominator, d loss / d x | None)."""
    x, y = _chk(x, "x"), _chk(y, "y")
    if x.shape != y.shape or x.numel() == 0:
        raise CnerfError("soft_lp_loss: x and y must be non-empty tensors of one shape")
    loss = torch.empty(1, device=x.device)
    d_x = torch.empty_like(x) if want_grad else None
    _lib.check(_lib.load().cnerf_soft_lp_loss(_p(x), _p(y), x.numel(), float(coef), _p(loss), _p(d_x), _stream()), "cnerf_soft_lp_loss")
    return loss[0], d_x


def patch_depth_loss(depth_pred: Tensor, mono: Tensor, P: int, n: int, g_scale: float = 1.0, want_grad: bool = True):
    """f-5 (V:1678-1720): (loss[1], d_depth[P*n] | None) over the first P*n rays."""
    depth_pred, mono = _chk(depth_pred, "depth_pred"), _chk(mono, "mono")
    if depth_pred.numel() < P * n or mono.numel() < P * n:
        raise CnerfError(f"patch_depth_loss: need {P}x{n} values, got {depth_pred.numel()} / {mono.numel()}")
    loss = torch.empty(1, device=depth_pred.device)
    d = torch.empty(P * n, device=depth_pred.device) if want_grad else None
    _lib.check(_lib.load().cnerf_patch_depth_loss(_p(depth_pred), _p(mono), int(P), int(n), float(g_scale), _p(loss),
                                                  _p(d), _stream()), "cnerf_patch_depth_loss")
    return loss, d


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8,
              clip: float = 0.0, grad_scale: float = 1.0):
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise CnerfError(f"adam_step: {n} must be a contiguous fp32 GPU tensor")
    _lib.check(_lib.load().cnerf_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), int(step), float(lr), float(beta1),
                                           float(beta2), float(eps), float(clip), float(grad_scale), _stream()),
               "cnerf_adam_step")


def adam_hyper(out_host: Tensor, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, clip: float = 0.0,
               grad_scale: float = 1.0):
    """cnerf_adam_hyper: the 8 scalars of step `step` into a (pinned) HOST float tensor."""
    if out_host.is_cuda or out_host.dtype != torch.float32 or out_host.numel() < 8 or not out_host.is_contiguous():
        raise CnerfError("adam_hyper: need a contiguous fp32 host tensor of 8 floats")
    _lib.check(_lib.load().cnerf_adam_hyper(int(step), float(lr), float(beta1), float(beta2), float(eps), float(clip),
                                            float(grad_scale), C.c_void_p(out_host.data_ptr())), "cnerf_adam_hyper")


def adam_step_dev(p: Tensor, g: Tensor, m: Tensor, v: Tensor, hyp_dev: Tensor):
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (hyp_dev, "hyp")):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise CnerfError(f"adam_step_dev: {n} must be a contiguous fp32 GPU tensor")
    _lib.check(_lib.load().cnerf_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyp_dev), _stream()),
               "cnerf_adam_step_dev")
