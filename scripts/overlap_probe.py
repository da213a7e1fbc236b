#!/usr/bin/env python3
"""VERDICT r04 item 2(b), measured: does the coarse level's backward (its loss term is final right after the coarse compositing,
R:397 detaches the fine depths) pay when it is issued on a SECOND stream under the fine level's forward, instead of riding in the
merged coarse+fine backward launches?  At B rays per GPU (default 512 = the C4 shard), HIP events around the whole sequence:

  merged (product):   fwd_fine ; dgrad_pair(fine, coarse) ; wgrad_pair(fine, coarse)                       one stream
  overlapped:         stream A: fwd_fine            | stream B: dgrad(coarse) ; wgrad(coarse)   -> join -> dgrad(fine) ; wgrad(fine)
  split, one stream:  fwd_fine ; dgrad(coarse) ; wgrad(coarse) ; dgrad(fine) ; wgrad(fine)                 (what the overlap must beat)

usage: python scripts/overlap_probe.py [B] [reps]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import _inputs as I  # noqa: E402
from consistentnerf_amd import _lib, ops  # noqa: E402
from consistentnerf_amd.run_nerf_helpers import NeRF  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")


def main():
    from consistentnerf_amd.run_nerf import _packed
    lib = _lib.load()
    nets = []
    for seed, S in ((22, 192), (21, 64)):
        sd = I.nerf_state_dict(8, 256, 10, 4, 5, True, seed=seed)
        m = NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.to(dev)
        spec, packed = m.spec(), _packed(m)
        rays = torch.from_numpy(I.ray_batch(B, seed=5, near=2.125, far=4.67)).to(dev)
        z = ops.coarse_z(rays, S, torch.rand(B, S, device=dev), False)
        raw, stash = ops.mlp_forward(spec, packed, B, S, rays=rays, z=z, want_stash=True)
        net = spec.c()
        ws = torch.empty(lib.cnerf_mlp_bwd_ws_floats(C.byref(net), B * S), device=dev)
        grads = [torch.empty(s, device=dev) for s in spec.tensor_shapes()]
        nets.append(dict(spec=spec, net=net, packed=packed, S=S, stash=stash, d_raw=torch.randn_like(raw), ws=ws, grads=grads,
                         ptrs=ops._ptrs(grads), rays=rays, z=z))
    f, c = nets
    p = ops._p
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    h = lambda s: C.c_void_p(s.cuda_stream)  # noqa: E731

    def fwd_fine(s):
        with torch.cuda.stream(s):
            ops.mlp_forward(f["spec"], f["packed"], B, f["S"], rays=f["rays"], z=f["z"], want_stash=True)

    def bwd_one(n, s):
        _lib.check(lib.cnerf_mlp_dgrad(C.byref(n["net"]), p(n["packed"]), p(n["d_raw"]), B, n["S"], p(n["stash"]), p(n["ws"]), h(s)), "dgrad")
        _lib.check(lib.cnerf_mlp_wgrad(C.byref(n["net"]), B, n["S"], p(n["stash"]), p(n["ws"]), C.byref(n["ptrs"]), 0, h(s)), "wgrad")

    def bwd_pair(s):
        _lib.check(lib.cnerf_mlp_dgrad_pair(C.byref(f["net"]), p(f["packed"]), p(f["d_raw"]), B, f["S"], p(f["stash"]), p(f["ws"]),
                                            C.byref(c["net"]), p(c["packed"]), p(c["d_raw"]), B, c["S"], p(c["stash"]), p(c["ws"]), h(s)), "dgrad_pair")
        _lib.check(lib.cnerf_mlp_wgrad_pair(C.byref(f["net"]), B, f["S"], p(f["stash"]), p(f["ws"]), C.byref(f["ptrs"]),
                                            C.byref(c["net"]), B, c["S"], p(c["stash"]), p(c["ws"]), C.byref(c["ptrs"]), 0, h(s)), "wgrad_pair")

    def merged():
        fwd_fine(sA)
        bwd_pair(sA)

    def split_serial():
        fwd_fine(sA)
        bwd_one(c, sA)
        bwd_one(f, sA)

    def overlapped():
        sB.wait_stream(sA)          # (the coarse compositing backward would sit here)
        fwd_fine(sA)
        bwd_one(c, sB)
        sA.wait_stream(sB)
        bwd_one(f, sA)

    def timeit(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(sA)
        for _ in range(REPS):
            fn()
        e1.record(sA)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / REPS

    res = {}
    for rnd in range(3):            # interleaved repetitions: the box's clock drifts by ~1 % over seconds
        for name, fn in (("merged", merged), ("split_serial", split_serial), ("overlapped", overlapped)):
            res.setdefault(name, []).append(timeit(fn))
    print(f"B={B} rays (fine {B * 192} + coarse {B * 64} points), fwd_fine + backward of both levels, ms per sequence (3 rounds):")
    for k, v in res.items():
        print(f"  {k:13s} " + "  ".join(f"{x:.4f}" for x in v) + f"   best {min(v):.4f}")


main()
