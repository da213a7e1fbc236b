"""Time cnerf_composite_fwd_mse (the loss-folded compositing forward) against the plain compositing + cnerf_mse pair, for the
publish forms of csrc/composite.hip: default (write-through store + vmcnt(0) + relaxed tickets), -DCN_MSE_RELEASE_TICKET (agent-scope
acq_rel RMW tickets) and -DCN_MSE_HEAVY_FENCE (__threadfence around them).  Usage (GPU box):
    CNERF_LIB_PATH=variants/libcnerf_<tag>.so python scripts/ticket_timing.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import _inputs as I  # noqa: E402
from consistentnerf_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
print("lib:", os.environ.get("CNERF_LIB_PATH", "product"))
for B, S in ((4096, 192), (4096, 64), (512, 192), (65280, 64)):
    g = torch.Generator(device=dev).manual_seed(1)
    raw = torch.randn(B, S, 4, device=dev, generator=g)
    rays = torch.from_numpy(I.ray_batch(B, seed=2)).to(dev)
    z = ops.coarse_z(rays, S, torch.rand(B, S, device=dev, generator=g), False)
    tgt = torch.rand(B, 3, device=dev, generator=g)

    def bench(fn, n=300):
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    fused = bench(lambda: ops.composite_forward_mse(raw, z, rays, None, False, tgt))
    plain = bench(lambda: ops.composite_forward(raw, z, rays, None, False))
    pair = bench(lambda: ops.mse(ops.composite_forward(raw, z, rays, None, False)[0], tgt, want_grad=True))
    print(f"B={B:6d} S={S:3d}  fused {fused:7.2f} us   compositing alone {plain:7.2f} us   compositing + mse_k {pair:7.2f} us")
