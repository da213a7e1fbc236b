#!/usr/bin/env python3
"""Micro-bench of the merged coarse+fine backward (GPU box): cnerf_mlp_dgrad_pair and cnerf_mlp_wgrad_pair at B rays
(fine 192 + coarse 64 samples), HIP events.  usage: python scripts/kbench_pair.py [B] [reps]
CNERF_WGRAD_NSPLIT=<n> forces one range count for every GEMM (the pre-round-3 behaviour was Mp/4096 capped at 128)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import _inputs as I  # noqa: E402
from consistentnerf_amd import _lib, ops  # noqa: E402
from consistentnerf_amd.run_nerf_helpers import NeRF  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
MAC = {"fwd": 593408, "dgrad": 557696, "wgrad": 593408}


def timeit(fn, reps=REPS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


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
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    p = ops._p
    t_f = [timeit(lambda n=n: ops.mlp_forward(n["spec"], n["packed"], B, n["S"], rays=n["rays"], z=n["z"], want_stash=True)) for n in nets]
    t_dg = timeit(lambda: lib.cnerf_mlp_dgrad_pair(C.byref(f["net"]), p(f["packed"]), p(f["d_raw"]), B, f["S"], p(f["stash"]), p(f["ws"]),
                                                   C.byref(c["net"]), p(c["packed"]), p(c["d_raw"]), B, c["S"], p(c["stash"]), p(c["ws"]), st()))
    wg = lambda: lib.cnerf_mlp_wgrad_pair(C.byref(f["net"]), B, f["S"], p(f["stash"]), p(f["ws"]), C.byref(f["ptrs"]),  # noqa: E731
                                          C.byref(c["net"]), B, c["S"], p(c["stash"]), p(c["ws"]), C.byref(c["ptrs"]), 0, st())
    t_wg = timeit(wg)
    M = B * 256
    tf = lambda k, ms, m=M: 2 * MAC[k] * m / (ms * 1e-3) / 1e12  # noqa: E731
    if len(sys.argv) > 3:     # sweep of forced range counts "a,b" (fine, coarse) in one process
        res = []
        for cfg in sys.argv[3:]:
            os.environ["CNERF_WGRAD_NSPLIT"] = cfg
            res.append((timeit(wg), cfg))
        os.environ.pop("CNERF_WGRAD_NSPLIT")
        print(f"B={B} planned {t_wg:.3f} ms | " + " | ".join(f"{cfg}: {ms:.3f}" for ms, cfg in res))
        print(f"   best: {min(res)}")
        return
    print(f"B={B:5d} nsplit={os.environ.get('CNERF_WGRAD_NSPLIT', 'planned'):>7s} | fwd+stash {t_f[0]:.3f} + {t_f[1]:.3f} ms "
          f"({tf('fwd', t_f[0], B * 192):.1f} / {tf('fwd', t_f[1], B * 64):.1f} TF) | dgrad pair {t_dg:.3f} ms {tf('dgrad', t_dg):.1f} TF | "
          f"wgrad pair (+reduce) {t_wg:.3f} ms {tf('wgrad', t_wg):.1f} TF", flush=True)


main()
