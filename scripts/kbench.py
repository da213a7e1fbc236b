#!/usr/bin/env python3
"""Kernel micro-bench (GPU box): times cnerf_mlp_fwd (inference / training), dgrad, wgrad, composite, resample at
C2 sizes with HIP events on the launch stream.  usage: python scripts/kbench.py [B] [reps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import _inputs as I  # noqa: E402
from consistentnerf_amd import ops  # noqa: E402
from consistentnerf_amd.run_nerf_helpers import NeRF  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
MAC = {"fwd": 593408, "dgrad": 557696, "wgrad": 593408}


def timeit(fn, reps=REPS):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def spec_g_floats(lib, net, Mp):
    """floats of the dZ part of the backward workspace (g_rows * Mp): the split partials follow it"""
    import ctypes as C
    a, b = lib.cnerf_mlp_bwd_ws_floats(C.byref(net), 32 * 4096), lib.cnerf_mlp_bwd_ws_floats(C.byref(net), 32 * 4096 + 32)
    return (b - a) * (Mp // 32)        # (the partial slices do not grow between these two sizes)


def main():
    from consistentnerf_amd.run_nerf import _packed
    sd = I.nerf_state_dict(8, 256, 10, 4, 5, True, seed=21)
    m = NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev)
    spec, packed = m.spec(), _packed(m)
    rays = torch.from_numpy(I.ray_batch(B, seed=5, near=2.125, far=4.67)).to(dev)
    for S in [int(x) for x in os.environ.get("KBENCH_LEVELS", "192,64").split(",")]:
        M = B * S
        z = ops.coarse_z(rays, S, torch.rand(B, S, device=dev), False)
        t_inf = timeit(lambda: ops.mlp_forward(spec, packed, B, S, rays=rays, z=z))
        raw, stash = ops.mlp_forward(spec, packed, B, S, rays=rays, z=z, want_stash=True)
        t_tr = timeit(lambda: ops.mlp_forward(spec, packed, B, S, rays=rays, z=z, want_stash=True))
        d_raw = torch.randn_like(raw)
        import ctypes as C
        from consistentnerf_amd import _lib
        lib, net = _lib.load(), spec.c()
        ws = torch.empty(lib.cnerf_mlp_bwd_ws_floats(C.byref(net), M), device=dev)
        grads = [torch.empty(s, device=dev) for s in spec.tensor_shapes()]
        ptrs = ops._ptrs(grads)
        st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
        t_dg = timeit(lambda: lib.cnerf_mlp_dgrad(C.byref(net), ops._p(packed), ops._p(d_raw), B, S, ops._p(stash),
                                                  ops._p(ws), st()))
        t_wg = timeit(lambda: lib.cnerf_mlp_wgrad(C.byref(net), B, S, ops._p(stash), ops._p(ws), C.byref(ptrs), 0, st()))
        tf = lambda k, ms: 2 * MAC[k] * M / (ms * 1e-3) / 1e12  # noqa: E731
        print(f"S={S:4d} M={M:8d}  fwd(inf) {t_inf:7.3f} ms {tf('fwd', t_inf):6.1f} TF | fwd(train) {t_tr:7.3f} ms "
              f"{tf('fwd', t_tr):6.1f} TF | dgrad {t_dg:7.3f} ms {tf('dgrad', t_dg):6.1f} TF | wgrad {t_wg:7.3f} ms "
              f"{tf('wgrad', t_wg):6.1f} TF", flush=True)
        for planes in (1, 2, 3):
            pk = ops.pack_weights_bf(spec, m.kernel_tensors(), planes)
            t_bf = timeit(lambda: ops.mlp_forward_bf(spec, pk, planes, B, S, rays=rays, z=z))
            os.environ["CNERF_BF_PERWAVE"] = "1"
            t_pw = timeit(lambda: ops.mlp_forward_bf(spec, pk, planes, B, S, rays=rays, z=z))
            del os.environ["CNERF_BF_PERWAVE"]
            print(f"          fwd bf16 x{planes} (inference, opt-in) {t_bf:7.3f} ms = {tf('fwd', t_bf):7.1f} TFLOP/s fp32-equivalent, "
                  f"{t_inf / t_bf:5.2f}x the fp32 kernel  (per-wave panel streaming instead of the shared LDS ring: {t_pw:7.3f} ms)",
                  flush=True)
        # the opt-in bf16x3 TRAINING forward (same stash): time, and its stash / raw against the fp32 kernel's
        pk3 = ops.pack_weights_bf(spec, m.kernel_tensors(), 3)
        raw3, stash3 = ops.mlp_forward_bf_train(spec, pk3, B, S, rays=rays, z=z)
        t_tr3 = timeit(lambda: ops.mlp_forward_bf_train(spec, pk3, B, S, rays=rays, z=z))
        torch.cuda.synchronize()
        d_raw3 = float((raw3 - raw).abs().max() / raw.abs().max())
        d_st3 = float((stash3 - stash).abs().max())
        print(f"          fwd bf16x3 TRAINING (opt-in, stash written) {t_tr3:7.3f} ms = {tf('fwd', t_tr3):7.1f} TFLOP/s fp32-equivalent, "
              f"{t_tr / t_tr3:5.2f}x the fp32 training forward; raw vs fp32 kernel {d_raw3:.2e} of max, stash max|d| {d_st3:.2e} "
              f"(activations + sign-bit words reinterpreted as floats)", flush=True)
        del raw3, stash3
        # the opt-in bf16x3 dgrad: time, and its gradient workspace against the fp32 kernel's (same stash, same d_raw)
        lib.cnerf_mlp_dgrad(C.byref(net), ops._p(packed), ops._p(d_raw), B, S, ops._p(stash), ops._p(ws), st())
        Mp = (M + 31) // 32 * 32
        ws.zero_(); lib.cnerf_mlp_dgrad(C.byref(net), ops._p(packed), ops._p(d_raw), B, S, ops._p(stash), ops._p(ws), st())
        ws3 = torch.zeros_like(ws)
        rc = lib.cnerf_mlp_dgrad_bf(C.byref(net), ops._p(pk3), ops._p(d_raw), B, S, ops._p(stash), ops._p(ws3), st())
        torch.cuda.synchronize()
        t_dg3 = timeit(lambda: lib.cnerf_mlp_dgrad_bf(C.byref(net), ops._p(pk3), ops._p(d_raw), B, S, ops._p(stash), ops._p(ws3), st()))
        ng = spec_g_floats(lib, net, Mp)
        fin = torch.isfinite(ws[:ng]) & torch.isfinite(ws3[:ng])      # (columns no kernel writes hold whatever torch.empty left)
        dG = float((ws3[:ng] - ws[:ng])[fin].abs().max() / ws[:ng][fin].abs().max())
        print(f"          dgrad bf16x3 (opt-in) rc={rc} {t_dg3:7.3f} ms = {tf('dgrad', t_dg3):7.1f} TFLOP/s fp32-equivalent, {t_dg / t_dg3:5.2f}x "
              f"the fp32 dgrad; gradient workspace vs fp32 kernel: max|d| {dG:.2e} of max", flush=True)
        del ws3
        # the opt-in bf16x3 wgrad (wide GEMMs on the bf16 cores, narrow ones fp32) on the SAME workspace: time + gradients vs fp32
        lib.cnerf_mlp_dgrad(C.byref(net), ops._p(packed), ops._p(d_raw), B, S, ops._p(stash), ops._p(ws), st())
        lib.cnerf_mlp_wgrad(C.byref(net), B, S, ops._p(stash), ops._p(ws), C.byref(ptrs), 0, st())
        grads3 = [torch.zeros_like(g_) for g_ in grads]
        ptrs3 = ops._ptrs(grads3)
        rc = lib.cnerf_mlp_wgrad_bf(C.byref(net), B, S, ops._p(stash), ops._p(ws), C.byref(ptrs3), 0, st())
        torch.cuda.synchronize()
        t_wg3 = timeit(lambda: lib.cnerf_mlp_wgrad_bf(C.byref(net), B, S, ops._p(stash), ops._p(ws), C.byref(ptrs3), 0, st()))
        worst = max(float((a_ - b_).abs().max() / b_.abs().max().clamp(min=1e-30)) for a_, b_ in zip(grads3, grads) if b_.numel())
        print(f"          wgrad bf16x3 (opt-in) rc={rc} {t_wg3:7.3f} ms = {tf('wgrad', t_wg3):7.1f} TFLOP/s fp32-equivalent, {t_wg / t_wg3:5.2f}x "
              f"the fp32 wgrad; gradients vs fp32 kernel: worst tensor max|d| {worst:.2e} of its max", flush=True)
        del grads3
        if S == 192:
            keep = (spec, packed, d_raw, B, S, stash)
        else:                                   # both levels are around: the paired backward (one dgrad + one wgrad grid)
            fs, fp, fd, fB, fS, fst = keep
            g0 = [torch.empty(s_, device=dev) for s_ in spec.tensor_shapes()]
            g1 = [torch.empty(s_, device=dev) for s_ in spec.tensor_shapes()]
            n0, n1 = fs.c(), spec.c()
            ws0 = torch.empty(lib.cnerf_mlp_bwd_ws_floats(C.byref(n0), fB * fS), device=dev)
            ws1 = torch.empty(lib.cnerf_mlp_bwd_ws_floats(C.byref(n1), M), device=dev)
            p0, p1 = ops._ptrs(g0), ops._ptrs(g1)
            t_dp = timeit(lambda: lib.cnerf_mlp_dgrad_pair(C.byref(n0), ops._p(fp), ops._p(fd), fB, fS, ops._p(fst), ops._p(ws0),
                                                           C.byref(n1), ops._p(packed), ops._p(d_raw), B, S, ops._p(stash),
                                                           ops._p(ws1), st()))
            t_wp = timeit(lambda: lib.cnerf_mlp_wgrad_pair(C.byref(n0), fB, fS, ops._p(fst), ops._p(ws0), C.byref(p0),
                                                           C.byref(n1), B, S, ops._p(stash), ops._p(ws1), C.byref(p1), 0, st()))
            Mt = fB * fS + M
            print(f"pair  M={Mt:8d}  dgrad {t_dp:7.3f} ms {2 * MAC['dgrad'] * Mt / (t_dp * 1e-3) / 1e12:6.1f} TF | wgrad {t_wp:7.3f} ms "
                  f"{2 * MAC['wgrad'] * Mt / (t_wp * 1e-3) / 1e12:6.1f} TF", flush=True)
            del ws0, ws1, keep
        if S == 192:
            w = torch.rand(B, 64, device=dev)
            zc = ops.coarse_z(rays, 64, None, False)
            u = torch.rand(B, 128, device=dev)
            t_rs = timeit(lambda: ops.resample(zc, w, u))
            t_cf = timeit(lambda: ops.composite_forward(raw, z, rays, None, False))
            g = torch.randn(B, 3, device=dev)
            t_cb = timeit(lambda: ops.composite_backward(raw, z, rays, None, False, g, None, None, None))
            print(f"          resample {t_rs*1e3:7.1f} us | composite fwd {t_cf*1e3:7.1f} us ({24*M/t_cf/1e6:6.1f} GB/s) | "
                  f"bwd {t_cb*1e3:7.1f} us ({40*M/t_cb/1e6:6.1f} GB/s)", flush=True)
        if S != 192:
            del stash
        del ws


if __name__ == "__main__":
    main()
