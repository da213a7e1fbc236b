#!/usr/bin/env python3
"""Per-wave phase accounting of mlp_fwd_k / mlp_dgrad_k (GPU box, experiment build with -DCN_TIMING):
   CNERF_LIB_PATH=variants/libcnerf_timing.so python scripts/ktiming.py [B] [S] [train] [which=fwd|bwd]
Prints the mean s_memtime cycles a wave spends in each phase of the kernel, the wave lifetime, how the workgroups
were spread over the kernel's wall time (100 MHz realtime counter), and for how much of a SIMD's time 0 / 1 / 2 of
its resident waves were inside a GEMM phase."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import _inputs as I  # noqa: E402
from consistentnerf_amd import _lib, ops  # noqa: E402
from consistentnerf_amd.run_nerf_helpers import NeRF  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 192
TRAIN = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
WHICH = sys.argv[4] if len(sys.argv) > 4 else "fwd"
PH = ["prologue", "barrier", "gemm", "park", "heads", "total"]
NS = 40


def simd_overlap(t):
    hw, xcc = t[:, 8].astype(np.int64), t[:, 9].astype(np.int64) & 0xf
    key = (xcc << 20) | (((hw >> 13) & 7) << 16) | (((hw >> 8) & 0xf) << 4) | ((hw >> 4) & 3)
    ev = t[:, 10:40].astype(np.int64)
    acc = np.zeros(4)
    for k in np.unique(key)[:128]:
        ints = []
        for i in np.where(key == k)[0]:
            e = ev[i]
            for j in range(0, 30, 2):
                if e[j + 1] > e[j] > 0:
                    ints.append((e[j], 1)); ints.append((e[j + 1], -1))
        ints.sort()
        cur, last = 0, ints[0][0]
        for tm, d in ints:
            acc[min(cur, 3)] += tm - last; last = tm; cur += d
    return acc / acc.sum()


def main():
    from consistentnerf_amd.run_nerf import _packed
    dev = torch.device("cuda:0")
    lib = _lib.load()
    sd = I.nerf_state_dict(8, 256, 10, 4, 5, True, seed=21)
    m = NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev)
    spec, packed = m.spec(), _packed(m)
    rays = torch.from_numpy(I.ray_batch(B, seed=5, near=2.125, far=4.67)).to(dev)
    z = ops.coarse_z(rays, S, torch.rand(B, S, device=dev), False)
    M = B * S
    if WHICH == "fwd":
        fn = lambda: ops.mlp_forward(spec, packed, B, S, rays=rays, z=z, want_stash=TRAIN)  # noqa: E731
        getter, wpb = lib.cnerf_debug_timing, 1
    elif WHICH == "wgrad":
        raise SystemExit("the weight-gradient launch has its own per-workgroup trace: scripts/wgrad_trace.py (1-D grid, ranges per GEMM)")
    else:
        raw, stash = ops.mlp_forward(spec, packed, B, S, rays=rays, z=z, want_stash=True)
        d_raw = torch.randn_like(raw)
        net = spec.c()
        ws = torch.empty(lib.cnerf_mlp_bwd_ws_floats(C.byref(net), M), device=dev)
        st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
        fn = lambda: lib.cnerf_mlp_dgrad(C.byref(net), ops._p(packed), ops._p(d_raw), B, S, ops._p(stash),  # noqa: E731
                                         ops._p(ws), st())
        getter, wpb = lib.cnerf_debug_timing_bwd, 1
    getter.restype, getter.argtypes = C.c_int, [C.c_void_p, C.c_int64]
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    nw = min(65536, wpb * ((M + 31) // 32))
    buf = np.zeros(nw * NS, dtype=np.uint64)
    assert getter(buf.ctypes.data, buf.size) == 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", f"ktiming_{WHICH}_{int(TRAIN)}.npy"), buf.reshape(nw, NS))
    t = buf.reshape(nw, NS).astype(np.float64)
    print(f"{WHICH} B={B} S={S} train={TRAIN}: kernel {ms:.3f} ms, {nw} waves sampled")
    for w in range(wpb):
        r = t[w::wpb]
        tot = r[:, 5].mean()
        print(f"  wave {w}: lifetime {tot:9.0f} cyc | " +
              " | ".join(f"{PH[i]} {r[:, i].mean():8.0f} ({100 * r[:, i].mean() / tot:4.1f}%)" for i in range(5)))
    rt0, rt1 = t[:, 6], t[:, 7]
    span = (rt1.max() - rt0.min()) / 100.0   # us
    life = (rt1 - rt0) / 100.0
    print(f"  realtime: first start -> last end {span:.1f} us; wave lifetime us: mean {life.mean():.1f} "
          f"p5 {np.percentile(life, 5):.1f} p50 {np.percentile(life, 50):.1f} p95 {np.percentile(life, 95):.1f}; "
          f"shader clock ~ {t[:, 5].mean() / life.mean():.0f} MHz")
    # per-GEMM-phase durations of one wave role (mean over waves)
    ev = t[0::wpb, 10:40]
    d = ev[:, 1::2] - ev[:, 0::2]
    print("  mean GEMM phase durations (cycles):", " ".join(f"{x:.0f}" for x in d.mean(axis=0) if x > 0))
    o = simd_overlap(buf.reshape(nw, NS))
    print(f"  SIMD time with 0/1/2/3+ waves inside a GEMM phase: {o[0]:.3f} {o[1]:.3f} {o[2]:.3f} {o[3]:.3f}")


if __name__ == "__main__":
    main()
