#!/usr/bin/env python3
"""Per-workgroup trace of the merged weight-gradient launch (GPU box, -DCN_TIMING build):
   CNERF_LIB_PATH=variants/libcnerf_timing.so python scripts/wgrad_trace.py [B]
For every GEMM job of the plan: workgroup lifetime (100 MHz realtime counter), microseconds per 32-point slab, start times;
and how the blocks were spread over XCDs / CUs — the data the range planner's model (csrc/wgrad.hip::slab_us) is fitted to."""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
NS = 40


def main():
    from consistentnerf_amd.run_nerf import _packed
    lib = _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    nets = []
    for seed, S in ((22, 192), (21, 64)):
        sd = I.nerf_state_dict(8, 256, 10, 4, 5, True, seed=seed)
        m = NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.to(dev)
        spec, packed = m.spec(), _packed(m)
        rays = torch.from_numpy(I.ray_batch(B, seed=5, near=2.125, far=4.67)).to(dev)
        z = ops.coarse_z(rays, S, torch.rand(B, S, device=dev), False)
        r_, stash = ops.mlp_forward(spec, packed, B, S, rays=rays, z=z, want_stash=True)
        net = spec.c()
        ws = torch.empty(lib.cnerf_mlp_bwd_ws_floats(C.byref(net), B * S), device=dev)
        grads = [torch.empty(s, device=dev) for s in spec.tensor_shapes()]
        nets.append(dict(net=net, packed=packed, S=S, stash=stash, d_raw=torch.randn_like(r_), ws=ws, ptrs=ops._ptrs(grads), grads=grads))
    f, c = nets
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    p = ops._p
    lib.cnerf_mlp_dgrad_pair(C.byref(f["net"]), p(f["packed"]), p(f["d_raw"]), B, f["S"], p(f["stash"]), p(f["ws"]),
                             C.byref(c["net"]), p(c["packed"]), p(c["d_raw"]), B, c["S"], p(c["stash"]), p(c["ws"]), st())
    fn = lambda: lib.cnerf_mlp_wgrad_pair(C.byref(f["net"]), B, f["S"], p(f["stash"]), p(f["ws"]), C.byref(f["ptrs"]),  # noqa: E731
                                          C.byref(c["net"]), B, c["S"], p(c["stash"]), p(c["ws"]), C.byref(c["ptrs"]), 0, st())
    for _ in range(3):
        rc = fn()
        if rc != 0:
            raise SystemExit(f"cnerf_mlp_wgrad_pair -> {rc}")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    plan = raw.cnerf_debug_wgrad_plan
    plan.restype = C.c_int
    plan.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
    out = (C.c_int * (7 * 48))()
    nj = plan(C.byref(f["net"]), B * 192, C.byref(c["net"]), B * 64, out, 48)   # (net, N, K, tiles, ranges, chunk, tensor) per job
    assert nj > 0, nj
    jobs = [tuple(out[7 * i:7 * i + 6]) for i in range(nj)]
    nblocks = sum(j[4] for j in jobs)
    getter = raw.cnerf_debug_timing_wgrad
    getter.restype, getter.argtypes = C.c_int, [C.c_void_p, C.c_int64]
    nw = min(65536, 4 * nblocks)
    buf = np.zeros(nw * NS, dtype=np.uint64)
    assert getter(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(nw, NS).astype(np.float64)
    t0 = t[:, 6][t[:, 6] > 0].min()
    print(f"B={B}: wgrad pair (+reduce) {ms:.3f} ms; {nblocks} workgroups; "
          f"measured span of the wgrad kernel {(t[:, 7].max() - t0) / 100:.0f} us")
    b0 = 0
    for (n_, N, K, tiles, ns, ch) in jobs:
        r = t[4 * b0:4 * (b0 + ns)].reshape(ns, 4, NS)
        beg = np.where(r[:, :, 6] > 0, r[:, :, 6], np.inf).min(1)      # (idle waves of a narrow GEMM leave no record)
        life = (r[:, :, 7].max(1) - beg) / 100.0
        start = (beg - t0) / 100.0
        slabs = ch // 32
        cyc = r[:, :, 5].max(1)
        act = r[:, :, 5] > 0
        print(f"  net{n_} {N:3d}x{K:3d} tiles {tiles:2d} ranges {ns:3d} x {ch:6d} pts: life {life.mean():7.1f} us (min {life.min():7.1f} max {life.max():7.1f}) "
              f"= {life.mean() / slabs:6.3f} us/slab, {cyc.mean() / slabs:7.0f} cyc/slab; starts {start.min():7.1f}..{start.max():7.1f} us; "
              f"mfma-loop {100 * r[:, :, 2][act].mean() / r[:, :, 5][act].mean():4.1f}% barrier {100 * r[:, :, 1][act].mean() / r[:, :, 5][act].mean():4.1f}% "
              f"prologue {r[:, :, 0][act].mean():6.0f} cyc epilogue {r[:, :, 3][act].mean():6.0f} cyc of {r[:, :, 5][act].mean():8.0f}")
        b0 += ns
    # dispatch pattern: XCC of block i, and how many distinct CUs were used
    w0 = t[0::4]      # wave 0 of every block is always active
    xcc = (w0[:, 9].astype(np.int64) & 0xf)
    hw = w0[:, 8].astype(np.int64)
    cu = ((hw >> 13) & 7) * 16 + ((hw >> 8) & 0xf)      # SE_ID x CU_ID
    print("  XCC of blocks 0..31:", xcc[:32].tolist())
    print("  block i on XCC i % 8:", float((xcc[:nblocks] == (np.arange(nblocks) % 8)).mean()))
    print("  distinct (xcc, se, cu):", len({(int(a), int(b)) for a, b in zip(xcc[:nblocks], cu[:nblocks])}))
    # busy time per CU
    key = xcc[:nblocks] * 1000 + cu[:nblocks]
    life = (t[0::4][:nblocks, 7] - t[0::4][:nblocks, 6]) / 100.0
    busy = {}
    for k_, l_ in zip(key, life):
        busy[int(k_)] = busy.get(int(k_), 0.0) + l_
    bv = np.array(list(busy.values()))
    print(f"  per-CU busy time: mean {bv.mean():.0f} us, min {bv.min():.0f}, max {bv.max():.0f}")


main()
