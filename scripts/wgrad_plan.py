#!/usr/bin/env python3
"""Prints the point-range plan of the weight-gradient launch (csrc/wgrad.hip::plan_ranges) for a ray batch — no GPU needed.
usage: python scripts/wgrad_plan.py [rays=512]      (CNERF_WGRAD_NSPLIT="a,b" forces the fine / coarse counts)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consistentnerf_amd import _lib, ops  # noqa: E402


def plan(B, spec=None, S=(192, 64)):
    """-> list of (net, N, K, tiles of the busiest wave, ranges, points per range, tensor) in grid order"""
    _lib.load()
    fn = C.CDLL(_lib.LIB_PATH).cnerf_debug_wgrad_plan
    fn.restype = C.c_int
    net = (spec or ops.NetSpec(output_ch=5)).c()
    out = (C.c_int * (7 * 48))()
    second = C.byref(net) if len(S) > 1 else None
    nj = fn(C.byref(net), C.c_int64(B * S[0]), second, C.c_int64(B * S[1] if len(S) > 1 else 0), out, 48)
    assert nj > 0, nj
    return [tuple(out[7 * i:7 * i + 7]) for i in range(nj)]


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    jobs = plan(B)
    cyc = 0.0
    for n_, N, K, tiles, ns, ch, t in jobs:
        print(f"net{n_} {N:3d}x{K:3d} tiles/wave {tiles:2d}  ranges {ns:3d} x {ch:5d} points  ({ch // 32 * (1024 * tiles + 560) / 2300:7.1f} us per workgroup at 2.3 GHz)")
        cyc += (B * (192 if n_ == 0 else 64) / 32) * (1024 * tiles + 560)
    print(f"{sum(j[4] for j in jobs)} workgroups; {cyc / 2300 / 256:.0f} us per CU at 2.3 GHz if perfectly packed on 256 CUs")
