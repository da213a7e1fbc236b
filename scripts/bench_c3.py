#!/usr/bin/env python3
"""BASELINE config C3 on the GPU box (side bench, not the headline metric): the legs of bench.py that time the ConsistentNeRF
training step — `c3` (hard masks, masked rgb + depth losses on both levels, monocular patch term, clip 0.1 + Adam; through the
one-call surface and as the reference's lines) and `c3_ss` (the in-loop consistency step, VT:895-972) — on their own.
usage: python scripts/bench_c3.py [steps]            (prints one JSON object per leg)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    print(json.dumps({"c3": bench.c3_leg(dev, steps)}))
    print(json.dumps({"c3_ss": bench.c3_ss_leg(dev, max(steps // 2, 3))}))


if __name__ == "__main__":
    main()
