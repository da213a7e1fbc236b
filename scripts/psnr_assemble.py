#!/usr/bin/env python3
"""profiles/r03_psnr_parity.json = the three blocks VERDICT r02 item 4 asked for, assembled from the per-run files
(profiles/r03_psnr_chaos*.json, r03_psnr_curve*.json; criteria: scripts/psnr_parity.py docstring)."""
import json
import os

import numpy as np

P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
L = lambda n: json.load(open(os.path.join(P, n)))  # noqa: E731
chaos, curve = L("r03_psnr_chaos.json"), L("r03_psnr_curve.json")
chaos_c2, curve_c2 = L("r03_psnr_chaos_c2.json"), L("r03_psnr_curve_c2.json")
slim = lambda d: {k: v for k, v in d.items() if k != "runs"}  # noqa: E731


def paired(curve_d, chaos_d):
    """per seed and milestone: |HIP - oracle| next to |HIP+1ulp - HIP| of the SAME seed (heavy-tailed spreads: the pairing says
    more than an rms)"""
    twin = {r["seed"]: [abs(a - b) for a, b in zip(r["psnr_hip_1ulp"], r["psnr_hip"])] for r in chaos_d["runs"]}
    ms_c = chaos_d["milestones"]
    rows, wins, n = [], 0, 0
    for r in curve_d["runs"]:
        g = [abs(h - o) for h, o in zip(r["psnr_hip"], r["psnr_oracle"])]
        t = [twin[r["seed"]][ms_c.index(m)] if (r["seed"] in twin and m in ms_c) else None for m in curve_d["milestones"]]
        rows.append({"seed": r["seed"], "abs_gap_hip_vs_oracle_dB": [round(x, 4) for x in g],
                     "abs_diff_hip_vs_its_1ulp_twin_dB": [None if x is None else round(x, 4) for x in t]})
        for a, b in zip(g, t):
            if b is not None:
                n += 1
                wins += a <= b
    return {"milestones": curve_d["milestones"], "rows": rows, "gap_not_larger_than_the_twin_difference": f"{wins} of {n} (seed, milestone) pairs"}


r2 = np.load(os.path.join(P, "r02_psnr_oracle.npz"))
r3 = np.load(os.path.join(P, "r03_psnr_oracle.npz"))
thr = [float(r3[f"s{s}_ref_psnr"]) - float(r2[f"s{s}_ref_psnr"]) for s in range(8)]
out = {
    "definition": "held-out PSNR = -10 log10(mean((rgb - gt)^2)) over the held-out view (H:10, V:2047-2048); synthetic DTU-like 3-view scene, "
                  "coarse 64 + fine 64+128 samples, D=8/W=256, the reference's deterministic RNG hook on both sides, identical initial weights and "
                  "batches; HIP = this library on the MI355X, oracle = the CPU restatement of the reference (stock ATen fp32)",
    "criteria_fixed_before_the_runs": "scripts/psnr_parity.py docstring (A: |gap| <= 0.02 dB per seed while the loss curves track to 1e-3; "
                                      "B: |mean gap| <= 2 sigma_chaos / sqrt(n) at every milestone; C: C2-size |gap(150)| <= 2 sigma_chaos_c2(150))",
    "block_a_chaos_32_seeds_small": slim(chaos),
    "block_b_gap_vs_steps_8_seeds_small": dict(slim(curve), paired_with_the_hip_twin=paired(curve, chaos)),
    "block_c_true_c2_batch_size": {"chaos_32_seeds": slim(chaos_c2), "curve": slim(curve_c2), "paired_with_the_hip_twin": paired(curve_c2, chaos_c2),
                                   "criterion_C": {"abs_gap_at_150_dB": [abs(r["psnr_hip"][-1] - r["psnr_oracle"][-1]) for r in curve_c2["runs"]],
                                                   "limit_dB": 2.0 * chaos_c2["sigma_chaos_dB"][-1]}},
    "oracle_run_to_run": {"what": "the SAME oracle runs (seeds 0-7, 600 steps) in round 2 (2 ATen threads per job) and round 3 (1 thread): "
                                  "ATen's GEMM blocking changes with the thread count, the trajectories decorrelate — the reference arithmetic "
                                  "is not a single trajectory either", "psnr_600_r03_minus_r02_dB": [round(x, 3) for x in thr],
                          "rms_dB": float(np.sqrt(np.mean(np.square(thr))))},
}
out["block_c_true_c2_batch_size"]["criterion_C"]["pass"] = bool(all(g <= out["block_c_true_c2_batch_size"]["criterion_C"]["limit_dB"]
                                                                  for g in out["block_c_true_c2_batch_size"]["criterion_C"]["abs_gap_at_150_dB"]))
json.dump(out, open(os.path.join(P, "r03_psnr_parity.json"), "w"), indent=1)
print(json.dumps({"A": curve["criterion_A"]["pass"], "B": curve["criterion_B"]["pass"], "C": out["block_c_true_c2_batch_size"]["criterion_C"],
                  "paired_small": out["block_b_gap_vs_steps_8_seeds_small"]["paired_with_the_hip_twin"]["gap_not_larger_than_the_twin_difference"],
                  "paired_c2": out["block_c_true_c2_batch_size"]["paired_with_the_hip_twin"]["gap_not_larger_than_the_twin_difference"],
                  "B_c2": curve_c2.get("criterion_B", {}).get("pass"), "oracle_run_to_run_rms": out["oracle_run_to_run"]["rms_dB"]}))
