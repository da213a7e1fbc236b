#!/usr/bin/env python3
"""Condenses the rocprofv3 PMC passes of scripts/gpu_pmc.sh (gpurun_out/pmc/pass*_counter_collection.csv) into the
per-kernel / per-launch-size averages kept under profiles/<dir>/: pass<i>_summary.csv (kernel, grid_threads, counter,
avg_per_launch, launches, avg_duration_us) and summary.txt.   usage: python scripts/pmc_summary.py <in_dir> <out_dir>

Derived figures (MI355X_MICROARCH.md): MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs);
FETCH_SIZE / WRITE_SIZE are in KB, FETCH_SIZE x2 on gfx950; L2 hit = TCC_HIT / (TCC_HIT + TCC_MISS)."""
import collections
import csv
import glob
import os
import re
import sys

SHORT = (("mlp_fwd_bfs_k<8, 3, true>", "fwd_train_bf3"), ("mlp_fwd_bfs_k<8, 1", "mlp_fwd_bf1"), ("mlp_fwd_bfs_k<8, 2", "mlp_fwd_bf2"),
         ("mlp_fwd_bfs_k<8, 3", "mlp_fwd_bf3"), ("mlp_dgrad_bfs_k", "dgrad_bf3"), ("wgrad_mixed_k", "wgrad_bf3"),
         ("mlp_fwd_bf_k<8, 1>", "mlp_fwd_bf1"), ("mlp_fwd_bf_k<8, 2>", "mlp_fwd_bf2"), ("mlp_fwd_bf_k<8, 3>", "mlp_fwd_bf3"),
         ("mlp_fwd_k<8, true, true", "mlp_fwd_train"), ("mlp_fwd_k<8, true, false", "mlp_fwd_inf"), ("mlp_fwd_k", "mlp_fwd"),
         ("mlp_dgrad_k", "mlp_dgrad"), ("wgrad_reduce_k", "wgrad_reduce"), ("wgrad_k", "wgrad"))


def short(name):
    for pat, s in SHORT:
        if pat in name:
            return s
    return None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    allv = {}
    by_tag = collections.defaultdict(list)          # pass<i>a (kbench_pair.py) and pass<i>b (kbench.py) -> one summary per pass
    for path in sorted(glob.glob(os.path.join(src, "pass*_counter_collection.csv"))):
        by_tag[re.match(r"(pass\d+)[ab]?_", os.path.basename(path)).group(1)].append(path)
    for tag, paths in sorted(by_tag.items()):
        acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
        seen = set()
        for r in (row for path in paths for row in csv.DictReader(open(path))):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            key = (k, int(r["Grid_Size"]), r["Counter_Name"])
            a = acc[key]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            seen.add(r["Dispatch_Id"])
        with open(os.path.join(dst, tag + "_summary.csv"), "w") as f:
            f.write("kernel,grid_threads,counter,avg_per_launch,launches,avg_duration_us\n")
            for (k, g, c), (v, n, us) in sorted(acc.items()):
                f.write(f"{k},{g},{c},{v / n:.6g},{n},{us / n:.1f}\n")
                allv[(k, g, c)] = (v / n, us / n)
    lines = []
    for k, g in sorted({(k, g) for (k, g, _) in allv}):
        get = lambda c: allv.get((k, g, c), (0.0, 0.0))[0]   # noqa: E731
        cyc = get("GRBM_GUI_ACTIVE") / 8.0
        busy = 100.0 * get("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024) if cyc else 0.0
        hit, miss = get("TCC_HIT_sum"), get("TCC_MISS_sum")
        us = allv.get((k, g, "GRBM_GUI_ACTIVE"), (0.0, 0.0))[1]
        lines.append(f"{k:14s} grid={g:8d} {us:8.1f} us under collection  cycles/XCD={cyc:.4g}  MFMA busy {busy:5.1f}%  "
                     f"FETCH_SIZE {get('FETCH_SIZE'):.4g} KB (x2 on gfx950)  WRITE_SIZE {get('WRITE_SIZE'):.4g} KB  "
                     f"L2 hit {100.0 * hit / (hit + miss) if hit + miss else 0.0:.1f}%  LDS conflicts "
                     f"{get('SQ_LDS_BANK_CONFLICT'):.3g} / idx active {get('SQ_LDS_IDX_ACTIVE'):.3g}")
    open(os.path.join(dst, "summary.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
