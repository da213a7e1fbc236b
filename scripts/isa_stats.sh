#!/bin/bash
# Device-ISA statistics of the MFMA kernels (registers, spills, scratch, MFMA / scratch instruction counts):
#   scripts/isa_stats.sh [out_dir]   -> <out_dir>/<file>.s + a table on stdout (hipcc cross-compiles; no GPU needed)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/isa}
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$ROOT/include -I$ROOT/consistentnerf_amd/csrc -Wno-unused-result $CN_EXTRA_FLAGS"
for f in ${FILES:-mlp_fwd mlp_bwd wgrad}; do
  /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S "$ROOT/consistentnerf_amd/csrc/$f.hip" -o "$OUT/$f.s" &
done
wait
python3 - "$OUT" ${FILES:-mlp_fwd mlp_bwd wgrad} <<'PY'
import re, sys
out = sys.argv[1]
print(f"{'kernel':58s} vgpr agpr sgpr vspill sspill scratchB  mfma scratch_ins")
for f in sys.argv[2:]:
    s = open(f"{out}/{f}.s").read()
    # per-kernel bodies: from "<name>:" to ".Lfunc_end"
    meta = {}
    for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size", s, re.S):
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[name] = (g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"))
    for name, m in meta.items():
        body = re.search(rf"^{re.escape(name)}:(.*?)^\.Lfunc_end", s, re.S | re.M)
        b = body.group(1) if body else ""
        nm = len(re.findall(r"\bv_mfma", b)); ns = len(re.findall(r"\bscratch_", b))
        dem = name
        print(f"{dem[:58]:58s} {m[0]:4d} {m[1]:4d} {m[2]:4d} {m[3]:6d} {m[4]:6d} {m[5]:8d} {nm:5d} {ns:6d}")
# where the remaining spill traffic sits: scratch instructions (VGPR spills) and lane moves (SGPR spills) by the number of
# MFMAs issued before them
print()
for f in sys.argv[2:]:
    s = open(f"{out}/{f}.s").read()
    for name in re.findall(r"\.name:\s+(\S+)", s):
        body = re.search(rf"^{re.escape(name)}:(.*?)^\.Lfunc_end", s, re.S | re.M)
        if not body:
            continue
        n, sites = 0, {}
        for line in body.group(1).split("\n"):
            t = line.strip().split()
            if not t:
                continue
            if t[0].startswith("v_mfma"):
                n += 1
            elif t[0].startswith("scratch_") or t[0] in ("v_readlane_b32", "v_writelane_b32"):
                sites[(n, t[0])] = sites.get((n, t[0]), 0) + 1
        if sites:
            print(f"{name} ({n} MFMAs)")
            for (k, ins), c in sorted(sites.items()):
                print(f"   mfma# {k:5d}  {c:3d} x {ins}")
PY
