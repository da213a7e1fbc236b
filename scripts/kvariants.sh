#!/bin/bash
# GPU box: kernel micro-bench of the product library and of every experiment build under variants/
# (CN_BUILD_TAG=<tag> CN_EXTRA_FLAGS=... python -m consistentnerf_amd.build).  usage: scripts/kvariants.sh [tags...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== product"; python scripts/kbench.py 4096 5
for so in variants/libcnerf_*.so; do
  tag=$(basename $so .so); tag=${tag#libcnerf_}
  if [ $# -gt 0 ] && [[ ! " $* " =~ " $tag " ]]; then continue; fi
  echo "== $tag"
  if [[ $tag == *timing* ]]; then
    CNERF_LIB_PATH=$PWD/$so python scripts/ktiming.py 4096 192 1
    CNERF_LIB_PATH=$PWD/$so python scripts/ktiming.py 4096 192 0
  else
    CNERF_LIB_PATH=$PWD/$so python scripts/kbench.py 4096 5
  fi
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/kvariants.log
