mkdir -p gpurun_out
: > gpurun_out/wgrad_sweep.log
for B in 512 1024 2048 4096; do
  python scripts/kbench_pair.py $B 30 2>&1 | grep -v amdgpu.ids >> gpurun_out/wgrad_sweep.log
  for N in $@; do
    CNERF_WGRAD_NSPLIT=$N python scripts/kbench_pair.py $B 30 2>&1 | tail -1 >> gpurun_out/wgrad_sweep.log
  done
done
cat gpurun_out/wgrad_sweep.log
