mkdir -p gpurun_out/r4
export TMPDIR=/tmp KBENCH_LEVELS=192
{
echo "== product"; python scripts/kbench.py 4096 5 2>&1 | grep -E "wgrad bf16x3|^S="
for so in variants/libcnerf_abl_*.so; do
  echo "== $(basename $so)"; CNERF_LIB_PATH=$PWD/$so python scripts/kbench.py 4096 5 2>&1 | grep -E "wgrad bf16x3"
done
} | cut -c1-200 | tee gpurun_out/r4/wgrad_bf3_ablation.log
