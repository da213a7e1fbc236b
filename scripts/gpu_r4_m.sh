mkdir -p gpurun_out/r4
export TMPDIR=/tmp KBENCH_LEVELS=192
{
echo "== product"; python scripts/kbench.py 4096 5 2>&1 | grep -E "dgrad bf16x3|x3 \(inference|TRAINING"
for so in variants/libcnerf_ablr_*.so; do
  echo "== $(basename $so)"; CNERF_LIB_PATH=$PWD/$so python scripts/kbench.py 4096 5 2>&1 | grep -E "dgrad bf16x3|x3 \(inference|TRAINING"
done
} | cut -c1-170 | tee gpurun_out/r4/ring_bf3_ablation.log
