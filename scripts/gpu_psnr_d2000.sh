# Criterion D2000 of scripts/psnr_parity.py (pre-registered in its docstring, "Round 5"): 4 seeds x 2 000 steps at the true C2 size.
#   phase 1 (bash scripts/gpu_psnr_d2000.sh oracle): the oracle's code on stock ATen GPU kernels + one-ulp twins -> gpurun_out/r05_psnr_oracle_aten_gpu_2000.npz
#   phase 3 (bash scripts/gpu_psnr_d2000.sh twins_bf16x3): the same with the HIP runs in the opt-in bf16x3 arithmetic (extension D2000-bf16x3)
#   phase 2 (bash scripts/gpu_psnr_d2000.sh twins):  7 HIP runs per seed against profiles/r05_psnr_oracle_aten_gpu_2000.npz -> gpurun_out/r05_psnr_parity_2000.json
mkdir -p gpurun_out
if [ "$1" = "oracle" ]; then
  O=gpurun_out/r05_psnr_oracle_full.npz
  timeout 3200 python scripts/psnr_parity.py oracle_aten_gpu --seeds 0 1 2 3 --steps 2000 --milestones 500 1000 2000 --out $O > gpurun_out/r05_psnr_d2000_oracle.log 2>&1; echo "oracle rc=$?"
  python - <<'P'
import numpy as np
d = np.load("gpurun_out/r05_psnr_oracle_full.npz")
np.savez_compressed("gpurun_out/r05_psnr_oracle_aten_gpu_2000.npz", **{k: d[k] for k in d.files if not k.endswith("_img")})
print({k: np.round(d[k], 3).tolist() for k in d.files if k.endswith("psnr_at")})
P
  rm -rf gpurun_out/r05_psnr_oracle_full.npz gpurun_out/r05_psnr_oracle_full.npz.parts
  tail -3 gpurun_out/r05_psnr_d2000_oracle.log
elif [ "$1" = "twins_bf16x3" ]; then
  CNERF_TRAIN_PRECISION=bf16x3 timeout 3200 python scripts/psnr_parity.py twins --oracle profiles/r05_psnr_oracle_aten_gpu_2000.npz --draws 6 --size c2 --no-criterion-a --out gpurun_out/r05_psnr_parity_2000_bf16x3.json > gpurun_out/r05_psnr_d2000_twins_bf16x3.log 2>&1; echo "twins bf16x3 rc=$?"
  tail -2 gpurun_out/r05_psnr_d2000_twins_bf16x3.log | cut -c1-1500
else
  timeout 3200 python scripts/psnr_parity.py twins --oracle profiles/r05_psnr_oracle_aten_gpu_2000.npz --draws 6 --size c2 --no-criterion-a --out gpurun_out/r05_psnr_parity_2000.json > gpurun_out/r05_psnr_d2000_twins.log 2>&1; echo "twins rc=$?"
  tail -2 gpurun_out/r05_psnr_d2000_twins.log | cut -c1-1500
fi
