# Extension D16 of scripts/psnr_parity.py (pre-registered in its docstring): 16 seeds, the oracle's code on stock ATen GPU kernels + its
# one-ulp twin as the independent implementation, 6 HIP draws per seed; evaluated for the fp32 HIP path and the opt-in bf16x3 arithmetic.
mkdir -p gpurun_out
O=gpurun_out/psnr_oracle_aten_gpu_d16.npz
timeout 2400 python scripts/psnr_parity.py oracle_aten_gpu --seeds 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 --steps 150 --out $O > gpurun_out/psnr_d16_oracle.log 2>&1; echo "oracle rc=$?"
timeout 2400 python scripts/psnr_parity.py twins --oracle $O --draws 6 --size c2 --out gpurun_out/psnr_d16_fp32.json > gpurun_out/psnr_d16_fp32.log 2>&1; echo "fp32 rc=$?"
CNERF_TRAIN_PRECISION=bf16x3 timeout 2400 python scripts/psnr_parity.py twins --oracle $O --draws 6 --size c2 --out gpurun_out/psnr_d16_bf16x3.json > gpurun_out/psnr_d16_bf16x3.log 2>&1; echo "bf16x3 rc=$?"
python - <<'P'
import numpy as np, json
d = np.load("gpurun_out/psnr_oracle_aten_gpu_d16.npz")
np.savez_compressed("gpurun_out/psnr_oracle_aten_gpu_d16_noimg.npz", **{k: d[k] for k in d.files if not k.endswith("_img")})
for t in ("fp32", "bf16x3"):
    j = json.load(open(f"gpurun_out/psnr_d16_{t}.json"))
    print(t, "T_bias B", np.round(j["T_bias"]["B_dB"], 3).tolist(), "p", np.round(j["T_bias"]["p_two_sided"], 4).tolist(),
          "| T_dist D", np.round(j["T_dist"]["D_dB"], 3).tolist(), "p", np.round(j["T_dist"]["p_one_sided"], 4).tolist())
P
rm -rf gpurun_out/psnr_oracle_aten_gpu_d16.npz gpurun_out/psnr_oracle_aten_gpu_d16.npz.parts
