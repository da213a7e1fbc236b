# bf16x3 second line, round 5: the two 256 x 63 wgrad GEMMs (2 x 2 tiles per wave) on the bf16x3 body (CNERF_BF3_NARROW=1, default)
# against round 4's exact-fp32 body for them (=0): parity suite in this arithmetic, then the leg timed both ways
mkdir -p gpurun_out/bf3n; export TMPDIR=/tmp
CNERF_TRAIN_PRECISION=bf16x3 timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --tb=short -p no:cacheprovider --deselect tests/test_gpu_training_parity.py > gpurun_out/bf3n/test_gpu_bf16x3.log 2>&1; echo "suite bf16x3 rc=$?"
grep -E "passed|failed" gpurun_out/bf3n/test_gpu_bf16x3.log | tail -2; grep -E "^FAILED|Error" gpurun_out/bf3n/test_gpu_bf16x3.log | head -10
for v in 0 1 0 1; do
CNERF_BF3_NARROW=$v timeout 600 python - <<'P' 2>> gpurun_out/bf3n/leg.err | tee -a gpurun_out/bf3n/leg.txt
import os, sys, json, torch; sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import bench
o = bench.bf16x3_leg(torch.device('cuda:0'), 0, 1, 4096, 25.9, steps=40, warmup=10)
print("CNERF_BF3_NARROW=" + os.environ["CNERF_BF3_NARROW"], o["ms_per_step"], [(k["kernel"], k["points"], k["avg_ms"]) for k in o["roofline"]["kernels"]])
P
done
