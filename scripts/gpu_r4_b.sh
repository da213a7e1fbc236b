mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_training_parity.py -m gpu -q --timeout=2000 --tb=short -p no:cacheprovider -rA -s > gpurun_out/r4/teacher.log 2>&1; echo "teacher rc=$?" | tee -a gpurun_out/r4/teacher.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=900 --tb=short -p no:cacheprovider -rA -s -k "mlp_golden or render_rays_golden" > gpurun_out/r4/tight.log 2>&1; echo "tight rc=$?" | tee -a gpurun_out/r4/tight.log
grep -E "passed|failed" gpurun_out/r4/teacher.log gpurun_out/r4/tight.log | tail -5
grep -E "step=|Error|assert" gpurun_out/r4/teacher.log | cut -c1-900 | tail -20
grep -E "ReLU pattern|rel-max|FAIL|Error" gpurun_out/r4/tight.log | sort -t' ' -k6 | tail -30
