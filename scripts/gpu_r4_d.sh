mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 600 python scripts/kbench.py 4096 5 > gpurun_out/r4/kbench_d.log 2>&1; echo "kbench rc=$?"; grep -E "TRAINING|^S=|pair" gpurun_out/r4/kbench_d.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=900 --tb=short -p no:cacheprovider -rA -s -k "trained_network" > gpurun_out/r4/trained.log 2>&1; echo "trained rc=$?"; grep -E "max\|d\||PSNR|passed|failed" gpurun_out/r4/trained.log | tail -20
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --tb=short -p no:cacheprovider -rA --deselect tests/test_gpu_training_parity.py > gpurun_out/r4/test_gpu.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" gpurun_out/r4/test_gpu.log | tail -3; grep -E "^FAILED" gpurun_out/r4/test_gpu.log | head
CNERF_TRAIN_PRECISION=bf16x3 timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --tb=line -p no:cacheprovider -rA --deselect tests/test_gpu_training_parity.py > gpurun_out/r4/test_gpu_bf3.log 2>&1; echo "suite bf16x3 rc=$?"; grep -E "passed|failed" gpurun_out/r4/test_gpu_bf3.log | tail -3; grep -E "^FAILED" gpurun_out/r4/test_gpu_bf3.log | head -40
timeout 2400 python -m pytest tests/test_gpu_training_parity.py -m gpu -q --timeout=2000 --tb=short -p no:cacheprovider -rA -s > gpurun_out/r4/teacher.log 2>&1; echo "teacher rc=$?" | tee -a gpurun_out/r4/teacher.log
grep -E "passed|failed" gpurun_out/r4/teacher.log | tail -3
grep -E "Error|assert" gpurun_out/r4/teacher.log | cut -c1-600 | tail -8
