# round 4, call A: the new tests first (teacher-forced C2 steps, bench self-launch), then the whole GPU suite
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_training_parity.py -m gpu -q --timeout=2000 --tb=short -p no:cacheprovider -rA -s > gpurun_out/r4/teacher.log 2>&1; echo "teacher rc=$?" | tee -a gpurun_out/r4/teacher.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=900 --tb=short -p no:cacheprovider -rA -k "bench_gpus_2" > gpurun_out/r4/launch.log 2>&1; echo "launch rc=$?" | tee -a gpurun_out/r4/launch.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --tb=short -p no:cacheprovider -rA --deselect tests/test_gpu_training_parity.py > gpurun_out/r4/test_gpu.log 2>&1; echo "suite rc=$?" | tee -a gpurun_out/r4/test_gpu.log
grep -E "passed|failed" gpurun_out/r4/*.log | tail -5
grep -E "step=|Error|assert" gpurun_out/r4/teacher.log | tail -20
nproc; free -g | head -2
