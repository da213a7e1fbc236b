# usage: bash scripts/gpu_tests.sh   (on the GPU box via gpurun)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 --tb=short -p no:cacheprovider -rA > gpurun_out/test_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/test_gpu.log
grep -E "passed|failed|rc=" gpurun_out/test_gpu.log | tail -3
