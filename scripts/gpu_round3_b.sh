mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --tb=short -p no:cacheprovider -rA > gpurun_out/test_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/test_gpu.log
grep -E "passed|failed|rc=|^FAILED" gpurun_out/test_gpu.log | tail -5
bash scripts/prof_c5.sh
tail -3 gpurun_out/prof_c5/c5_under_rocprof.log | cut -c1-1500
