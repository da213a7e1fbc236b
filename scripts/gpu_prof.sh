# rocprofv3 kernel trace of the bench command (no PMC in the same run)   usage: bash scripts/gpu_prof.sh [steps]
mkdir -p gpurun_out/prof
STEPS=${1:-5}
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/prof/bench_under_rocprof.log 2>&1
echo "rocprof rc=$?" >> gpurun_out/prof/bench_under_rocprof.log
rm -f gpurun_out/prof/*.db
ls -la gpurun_out/prof | head -20
