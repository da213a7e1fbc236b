# smoke + bench (+ optional rocprof)   usage: bash scripts/gpu_bench.sh [steps] [warmup]
mkdir -p gpurun_out
STEPS=${1:-10}; WARM=${2:-3}
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
timeout 1200 python bench.py --steps $STEPS --warmup $WARM > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tail -2 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
# side benches: C5 frame + hard masks, C3 step with the consistency terms
if [ -n "$SIDE" ]; then
  timeout 600 python scripts/bench_render.py 2 > gpurun_out/bench_render.log 2>&1; tail -1 gpurun_out/bench_render.log
  timeout 600 python scripts/bench_c3.py 20 > gpurun_out/bench_c3.log 2>&1; tail -1 gpurun_out/bench_c3.log
fi
