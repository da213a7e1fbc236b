# end-of-round evidence: GPU suite, default bench, bench --pmc, rocprofv3 of the bench / the C4 shard / the C5 leg, PMC passes
mkdir -p gpurun_out/final gpurun_out/prof gpurun_out/prof512
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --tb=short -p no:cacheprovider -rA > gpurun_out/final/test_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/test_gpu.log
grep -E "passed|failed|rc=" gpurun_out/final/test_gpu.log | tail -2
timeout 900 python bench.py --steps 40 --warmup 10 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo "bench rc=$?"
timeout 900 python bench.py --steps 40 --warmup 10 --no-extra --no-cpu-baseline --pmc > gpurun_out/final/bench_pmc.json 2> gpurun_out/final/bench_pmc.err; echo "bench --pmc rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra > gpurun_out/prof/bench_under_rocprof.log 2>&1; echo "rocprof rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof512 -o bench512 -- python bench.py --rays-per-gpu 512 --steps 200 --warmup 20 --no-cpu-baseline --no-extra > gpurun_out/prof512/bench_under_rocprof.log 2>&1; echo "rocprof512 rc=$?"
bash scripts/prof_c5.sh > /dev/null 2>&1
rm -f gpurun_out/prof/*.db gpurun_out/prof512/*.db
rm -rf gpurun_out/pmc gpurun_out/pmc_summary
bash scripts/gpu_pmc.sh 4096 > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/pmc gpurun_out/pmc_summary | grep -E "wgrad|dgrad|fwd_train|fwd_inf"
python -c "
import json
d=json.loads(open('gpurun_out/final/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'])
d=json.loads(open('gpurun_out/final/bench_pmc.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_source'][:60])"
