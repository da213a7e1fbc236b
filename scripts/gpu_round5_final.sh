# end-of-round-5 evidence: GPU suites (fp32 default and the opt-in bf16x3 arithmetic), teacher-forced test, default bench, rocprofv3
# kernel traces of the bench / the C4 shard / the bf16x3 step, PMC passes
mkdir -p gpurun_out/final5 gpurun_out/prof_c3 gpurun_out/prof gpurun_out/prof512 gpurun_out/prof_bf3
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout=2000 --tb=short -p no:cacheprovider -rA > gpurun_out/final5/test_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final5/test_gpu.log
grep -E "passed|failed|rc=" gpurun_out/final5/test_gpu.log | tail -2
cp gpurun_out/teacher_forced_c2.json gpurun_out/final5/teacher_forced_c2.json; cp gpurun_out/teacher_forced_c3.json gpurun_out/final5/teacher_forced_c3.json
CNERF_TRAIN_PRECISION=bf16x3 timeout 2400 python -m pytest tests -m gpu -q --timeout=2000 --tb=short -p no:cacheprovider -rA > gpurun_out/final5/test_gpu_bf16x3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final5/test_gpu_bf16x3.log
grep -E "passed|failed|rc=" gpurun_out/final5/test_gpu_bf16x3.log | tail -2
cp gpurun_out/teacher_forced_c2.json gpurun_out/final5/teacher_forced_c2_bf16x3.json
timeout 1200 python bench.py > gpurun_out/final5/bench.json 2> gpurun_out/final5/bench.err; echo "bench rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra --pmc off > gpurun_out/prof/bench_under_rocprof.log 2>&1; echo "rocprof rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof512 -o bench512 -- python bench.py --rays-per-gpu 512 --steps 200 --warmup 20 --no-cpu-baseline --no-extra --pmc off > gpurun_out/prof512/bench_under_rocprof.log 2>&1; echo "rocprof512 rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bf3 -o bf3 -- python -c "
import sys, json, torch; sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import bench
print(json.dumps(bench.bf16x3_leg(torch.device('cuda:0'), 0, 1, 4096, 25.9)))
" > gpurun_out/prof_bf3/bf3_under_rocprof.log 2>&1; echo "rocprof bf3 rc=$?"
rm -f gpurun_out/prof/*.db gpurun_out/prof512/*.db gpurun_out/prof_bf3/*.db
bash scripts/prof_c3.sh
rm -rf gpurun_out/pmc gpurun_out/pmc_summary
bash scripts/gpu_pmc.sh 4096 > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/pmc gpurun_out/pmc_summary | grep -E "wgrad|dgrad|fwd_train|fwd_inf|bf3"
python -c "
import json
d=json.loads(open('gpurun_out/final5/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['extra']['c2_bf16x3']['ms_per_step'], d['extra']['c4_shard']['ms_per_step_graph'])"
