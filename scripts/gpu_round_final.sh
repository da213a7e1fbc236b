# End-of-round evidence in ONE gpurun call:   bash scripts/gpu_round_final.sh <tag>     (e.g. r06) -> gpurun_out/final_<tag>/
#   GPU suites (exact fp32 default, then the opt-in bf16x3 arithmetic), the teacher-forced JSONs, the default bench (stdout line +
#   side file), rocprofv3 --kernel-trace --stats of the same C2 step / of the 512-ray C4 shard / of the C3 and c3_ss legs, PMC passes.
TAG=${1:-rXX}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT gpurun_out/prof gpurun_out/prof512 gpurun_out/prof_c3 gpurun_out/prof_c3ss
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q --timeout=2400 --tb=short -p no:cacheprovider -rA > $OUT/test_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/test_gpu.log
grep -E "passed|failed|rc=" $OUT/test_gpu.log | tail -2
for f in teacher_forced_c2 teacher_forced_c3 teacher_forced_c3ss; do cp gpurun_out/$f.json $OUT/$f.json 2>/dev/null; done
if [ "${SKIP_BF3:-0}" != "1" ]; then
  CNERF_TRAIN_PRECISION=bf16x3 timeout 3000 python -m pytest tests -m gpu -q --timeout=2400 --tb=short -p no:cacheprovider -rA > $OUT/test_gpu_bf16x3.log 2>&1; echo "pytest rc=$?" >> $OUT/test_gpu_bf16x3.log
  grep -E "passed|failed|rc=" $OUT/test_gpu_bf16x3.log | tail -2
  cp gpurun_out/teacher_forced_c2.json $OUT/teacher_forced_c2_bf16x3.json 2>/dev/null
fi
timeout 1200 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$? line bytes=$(wc -c < $OUT/bench_line.json)"
cp gpurun_out/bench_detail.json $OUT/bench_detail.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra --pmc off > gpurun_out/prof/bench_under_rocprof.log 2>&1; echo "rocprof rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof512 -o bench512 -- python bench.py --rays-per-gpu 512 --steps 200 --warmup 20 --no-cpu-baseline --no-extra --pmc off > gpurun_out/prof512/bench_under_rocprof.log 2>&1; echo "rocprof512 rc=$?"
bash scripts/prof_c3.sh
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3ss -o c3ss -- python bench.py --only-leg c3_ss > gpurun_out/prof_c3ss/c3ss_under_rocprof.log 2>&1; echo "rocprof c3ss rc=$?"
rm -f gpurun_out/prof/*.db gpurun_out/prof512/*.db gpurun_out/prof_c3ss/*.db
rm -rf gpurun_out/pmc gpurun_out/pmc_summary
bash scripts/gpu_pmc.sh 4096 > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/pmc gpurun_out/pmc_summary | grep -E "wgrad|dgrad|fwd_train|fwd_inf|bf3"
python -c "
import json
d=json.loads(open('$OUT/bench_line.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], {k: v for k, v in d['config'].items() if k.startswith('leg_')})"
