mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_training_parity.py -m gpu -q --timeout=2000 --tb=short -p no:cacheprovider -rA -s > gpurun_out/r4/teacher.log 2>&1; echo "teacher rc=$?" | tee -a gpurun_out/r4/teacher.log
grep -E "passed|failed" gpurun_out/r4/teacher.log | tail -3
grep -E "step=|Error|assert" gpurun_out/r4/teacher.log | cut -c1-1800 | tail -20
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_c.json 2> gpurun_out/r4/bench_c.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4/bench_c.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['route'], d['extra'].get('launches_per_step'), {k:d['cpu_baseline'][k] for k in ('value','cores','value_all_physical_cores','value_32_threads','physical_cores')})
P
