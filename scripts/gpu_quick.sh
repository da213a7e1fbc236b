# quick GPU check between edits: both GPU suites (fp32 and the opt-in bf16x3 arithmetic, without the long teacher-forced test) and a short bench line
mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --tb=short -p no:cacheprovider --deselect tests/test_gpu_training_parity.py > gpurun_out/r4/test_gpu.log 2>&1; echo "suite fp32 rc=$?"; grep -E "passed|failed" gpurun_out/r4/test_gpu.log | tail -2; grep -E "^FAILED|Error" gpurun_out/r4/test_gpu.log | head -10
CNERF_TRAIN_PRECISION=bf16x3 timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --tb=line -p no:cacheprovider --deselect tests/test_gpu_training_parity.py 2>&1 | tail -2
timeout 900 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r4/bench_t.json 2> gpurun_out/r4/bench_t.err; echo "bench rc=$?"; tail -3 gpurun_out/r4/bench_t.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4/bench_t.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac']); l=d['extra']['launches_per_step']; print(l['total'], l['own'], l['kernels'])
c=d['extra']['c4_shard']; print(c['ms_per_step_eager'], c['ms_per_step_graph'], c['frac_of_peak_graph'], c['host_enqueue_ms_per_eager_step']); print(d['extra']['c2_bf16x3']['ms_per_step'])
P
