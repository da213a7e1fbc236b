mkdir -p gpurun_out/r4
export TMPDIR=/tmp
KBENCH_LEVELS=192 timeout 900 python scripts/kbench.py 4096 5 > gpurun_out/r4/kbench_l.log 2>&1; echo "kbench rc=$?"; grep -E "TRAINING|dgrad bf16x3|wgrad bf16x3|^S=|Error|error|x[123] \(inference" gpurun_out/r4/kbench_l.log | cut -c1-330
CNERF_TRAIN_PRECISION=bf16x3 timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --tb=line -p no:cacheprovider --deselect tests/test_gpu_training_parity.py > gpurun_out/r4/test_gpu_bf3.log 2>&1; echo "suite bf16x3 rc=$?"; grep -E "passed|failed" gpurun_out/r4/test_gpu_bf3.log | tail -3; grep -E "^FAILED" gpurun_out/r4/test_gpu_bf3.log | head
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --tb=line -p no:cacheprovider --deselect tests/test_gpu_training_parity.py > gpurun_out/r4/test_gpu.log 2>&1; echo "suite fp32 rc=$?"; grep -E "passed|failed" gpurun_out/r4/test_gpu.log | tail -3
timeout 900 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --pmc off > gpurun_out/r4/bench_l.json 2> gpurun_out/r4/bench_l.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4/bench_l.json').read().strip().splitlines()[-1])
print(d['ms_per_step']); b=d['extra']['c2_bf16x3']; print({k:v for k,v in b.items() if k!='roofline'}); [print(r) for r in b.get('roofline',{}).get('kernels',[])]
print('c5', d['extra']['c5']['frame_s'], {k:v['frame_s'] for k,v in d['extra']['c5']['opt_in_reduced_precision'].items()})
P
