mkdir -p gpurun_out/r4
export TMPDIR=/tmp
CNERF_TRAIN_PRECISION=bf16x3 timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --tb=line -p no:cacheprovider --deselect tests/test_gpu_training_parity.py > gpurun_out/r4/test_gpu_bf3.log 2>&1; echo "suite bf16x3 rc=$?"; grep -E "passed|failed" gpurun_out/r4/test_gpu_bf3.log | tail -3; grep -E "^FAILED" gpurun_out/r4/test_gpu_bf3.log | head
timeout 900 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --pmc off > gpurun_out/r4/bench_k.json 2> gpurun_out/r4/bench_k.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4/bench_k.json').read().strip().splitlines()[-1])
print(d['ms_per_step']); b=d['extra']['c2_bf16x3']; print({k:v for k,v in b.items() if k!='roofline'}); print(b.get('roofline',{}).get('frac')); [print(r) for r in b.get('roofline',{}).get('kernels',[])]
P
