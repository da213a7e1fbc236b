mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 900 python scripts/kbench.py 4096 5 > gpurun_out/r4/kbench_i.log 2>&1; echo "kbench rc=$?"; grep -E "TRAINING|dgrad bf16x3|wgrad bf16x3|^S=|pair|Error|error|x3 \(inference" gpurun_out/r4/kbench_i.log | cut -c1-330
CNERF_TRAIN_PRECISION=bf16x3 timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --tb=line -p no:cacheprovider -rA --deselect tests/test_gpu_training_parity.py > gpurun_out/r4/test_gpu_bf3.log 2>&1; echo "suite bf16x3 rc=$?"; grep -E "passed|failed" gpurun_out/r4/test_gpu_bf3.log | tail -3; grep -E "^FAILED" gpurun_out/r4/test_gpu_bf3.log | head -40
grep -n "Error\|assert" gpurun_out/r4/test_gpu_bf3.log | head -30 | cut -c1-300
timeout 900 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --pmc off > gpurun_out/r4/bench_i.json 2> gpurun_out/r4/bench_i.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4/bench_i.json').read().strip().splitlines()[-1])
print(d['ms_per_step']); b=d['extra']['c2_bf16x3']; print({k:v for k,v in b.items() if k!='roofline'}); print(b.get('roofline',{}).get('frac')); [print(r) for r in b.get('roofline',{}).get('kernels',[])]
print('c5', d['extra']['c5']['frame_s'], {k:v['frame_s'] for k,v in d['extra']['c5']['opt_in_reduced_precision'].items()})
P
CNERF_TRAIN_PRECISION=bf16x3 timeout 2400 python -m pytest tests/test_gpu_training_parity.py -m gpu -q --timeout=2000 --tb=short -p no:cacheprovider -rA -s > gpurun_out/r4/teacher_bf3.log 2>&1; echo "teacher bf16x3 rc=$?" | tee -a gpurun_out/r4/teacher_bf3.log
cp gpurun_out/teacher_forced_c2.json gpurun_out/r4/teacher_forced_c2_bf16x3.json
grep -E "passed|failed" gpurun_out/r4/teacher_bf3.log | tail -3
grep -E "Error|assert" gpurun_out/r4/teacher_bf3.log | cut -c1-600 | tail -8
grep -E "step=" gpurun_out/r4/teacher_bf3.log | sed -E 's/.*(step=[0-9]+).*(K_hip_worst=[^ ]+) (K_ref32_worst=[^ ]+).*/\1 \2 \3/' 
