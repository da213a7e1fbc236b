# a15 one-call leg after the coarse-detach fix, then criterion D2000 phase 2 (fp32 twins) and phase 3 (bf16x3 twins)
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout=500 --tb=short -p no:cacheprovider -k "ss_step_loss or in_loop or c3_teacher or graphed_sharded" > gpurun_out/r05_a15c_tests.log 2>&1; echo "a15c pytest rc=$?"
grep -E "passed|failed" gpurun_out/r05_a15c_tests.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r05_a15c_tests.log | head
timeout 400 python - > gpurun_out/r05_a15c_leg.json 2> gpurun_out/r05_a15c_leg.err <<'P'
import sys, json, torch; sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import bench
o = bench.c3_ss_leg(torch.device('cuda:0'))
print("JSON" + json.dumps(o))
P
python - <<'P'
import json
s = open('gpurun_out/r05_a15c_leg.json').read()
d = json.loads(s[s.index('JSON{') + 4:])
print('c3_ss', d['ms_per_step'], d['ms_per_step_reference_lines'], d['roofline']['frac'], d['ray_samples_per_step_avg'], d['launches_per_step']['total'], d['launches_per_step']['own'])
P
bash scripts/gpu_psnr_d2000.sh twins
bash scripts/gpu_psnr_d2000.sh twins_bf16x3
