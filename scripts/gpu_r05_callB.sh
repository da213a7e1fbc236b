# a15 one-call form (tests + leg), then the bf16x3 narrow-GEMM A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout=800 --tb=short -p no:cacheprovider -k "ss_step_loss or ss_ref_rays or in_loop or ss_primary or c3 or closs or masked" > gpurun_out/r05_a15b_tests.log 2>&1; echo "a15b pytest rc=$?"
grep -E "passed|failed" gpurun_out/r05_a15b_tests.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/r05_a15b_tests.log | head
timeout 400 python - > gpurun_out/r05_a15b_leg.json 2> gpurun_out/r05_a15b_leg.err <<'P'
import sys, json, torch; sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import bench
o = bench.c3_ss_leg(torch.device('cuda:0'))
print("JSON" + json.dumps(o))
P
python - <<'P'
import json
s = open('gpurun_out/r05_a15b_leg.json').read()
d = json.loads(s[s.index('JSON{') + 4:])
print('c3_ss', d['ms_per_step'], d['ms_per_step_reference_lines'], d['roofline']['frac'], d['final_loss'], d['launches_per_step'])
P
bash scripts/gpu_r05_bf3_narrow.sh
