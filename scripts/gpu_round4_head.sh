# HEAD-of-round verification after the container was re-created: fp32 GPU suite, smoke, default bench, C3 kernel trace
mkdir -p gpurun_out/head
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=1400 --tb=short -p no:cacheprovider -rA > gpurun_out/head/test_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/head/test_gpu.log
grep -E "passed|failed|rc=" gpurun_out/head/test_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/head/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1200 python bench.py > gpurun_out/head/bench.json 2> gpurun_out/head/bench.err; echo "bench rc=$?"
bash scripts/prof_c3.sh
python -c "
import json
d=json.loads(open('gpurun_out/head/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['extra']['c2_bf16x3']['ms_per_step'], d['extra']['c4_shard']['ms_per_step_graph'], d['extra']['c3']['ms_per_step'])"
