# Round-5 evidence at HEAD: fp32 GPU suite, smoke, default bench, rocprofv3 kernel trace of the bench command, C3 kernel trace
mkdir -p gpurun_out/r05head
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=1400 --tb=short -p no:cacheprovider -rA > gpurun_out/r05head/test_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05head/test_gpu.log
grep -E "passed|failed|rc=" gpurun_out/r05head/test_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05head/smoke.log 2>&1; echo "smoke rc=$?"
timeout 1200 python bench.py > gpurun_out/r05head/bench.json 2> gpurun_out/r05head/bench.err; echo "bench rc=$?"
bash scripts/gpu_prof.sh 5
bash scripts/prof_c3.sh
python -c "
import json
d=json.loads(open('gpurun_out/r05head/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], {k:v for k,v in d['config'].items() if not isinstance(v,(dict,list))})"
