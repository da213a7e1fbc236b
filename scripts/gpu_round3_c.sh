mkdir -p gpurun_out/prof512 gpurun_out/prof
export TMPDIR=/tmp
# rocprofv3 kernel trace of the 512-ray C4-shard bench and of the default C2 bench (no PMC in the same run)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof512 -o bench512 -- python bench.py --rays-per-gpu 512 --steps 100 --warmup 10 --no-cpu-baseline --no-extra > gpurun_out/prof512/bench_under_rocprof.log 2>&1
echo "rocprof rc=$?" >> gpurun_out/prof512/bench_under_rocprof.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra > gpurun_out/prof/bench_under_rocprof.log 2>&1
echo "rocprof rc=$?" >> gpurun_out/prof/bench_under_rocprof.log
rm -f gpurun_out/prof512/*.db gpurun_out/prof/*.db
head -32 gpurun_out/prof512/bench512_kernel_stats.csv | cut -c1-220
# PMC passes (merged launches) + summary
bash scripts/gpu_pmc.sh 4096 > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/pmc gpurun_out/pmc_summary
# the live --pmc mode of bench.py
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --pmc > gpurun_out/bench_pmc.log 2> gpurun_out/bench_pmc.err; echo "rc=$?" >> gpurun_out/bench_pmc.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_pmc.log').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_source'][:80], d['roofline'].get('pmc'))"
tail -3 gpurun_out/bench_pmc.err
