# rocprofv3 kernel trace of the C3 leg alone (consistency-term training step)   usage: bash scripts/prof_c3.sh
mkdir -p gpurun_out/prof_c3
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3 -o c3 -- python -c "
import sys, json, torch; sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import bench
print(json.dumps(bench.c3_leg(torch.device('cuda:0'), steps=20)))
" > gpurun_out/prof_c3/c3_under_rocprof.log 2>&1
echo "rocprof rc=$?" >> gpurun_out/prof_c3/c3_under_rocprof.log
rm -f gpurun_out/prof_c3/*.db
