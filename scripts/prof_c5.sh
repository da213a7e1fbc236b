# rocprofv3 kernel trace of the C5 leg alone (render_path over spiral poses, 756x1008 NDC frames)   usage: bash scripts/prof_c5.sh
mkdir -p gpurun_out/prof_c5
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -o c5 -- python -c "
import sys, json, torch; sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import bench
d = bench.c5_leg(torch.device('cuda:0'), n_frames=4)
d.pop('opt_in_reduced_precision', None)
print(json.dumps(d))
" > gpurun_out/prof_c5/c5_under_rocprof.log 2>&1
echo "rocprof rc=$?" >> gpurun_out/prof_c5/c5_under_rocprof.log
rm -f gpurun_out/prof_c5/*.db
head -30 gpurun_out/prof_c5/c5_kernel_stats.csv
