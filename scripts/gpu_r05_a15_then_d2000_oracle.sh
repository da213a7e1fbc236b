# quick check of the a15 one-launch form, then phase 1 of criterion D2000 (the long one)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout=800 --tb=short -p no:cacheprovider -k "ss_ref_rays or in_loop or ss_primary or warp or c3" > gpurun_out/r05_a15_tests.log 2>&1; echo "a15 pytest rc=$?"
tail -5 gpurun_out/r05_a15_tests.log
timeout 300 python - > gpurun_out/r05_a15_leg.json 2> gpurun_out/r05_a15_leg.err <<'P'
import sys, json, torch; sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import bench
o = bench.c3_ss_leg(torch.device('cuda:0'))
print(json.dumps(o))
P
python -c "
import json; d=json.load(open('gpurun_out/r05_a15_leg.json')); print('c3_ss', d['ms_per_step'], d['roofline']['frac'], d['launches_per_step'])" 2>&1 | cut -c1-1500
bash scripts/gpu_psnr_d2000.sh oracle
