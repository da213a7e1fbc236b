mkdir -p gpurun_out/psnr
timeout 1500 python scripts/psnr_parity.py chaos --seeds $(seq 0 31) --steps 600 --milestones 25 50 100 150 300 --out gpurun_out/psnr/r03_psnr_chaos.json > gpurun_out/psnr/chaos_small.log 2>&1; echo "rc=$?" >> gpurun_out/psnr/chaos_small.log
tail -2 gpurun_out/psnr/chaos_small.log | cut -c1-1200
timeout 1500 python scripts/psnr_parity.py chaos --size c2 --seeds 0 1 2 3 4 5 6 7 --steps 150 --milestones 25 50 100 --out gpurun_out/psnr/r03_psnr_chaos_c2.json > gpurun_out/psnr/chaos_c2.log 2>&1; echo "rc=$?" >> gpurun_out/psnr/chaos_c2.log
tail -2 gpurun_out/psnr/chaos_c2.log | cut -c1-1200
