# EXPERIMENT: dynamic work list in wgrad_k (-DCN_WGRAD_DYN, variants/libcnerf_wgdyn.so) against the product library
mkdir -p gpurun_out
L=gpurun_out/wgrad_dyn.log
: > $L
V=$PWD/variants/libcnerf_wgdyn.so
echo "== parity of the variant (backward tests)" >> $L
CNERF_LIB_PATH=$V timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "backward or wgrad or ragged_batch" --timeout=500 2>&1 | tail -3 >> $L
for B in 512 4096; do
  echo "== product B=$B" >> $L
  timeout 200 python scripts/kbench_pair.py $B 30 2>&1 | grep -v amdgpu.ids | tail -3 >> $L
  echo "== dyn B=$B" >> $L
  CNERF_LIB_PATH=$V timeout 200 python scripts/kbench_pair.py $B 30 2>&1 | grep -v amdgpu.ids | tail -3 >> $L
  echo "== dyn B=$B grid 512 (2 tickets chains per CU cannot co-reside: LDS; sanity)" >> $L
  for N in "96,48" "128,64"; do
    echo "-- product NSPLIT=$N" >> $L
    CNERF_WGRAD_NSPLIT=$N timeout 200 python scripts/kbench_pair.py $B 30 2>&1 | tail -1 >> $L
    echo "-- dyn NSPLIT=$N" >> $L
    CNERF_WGRAD_NSPLIT=$N CNERF_LIB_PATH=$V timeout 200 python scripts/kbench_pair.py $B 30 2>&1 | tail -1 >> $L
  done
done
echo "== product again B=512" >> $L
timeout 200 python scripts/kbench_pair.py 512 30 2>&1 | tail -1 >> $L
cat $L
