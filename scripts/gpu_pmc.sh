# PMC passes over the kernel micro-benches (each --pmc set is its own run; kernel-trace only, as gpurun requires):
# scripts/kbench_pair.py = the training step's launches (forward of both levels, merged dgrad / wgrad of BOTH networks),
# scripts/kbench.py      = the single-network launches of the fine level incl. the inference forward (one level only: the
#                          summaries are keyed by grid size, and wgrad's grid no longer depends on the level).     usage: bash scripts/gpu_pmc.sh [B]
B=${1:-4096}
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
i=0
for SET in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS"; do
  i=$((i+1))
  # (one rocprofv3 run per script: two children of one run write the same output file)
  timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/pmc -o pass${i}a -- python scripts/kbench_pair.py $B 2 > gpurun_out/pmc/pass${i}a.log 2>&1
  ra=$?
  KBENCH_LEVELS=192 timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/pmc -o pass${i}b -- python scripts/kbench.py $B 2 > gpurun_out/pmc/pass${i}b.log 2>&1
  echo "pass$i rc=$ra/$? : $SET" >> gpurun_out/pmc/passes.txt
done
rm -f gpurun_out/pmc/*.db
ls gpurun_out/pmc
