# `python bench.py --gpus N` launched PLAINLY (no torchrun) on a 1-GPU box: the command re-executes itself under torch.distributed.run;
# the ranks share the GPU and exchange through gloo (CNERF_DIST_BACKEND) — the N > 1 code paths end to end, NOT a scaling measurement.
mkdir -p gpurun_out
export CNERF_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2
for ARGS in "--gpus 8 --scaling strong" "--gpus 8" "--gpus 2 --scaling strong --graph"; do
  tag=$(echo "$ARGS" | tr -d ' -')
  timeout 1500 python bench.py $ARGS --steps 8 --warmup 2 --no-extra --no-cpu-baseline --pmc off > gpurun_out/self_$tag.log 2> gpurun_out/self_$tag.err; rc=$?
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/self_$tag.log").read().strip().splitlines()[-1])
    print("$ARGS rc=$rc ->", d.get("scaling"), d.get("n_gpus"), d.get("ms_per_step"), d.get("config", {}).get("rays_per_gpu"), d.get("config", {}).get("parallelism"), d.get("hip_graph"), d.get("dist"), d.get("error"))
except Exception as e:
    print("$ARGS rc=$rc ERR", e); print(open("gpurun_out/self_$tag.err").read()[-2500:])
PY
done
