# the N>1 code path of bench.py on a 1-GPU box: two ranks share the GPU, collectives through gloo (NOT a scaling measurement)
mkdir -p gpurun_out
export CNERF_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=4
for MODE in "" "--scaling strong" "--scaling strong --graph"; do
  tag=$(echo "weak$MODE" | tr -d ' -')
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29650 bench.py --gpus 2 --steps 10 --warmup 3 $MODE > gpurun_out/bench2_$tag.log 2> gpurun_out/bench2_$tag.err
  echo "rc=$?" >> gpurun_out/bench2_$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench2_$tag.log").read().strip().splitlines()[-1])
    print("$tag", d["scaling"], d["ms_per_step"], d["config"]["rays_per_gpu"], d["config"]["parallelism"], d["hip_graph"], d["dist"]["backend"], d["dist"]["rccl_ranks"], list((d.get("extra") or {}).keys()))
    leg = (d.get("extra") or {}).get("c4_strong")
    if leg: print("   c4_strong:", {k: v for k, v in leg.items() if k != "kernels"})
except Exception as e:
    print("$tag ERR", e); print(open("gpurun_out/bench2_$tag.err").read()[-2000:])
PY
done
# the strong-scaling leg of an N>1 run incl. the RCCL all-reduce recorded inside the graph, on a 1-rank RCCL group
unset CNERF_DIST_BACKEND
CNERF_BENCH_FORCE_LEG=1 CNERF_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_rccl1_leg.log 2> gpurun_out/bench_rccl1_leg.err; echo "rc=$?" >> gpurun_out/bench_rccl1_leg.err
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_rccl1_leg.log").read().strip().splitlines()[-1])
print("rccl1 leg", d["ms_per_step"], {k: v for k, v in d["extra"]["c4_strong"].items() if k != "kernels"})
PY
tail -2 gpurun_out/bench_rccl1_leg.err
