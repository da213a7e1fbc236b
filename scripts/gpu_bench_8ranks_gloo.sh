# world size 8 on a 1-GPU box: eight ranks share the GPU, collectives through gloo — the C4 code path (4096-ray batch sharded 8 ways,
# 512 rays per rank, one 4.77 MB message per step, replicated Adam) end to end.  NOT a scaling measurement.
mkdir -p gpurun_out
export CNERF_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29660 bench.py --gpus 8 --steps 10 --warmup 3 --scaling strong --no-extra > gpurun_out/bench8_strong.log 2> gpurun_out/bench8_strong.err
echo "rc=$?" >> gpurun_out/bench8_strong.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench8_strong.log").read().strip().splitlines()[-1])
    print("world 8 strong:", d["scaling"], d["n_gpus"], d["ms_per_step"], d["config"]["rays_per_gpu"], d["config"]["global_batch"], d["config"]["parallelism"], d["dist"], d["config"]["final_loss"])
except Exception as e:
    print("ERR", e); print(open("gpurun_out/bench8_strong.err").read()[-3000:])
PY
