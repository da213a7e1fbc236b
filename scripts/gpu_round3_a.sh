# round 3, first GPU pass: the whole GPU suite, then the bench at the C2 batch, at the 512-ray C4 shard (eager / graph), and
# the graph forms with a 1-rank RCCL group
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --tb=short -p no:cacheprovider -rA > gpurun_out/test_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/test_gpu.log
grep -E "passed|failed|rc=" gpurun_out/test_gpu.log | tail -3
timeout 900 python bench.py --steps 40 --warmup 10 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
tail -c 600 gpurun_out/bench.log; tail -3 gpurun_out/bench.err
timeout 600 python bench.py --rays-per-gpu 512 --steps 200 --warmup 20 --no-extra --no-cpu-baseline > gpurun_out/bench_512.log 2> gpurun_out/bench_512.err; echo "rc=$?" >> gpurun_out/bench_512.err
timeout 600 python bench.py --rays-per-gpu 512 --steps 200 --warmup 20 --no-extra --no-cpu-baseline --graph > gpurun_out/bench_512_graph.log 2> gpurun_out/bench_512_graph.err; echo "rc=$?" >> gpurun_out/bench_512_graph.err
for C in split capture; do
CNERF_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout 600 python bench.py --rays-per-gpu 512 --steps 200 --warmup 20 --no-extra --no-cpu-baseline --graph --graph-collective $C > gpurun_out/bench_512_graph_rccl1_$C.log 2> gpurun_out/bench_512_graph_rccl1_$C.err; echo "rc=$?" >> gpurun_out/bench_512_graph_rccl1_$C.err
done
for f in bench_512 bench_512_graph bench_512_graph_rccl1_split bench_512_graph_rccl1_capture; do echo "== $f"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$f.log").read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["value"], [(k["kernel"],k["points"],k["avg_ms"],k["tflops"]) for k in d["roofline"]["kernels"]], d.get("dist"))
except Exception as e:
    print("ERR", e); print(open("gpurun_out/$f.err").read()[-1500:])
PY
done
