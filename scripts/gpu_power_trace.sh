# Package power / core clock while the C2 training step loops (rocm-smi sampled twice a second): the exact-fp32 step and the opt-in
# bf16x3 step.  Supports DESIGN 4 findings 1, 6, 7 (the step is power-limited; the clock is what gives).
mkdir -p gpurun_out
sample() {  # $1 = tag, $2 = pid to follow
  : > gpurun_out/power_$1.txt
  while kill -0 $2 2>/dev/null; do
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|Temperature \(Sensor junction\)" >> gpurun_out/power_$1.txt
    echo "--" >> gpurun_out/power_$1.txt
    sleep 0.5
  done
}
for mode in fp32 bf16x3; do
  if [ $mode = fp32 ]; then
    python bench.py --steps 1200 --warmup 20 --no-cpu-baseline --no-extra --pmc off > gpurun_out/power_bench_$mode.json 2>/dev/null &
  else
    python -c "
import sys, json, torch; sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import bench
print(json.dumps(bench.bf16x3_leg(torch.device('cuda:0'), 0, 1, 4096, 26.0, steps=1700, warmup=20)))" > gpurun_out/power_bench_$mode.json 2>/dev/null &
  fi
  pid=$!
  sleep 12   # (import + setup)
  sample $mode $pid
  wait $pid
done
python - <<'P'
import re, json
for mode in ("fp32", "bf16x3"):
    txt = open(f"gpurun_out/power_{mode}.txt").read()
    pw = [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", txt)]
    ck = [float(x) for x in re.findall(r"\((\d+)Mhz\)", txt)]
    tj = [float(x) for x in re.findall(r"junction\) \(C\): ([0-9.]+)", txt)]
    d = json.loads(open(f"gpurun_out/power_bench_{mode}.json").read().strip().splitlines()[-1])
    mid = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
    print(f"{mode:7s} step {d['ms_per_step']:.2f} ms | samples {len(pw)} | power median {mid(pw):.0f} W (max {max(pw) if pw else 0:.0f}) | sclk median {mid(ck):.0f} MHz (min {min(ck) if ck else 0:.0f}, max {max(ck) if ck else 0:.0f}) | junction {mid(tj):.0f} C")
P
head -12 gpurun_out/power_fp32.txt
