# the encodings' own sincos (csrc/sincos.hpp) against the library call (product build) — the variant -DCN_SINCOS_OWN: the whole GPU suite on the variant, then kernel
# timings of both builds interleaved, then the bench's headline step with each
mkdir -p gpurun_out/sincos; export TMPDIR=/tmp
cp /dev/null gpurun_out/sincos/tests.log
CNERF_LIB_PATH=$PWD/variants/libcnerf_own.so timeout 1500 python -m pytest tests -m gpu -q --timeout=1000 --tb=short -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_plain_c_program_uses_the_abi > gpurun_out/sincos/tests.log 2>&1; echo "sincos pytest rc=$?"
grep -E "passed|failed" gpurun_out/sincos/tests.log | tail -2; grep -E "^FAILED|^ERROR|max \|err\|" gpurun_out/sincos/tests.log | head
for i in 1 2; do
  echo "== product (library sincosf)"; python scripts/kbench.py 4096 10 2>&1 | grep -iE "fwd|forward" | head -8
  echo "== variant (own sincos)"; CNERF_LIB_PATH=$PWD/variants/libcnerf_own.so python scripts/kbench.py 4096 10 2>&1 | grep -iE "fwd|forward" | head -8
done 2>&1 | tee gpurun_out/sincos/kbench_ab.txt
for i in 1 2; do
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('libm   ', d['ms_per_step'], d['roofline']['frac'], [(k['kernel'],k['points'],k['avg_ms']) for k in d['roofline'].get('kernels',[])])"
  CNERF_LIB_PATH=$PWD/variants/libcnerf_own.so timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('own    ', d['ms_per_step'], d['roofline']['frac'], [(k['kernel'],k['points'],k['avg_ms']) for k in d['roofline'].get('kernels',[])])"
done 2>&1 | tee gpurun_out/sincos/bench_ab.txt
