mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 900 python scripts/kbench.py 4096 5 > gpurun_out/r4/kbench_h.log 2>&1; echo "kbench rc=$?"; grep -E "TRAINING|dgrad bf16x3|wgrad bf16x3|^S=|pair|Error|error|x3 \(inference" gpurun_out/r4/kbench_h.log | cut -c1-330
