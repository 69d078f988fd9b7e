#!/bin/bash
# GPU-box call for the pipelined conv kernels: op parity tests, then the per-shape micro-benchmark old vs new.
TAG=${1:-x6p_a}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --timeout=600 > gpurun_out/ops_$TAG.log 2>&1
echo "ops tests exit $?"; tail -3 gpurun_out/ops_$TAG.log
for v in "SEGHIERO_X6P=0" "SEGHIERO_X6P=1"; do
  echo "== $v" >> gpurun_out/convbench_$TAG.txt
  env $v timeout -k 10 300 python tools/bench_conv.py >> gpurun_out/convbench_$TAG.txt 2>&1 || echo "bench failed: $v"
done
grep -E "^==|TOTAL" gpurun_out/convbench_$TAG.txt
