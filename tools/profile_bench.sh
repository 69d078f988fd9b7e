#!/bin/bash
# Round profile of the bench command on the GPU box:  bash tools/profile_bench.sh <tag>
#  pass 1: rocprofv3 --kernel-trace --stats          -> per-kernel time
#  pass 2/3: --pmc FETCH_SIZE / --pmc WRITE_SIZE      -> HBM-side bytes per dispatch (separate passes: TCC has 4 slots)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# per-kernel durations and counters are taken with the weight-gradient side stream off (co-running kernels share the CUs and
# inflate each other's durations); bench.py's own instrumented step does the same.  Throughput is reported with it on.
export SEGHIERO_WGRAD_STREAM=0
TAG=${1:-r01}; OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-units --no-bf16 > $OUT/trace.log 2>&1
echo "trace done" 
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-units --no-bf16 > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-units --no-bf16 > $OUT/write.log 2>&1
echo "write done"
find $OUT -name "*.csv" | head -20
