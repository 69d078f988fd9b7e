#!/bin/bash
# One GPU-box call: new parity tests first (no -x, full report), then the whole GPU suite, then the bench.  Usage: bash tools/gpu_round.sh <tag>
TAG=${1:-r02a}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/gputest_$TAG.log 2>&1
echo "gpu tests exit $?" | tee -a gpurun_out/gputest_$TAG.log
tail -5 gpurun_out/gputest_$TAG.log
[ -n "$NOBENCH" ] || python bench.py --steps 20 --warmup 3 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench exit $?"
cat gpurun_out/bench_$TAG.json
