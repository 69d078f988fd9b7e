#!/bin/bash
# One GPU-box call: the whole GPU suite, smoke(), then the bench with its defaults.  Usage: bash tools/gpu_round.sh <tag>
TAG=${1:-r02a}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/gputest_$TAG.log 2>&1
echo "gpu tests exit $?" | tee -a gpurun_out/gputest_$TAG.log
tail -5 gpurun_out/gputest_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$(date +%s)
[ -n "$NOBENCH" ] || python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench exit $? in $(( $(date +%s) - T0 )) s"
cat gpurun_out/bench_$TAG.json
