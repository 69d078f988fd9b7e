#!/bin/bash
# One GPU-box call: selected GPU tests.  Usage: bash tools/gpu_sel.sh <tag> <pytest args...>
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest -m gpu -q --timeout=600 -x "$@" > gpurun_out/sel_$TAG.log 2>&1
echo "exit $?" | tee -a gpurun_out/sel_$TAG.log
tail -25 gpurun_out/sel_$TAG.log
