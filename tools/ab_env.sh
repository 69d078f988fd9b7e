#!/bin/bash
# One program under two environments on one box, alternating processes:  bash tools/ab_env.sh <tag> "<VAR=val ...>" <rounds> <python args...>
TAG=$1; ENVB=$2; R=$3; shift 3
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
OUT=gpurun_out/abenv_$TAG.log; : > $OUT
for i in $(seq $R); do
  echo "== base round $i" >> $OUT
  timeout -k 10 400 python "$@" >> $OUT 2>&1 || { tail -20 $OUT; exit 1; }
  echo "== [$ENVB] round $i" >> $OUT
  timeout -k 10 400 env $ENVB python "$@" >> $OUT 2>&1 || { tail -20 $OUT; exit 1; }
done
cat $OUT
