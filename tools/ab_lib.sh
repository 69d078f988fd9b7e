#!/bin/bash
# Two builds of the library against each other on one box, alternating processes (kernel experiments):
#   bash tools/ab_lib.sh <tag> <old.so> [cfg] [batch] [rounds]
TAG=$1; OLD=$2; CFG=${3:-C2}; B=${4:-16}; R=${5:-3}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
OUT=gpurun_out/ab_$TAG.log; : > $OUT
for i in $(seq $R); do
  for which in old new; do
    if [ $which = old ]; then export SEGHIERO_LIB=$PWD/$OLD; else unset SEGHIERO_LIB; fi
    echo "== $which round $i" >> $OUT
    timeout -k 10 300 python tools/time_config.py $CFG $B >> $OUT 2>&1 || exit 1
  done
done
grep -E "^== |ms/step|one-stream|sh_conv_(fprop|dgrad)" $OUT
