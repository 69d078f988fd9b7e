#!/bin/bash
# rocprofv3 evidence for one of the OTHER BASELINE configurations (tools/time_config.py <cfg> <batch> [mode]) on the GPU box:
#   bash tools/profile_config.sh <tag> <cfg> <batch> [mode]      e.g.  r03a C5 4 b16   |   r03a C4 16
#  pass 1: --kernel-trace --stats (per-kernel time), pass 2 / 3: --pmc FETCH_SIZE / WRITE_SIZE (HBM bytes; separate passes, never with tracing domains)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SEGHIERO_WGRAD_STREAM=0
TAG=$1; CFG=$2; B=$3; MODE=${4:-f32}
OUT=gpurun_out/prof_${TAG}_${CFG}_${MODE}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/time_config.py $CFG $B $MODE > $OUT/trace.log 2>&1
echo "trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python tools/time_config.py $CFG $B $MODE > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python tools/time_config.py $CFG $B $MODE > $OUT/write.log 2>&1
echo "write done"
mkdir -p gpurun_out/profiles_out
python tools/summarize_config_profile.py $OUT gpurun_out/profiles_out/${TAG}_${CFG}_${MODE}
rm -rf $OUT/trace $OUT/fetch $OUT/write          # (raw traces stay on the box: gpurun merges at most 64 MiB back)
