#!/bin/bash
# Evidence of one round on the GPU box:  bash tools/profile_round.sh <tag>
#   1. bench under rocprofv3 --kernel-trace --stats (per-kernel time) and the two HBM PMC passes (tools/profile_bench.sh)
#   2. MFMA / LDS counters of the dominant conv shape (tools/pmc_conv.sh sep1.pw)
#   3. a sustained-clock throughput run (400 steps, ~14 s timed region)
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/profile_bench.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
echo "profile_bench done"
bash tools/pmc_conv.sh sep1.pw gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG.log 2>&1
echo "pmc done"
# 4. the north-star unit (ASPP depthwise-separable branches, forward) under the kernel trace
(cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG/aspp -- python tools/bench_aspp_branch.py > gpurun_out/aspp_unit_$TAG.json 2> gpurun_out/aspp_unit_$TAG.err)
echo "aspp unit done"
python bench.py --steps 400 --warmup 10 --no-cpu-baseline > gpurun_out/sustained_$TAG.json 2> gpurun_out/sustained_$TAG.err
echo "sustained done"; cat gpurun_out/sustained_$TAG.json | cut -c1-400
