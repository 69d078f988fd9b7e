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
# condense on the box (the raw traces are too large to merge back) and leave the summaries under gpurun_out/profiles_out/
mkdir -p gpurun_out/profiles_out profiles_tmp
python tools/summarize_profile.py $TAG gpurun_out/profiles_out > gpurun_out/summarize_$TAG.log 2>&1
python tools/summarize_pmc.py gpurun_out/pmc_$TAG "conv_x6p_kernel|conv_wgrad_x6p" gpurun_out/profiles_out/${TAG}_pmc_sep1pw.json >> gpurun_out/summarize_$TAG.log 2>&1
cp gpurun_out/sustained_$TAG.json gpurun_out/profiles_out/${TAG}_sustained_bench.json
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_$TAG
