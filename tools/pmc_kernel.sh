#!/bin/bash
# SQ counters of the kernels matching a regex during one bench step (GPU box):  bash tools/pmc_kernel.sh <regex> <out_dir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RE=${1:-loss_bwd_tile}; OUT=${2:-gpurun_out/pmck}
mkdir -p $OUT
rocprofv3 --kernel-trace --kernel-include-regex "$RE" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p1 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --kernel-include-regex "$RE" --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p2 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/p2.log 2>&1
find $OUT -name "*counter_collection.csv" | head
