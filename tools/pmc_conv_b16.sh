#!/bin/bash
# PMC counters for one conv shape, bf16 compute kernels (usage on the GPU box: bash tools/pmc_conv.sh "sep1.pw" out_dir)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SHAPE=${1:-sep1.pw}; OUT=${2:-gpurun_out/pmc}
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python tools/bench_conv_b16.py "$SHAPE" > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d $OUT/p2 -- python tools/bench_conv_b16.py "$SHAPE" > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr --output-format csv -d $OUT/p3 -- python tools/bench_conv_b16.py "$SHAPE" > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/p4 -- python tools/bench_conv_b16.py "$SHAPE" > $OUT/p4.log 2>&1
find $OUT -name "*counter_collection.csv" | head
