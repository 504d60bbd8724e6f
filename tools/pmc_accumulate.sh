#!/bin/bash
# PMC passes over the accumulate kernel (separate rocprofv3 runs per counter group, --pmc only).
#   tools/pmc_accumulate.sh <out-prefix> <d> <G> <A> <W>
set -e
out=$1; d=$2; G=$3; A=$4; W=$5
cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/pmc
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES"; do
    dir=$ROOT/gpurun_out/pmc/$(echo $out | tr '/' '_')_$(echo $set | tr ' ' '_' | cut -c1-40)
    rm -rf $dir
    rocprofv3 --pmc $set --output-format csv -d $dir -- python3 $ROOT/tools/tune_accumulate.py --d $d --G $G --A $A --W $W --chunks 0 --reps 5 > /dev/null 2>&1
    python3 $ROOT/tools/pmc_summary.py $dir ctrl_accumulate
done
