#!/bin/bash
# PMC passes over one kernel of any tool script (separate rocprofv3 runs per counter group, --pmc only).
#   tools/pmc_kernel.sh <out-prefix> <kernel-name-substring> <tool.py> [tool args ...]
set -e
out=$1; kern=$2; shift 2
cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/pmc
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT"; do
    dir=$ROOT/gpurun_out/pmc/$(echo $out | tr '/' '_')_$(echo $set | tr ' ' '_' | cut -c1-40)
    rm -rf $dir
    rocprofv3 --pmc $set --output-format csv -d $dir -- python3 $ROOT/"$@" > /dev/null 2>&1 || echo "(counter set failed: $set)"
    python3 $ROOT/tools/pmc_summary.py $dir $kern || true
done
