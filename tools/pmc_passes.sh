#!/bin/bash
# rocprofv3 PMC passes over the path kernel (counters only; never combined with trace domains other
# than --kernel-trace).  Usage (on the GPU box): tools/pmc_passes.sh <outdir> [profile_paths.py args]
set -e
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
         "GRBM_GUI_ACTIVE GRBM_COUNT" \
         "WRITE_SIZE" \
         "FETCH_SIZE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pass$i -- python3 $R/tools/profile_paths.py "$@" > $R/$OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
ls -R $R/$OUT | head -40
