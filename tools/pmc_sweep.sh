#!/bin/bash
# PMC passes over the MFMA sweep kernel (bench.py --sweep with 20 whole 512-portfolio workgroup rows: one kernel variant).  Usage on the GPU box: tools/pmc_sweep.sh <outdir>
set -e
OUT=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pass$i -- python3 $R/bench.py --sweep --sweep-portfolios 10240 --steps 1 --warmup 1 --sweep-paths 65536 > $R/$OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
