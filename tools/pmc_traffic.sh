#!/bin/bash
# HBM traffic per kernel of a bench.py --sweep step (statistics passes and the MFMA kernel): FETCH_SIZE and WRITE_SIZE in passes of
# their own (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2).  Usage on the GPU box: tools/pmc_traffic.sh <outdir> [bench args]
set -e
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/$C -- python3 $R/bench.py "$@" > $R/$OUT/$C.log 2>&1 || echo "$C pass failed"
done
